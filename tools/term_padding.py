"""CPU: lock-step padding of the back end's term loop on the bench workload.  The back end sums, per 8x8 block, one term per NON-ZERO AC
coefficient; blocks that share a wave advance in lock step, a step costs the wave the same whether one or all of its lists have a term left.  From the oracle's
dequantised coefficients of the bench pictures (1920x1080 4:2:0 q85, seeds of bench.py): terms needed per MCU, steps executed by
  pair form (the tree: two blocks per wave -- (Y0,Y1) (Y2,Y3) (Cb,Cr)),
  quad form A (four blocks per wave, one MCU: (Y0..Y3) (Cb,Cr,-,-)),  quad form B (two MCUs per wave: (Y0..Y3) (Y0'..Y3') (Cb,Cr,Cb',Cr')),
and the vector-issue cycles per MCU under the instruction costs of profiles/r02_instr_rates.txt (pair step: 2 DPP multiplies + 2 adds + DPP address add = 16.7;
quad step: DPP move + DPP address add + 4 multiplies + 4 adds = 25.8).   usage: python tools/term_padding.py [pictures]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import harness as H
H.build(["oracle", "synth"])
n_pic = int(sys.argv[1]) if len(sys.argv) > 1 else 8
orc = H.oracle_backend()
need = pair = quad_a = quad_b = mcus = 0
for i in range(n_pic):
    f = H.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=i + 1)
    H.drive(orc, f)
    c = H.oracle_coefs(orc).reshape(-1, 6, 64)                       # [MCU][block][natural index]
    nz = (c[:, :, 1:] != 0).sum(axis=2)                              # terms per block (the DC slot is not part of the sum)
    need += int(nz.sum()); mcus += nz.shape[0]
    pair += int(np.maximum(nz[:, 0], nz[:, 1]).sum() + np.maximum(nz[:, 2], nz[:, 3]).sum() + np.maximum(nz[:, 4], nz[:, 5]).sum())
    quad_a += int(nz[:, :4].max(axis=1).sum() + nz[:, 4:].max(axis=1).sum())
    m2 = nz[: nz.shape[0] // 2 * 2].reshape(-1, 2, 6)
    quad_b += int(m2[:, 0, :4].max(axis=1).sum() + m2[:, 1, :4].max(axis=1).sum() + np.maximum(m2[:, 0, 4:].max(axis=1), m2[:, 1, 4:].max(axis=1)).sum())
orc.close()
print("pictures %d, MCUs %d, terms needed per MCU %.2f (per block %.2f)" % (n_pic, mcus, need / mcus, need / mcus / 6))
print("pair form:   %.2f steps per MCU = %.2f block-terms executed per term needed, %.0f vector cycles per MCU" % (pair / mcus, 2 * pair / need, 16.7 * pair / mcus))
print("quad form A: %.2f steps per MCU = %.2f block-terms executed per term needed, %.0f vector cycles per MCU" % (quad_a / mcus, 4 * quad_a / need, 25.8 * quad_a / mcus))
print("quad form B: %.2f steps per MCU = %.2f block-terms executed per term needed, %.0f vector cycles per MCU" % (quad_b / mcus, 4 * quad_b / need, 25.8 * quad_b / mcus))
