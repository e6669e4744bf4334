"""The back end issues its row / DC loads by inline asm and waits for them by hand (back_end_pairs, jsnoop_kernels.hip): the compiler does not know that
those registers are in flight.  This check compiles the kernels to assembly and, for every k_idct_color instance, lists every instruction that READS a
register a `global_load_*` of the MCU loop writes: only the masking `v_and_b32` (rows), the `v_add_u32` of the DC word and the loads themselves may appear;
a `v_mov` / copy of such a register would read it before its data has landed.   usage: python tools/check_inflight_regs.py   (CPU only, ~1 min)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "jpegsnoop_amd", "csrc", "jsnoop_kernels.hip")
out = os.path.join(tempfile.gettempdir(), "jsnoop_kernels_check.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only", "-S",
                       "-I" + os.path.dirname(src), "-o", out, src], stderr=subprocess.DEVNULL)
text = open(out).read()
bad = 0
for m in re.finditer(r"^(_Z12k_idct_colorILi([1-4])E\w+):\s*; @.*?^\.Lfunc_end", text, re.S | re.M):     # (<0>, the any-layout kernel, leaves loads and waits to the compiler)
    body = m.group(0).split("\n")
    # the MCU loop: from the first in-loop s_waitcnt vmcnt placed by hand (the value 2*np repeats) -- take every global_load with an SGPR base in the function
    loads = [(i, l) for i, l in enumerate(body) if re.match(r"\s+global_load_(dword|sshort) v\d+, v\d+, s\[", l)]
    regs = sorted({re.match(r"\s+global_load_\w+ (v\d+),", l).group(1) for _, l in loads[7:]} or {re.match(r"\s+global_load_\w+ (v\d+),", l).group(1) for _, l in loads})
    readers = {}
    for l in body:
        l2 = l.split(";")[0]
        mm = re.match(r"\s+(\w+)\s+(.*)", l2)
        if not mm or mm.group(1).startswith("global_load") or mm.group(1) == "s_waitcnt": continue
        ops = [o.strip() for o in mm.group(2).split(",")]
        for r in regs:
            if r in ops[1:]:
                readers.setdefault(r, set()).add(mm.group(1))
    ok = all(v <= {"v_and_b32_e32", "v_add_u32_e32", "v_mov_b32_e32"} - {"v_mov_b32_e32"} for v in readers.values())
    print("k_idct_color<%s>: in-flight registers %s; read by %s -> %s" % (m.group(2), regs, {k: sorted(v) for k, v in readers.items()}, "ok" if ok else "CHECK"))
    bad += 0 if ok else 1
sys.exit(1 if bad else 0)
