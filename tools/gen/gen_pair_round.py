#!/usr/bin/env python3
"""Generates jpegsnoop_amd/csrc/jsnoop_pair_round.inc: the sixteen-step rounds of the pair form's term loop (k_idct_color, idct_pair).

A step = term K of both blocks of a pair: one ds_read_b64 of the lane's two table entries, two multiplies, two adds.  Round 6: the
COEFFICIENT no longer rides on the multiplies as a DPP operand (v_mul_f32_dpp issues in 4.2 cycles, a plain v_mul_f32 in 2.2,
profiles/r04_instr_rates.txt) -- the coefficients of two consecutive terms come as ONE broadcast ds_read_b64 (every lane of a half reads
the same eight bytes of its half's list: one LDS cycle per half, no conflict), i.e. half an LDS instruction more per step for 4 vector
cycles less.  The row address stays a v_add_u32_dpp on the round's row words.

The LDS returns in order: every wait is `lgkmcnt(number of LDS instructions issued behind the one waited for)`, computed here.
"""
import sys

import os
T_RING = int(os.environ.get("GEN_T_RING", 5))    # table pairs: four reads in flight + the one being consumed
C_RING = 3           # coefficient pairs
T_LEAD = int(os.environ.get("GEN_T_LEAD", 4))    # a table pair is fetched this many steps ahead of its step
C_LEAD = 4           # a coefficient pair (steps 2j, 2j+1) is fetched at step 2j - C_LEAD
MIX = int(os.environ.get("GEN_MIX", 0))          # experiment: every MIX-th coefficient pair rides on its multiplies as a DPP operand of %[ey] instead (0: none)
T_REG0 = 54          # v[54:63]: five table pairs
C_REG0 = 48          # v[48:53]: three coefficient pairs


def treg(s):
    r = T_REG0 + 2 * (s % T_RING)
    return r, r + 1


def creg(j):
    r = C_REG0 + 2 * (j % C_RING)
    return r, r + 1


def gen_round(r, steps=16):
    out = []
    q = []                                  # LDS instructions in flight, in issue order (names)

    def emit(s):
        out.append(s)

    def rd_c(j):
        lo, hi = creg(j)
        emit(f"ds_read_b64 v[{lo}:{hi}], %[ah] offset:{r * 64 + j * 8}")
        q.append(("C", j))

    def rd_t(s):
        lo, hi = treg(s)
        emit(f"v_add_u32_dpp %[ad], %[rw], %[l8] row_newbcast:{s} row_mask:0xf bank_mask:0xf")
        emit(f"ds_read_b64 v[{lo}:{hi}], %[ad]")
        q.append(("T", s))

    def wait(*names):
        # everything up to and including the LAST of `names` in the queue must have returned
        idx = max(q.index(n) for n in names if n in q) if any(n in q for n in names) else -1
        if idx < 0:
            return
        behind = len(q) - 1 - idx
        emit(f"s_waitcnt lgkmcnt({behind})")
        del q[:idx + 1]

    def dpp_pair(j):
        return MIX > 0 and j % MIX == MIX - 1

    # prologue: the round's row words, the first two coefficient pairs, four table pairs
    if MIX:
        emit(f"ds_read_b32 %[ey], %[aey] offset:{r * 64}")
        q.append(("EY", 0))
    emit(f"ds_read_b32 %[rw], %[arw] offset:{r * 64}")
    q.append(("RW", 0))
    for j in range(0, C_LEAD // 2):
        if not dpp_pair(j):
            rd_c(j)
    wait(("RW", 0))
    for s in range(T_LEAD):
        rd_t(s)
    for s in range(steps):
        if s:
            emit(f"s_cmp_le_u32 %[nl], {s}")
            emit("s_cbranch_scc1 .Lpe%=")
        lo, hi = treg(s)
        if dpp_pair(s // 2):
            wait(("T", s), ("EY", 0))
            emit(f"v_mul_f32_dpp v{lo}, %[ey], v{lo} row_newbcast:{s} row_mask:0xf bank_mask:0xf")
            emit(f"v_mul_f32_dpp v{hi}, %[ey], v{hi} row_newbcast:{s} row_mask:0xf bank_mask:0xf")
        else:
            wait(("T", s), ("C", s // 2))
            c = creg(s // 2)[s & 1]
            emit(f"v_mul_f32 v{lo}, v{c}, v{lo}")
            emit(f"v_mul_f32 v{hi}, v{c}, v{hi}")
        if s + T_LEAD < steps:
            rd_t(s + T_LEAD)
        if s % 2 == 0 and (s + C_LEAD) // 2 < steps // 2 and not dpp_pair((s + C_LEAD) // 2):
            rd_c((s + C_LEAD) // 2)
        emit(f"v_add_f32 %[acc0], %[acc0], v{lo}")
        emit(f"v_add_f32 %[acc1], %[acc1], v{hi}")
    emit(".Lpe%=:")
    emit("s_waitcnt lgkmcnt(0)")
    return out


def gen_round_entries(r, steps=16, t_lead=3, e_lead=3, t_ring=4, e_ring=7, t_reg0=56, e_reg0=40):
    """Probe only (tools/probes/idct_bcast.hip): coefficient AND row word of a term as one broadcast ds_read_b64 (an entry), the
    address a plain v_add_u32 -- 11 vector cycles per step, but two LDS instructions."""
    out, q = [], []
    emit = out.append
    tr = lambda s: (t_reg0 + 2 * (s % t_ring), t_reg0 + 2 * (s % t_ring) + 1)
    er = lambda k: (e_reg0 + 2 * (k % e_ring), e_reg0 + 2 * (k % e_ring) + 1)

    def rd_e(k):
        lo, hi = er(k)
        emit(f"ds_read_b64 v[{lo}:{hi}], %[ah] offset:{r * 128 + k * 8}")
        q.append(("E", k))

    def rd_t(s):
        lo, hi = tr(s)
        emit(f"v_add_u32 %[ad], v{er(s)[1]}, %[l8]")
        emit(f"ds_read_b64 v[{lo}:{hi}], %[ad]")
        q.append(("T", s))

    def wait(*names):
        live = [n for n in names if n in q]
        if not live:
            return
        idx = max(q.index(n) for n in live)
        emit(f"s_waitcnt lgkmcnt({len(q) - 1 - idx})")
        del q[:idx + 1]

    for k in range(min(steps, t_lead + e_lead)):
        rd_e(k)
    for s in range(t_lead):
        wait(("E", s))
        rd_t(s)
    for s in range(steps):
        if s:
            emit(f"s_cmp_le_u32 %[nl], {s}")
            emit("s_cbranch_scc1 .Lpe%=")
        need = [("T", s)]
        if s + t_lead < steps:
            need.append(("E", s + t_lead))
        wait(*need)
        lo, hi = tr(s)
        c = er(s)[0]
        emit(f"v_mul_f32 v{lo}, v{c}, v{lo}")
        emit(f"v_mul_f32 v{hi}, v{c}, v{hi}")
        emit(f"v_add_f32 %[acc0], %[acc0], v{lo}")
        emit(f"v_add_f32 %[acc1], %[acc1], v{hi}")
        if s + t_lead < steps:
            rd_t(s + t_lead)
        if s + t_lead + e_lead < steps:
            rd_e(s + t_lead + e_lead)
    emit(".Lpe%=:")
    emit("s_waitcnt lgkmcnt(0)")
    return out


def main():
    path = sys.argv[1]
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen/gen_pair_round.py -- do not edit.  The sixteen-step rounds of idct_pair (jsnoop_kernels.hip).\n")
        f.write(f"// table pairs v[{T_REG0}:{T_REG0 + 2 * T_RING - 1}], coefficient pairs v[{C_REG0}:{C_REG0 + 2 * C_RING - 1}]\n")
        for r in range(4):
            f.write(f"#define PAIR_ROUND_ASM_{r} \\\n")
            lines = gen_round(r)
            for i, l in enumerate(lines):
                f.write(f'    "{l}\\n\\t"' + (" \\\n" if i + 1 < len(lines) else "\n"))
        regs = ", ".join(f'"v{i}"' for i in range(C_REG0, T_REG0 + 2 * T_RING))
        f.write(f"#define PAIR_ROUND_CLOBBERS {regs}\n")
        if len(sys.argv) > 2 and sys.argv[2] == "--probe":
            f.write("#define ENTRY_ROUND_ASM_0 \\\n")
            lines = gen_round_entries(0)
            for i, l in enumerate(lines):
                f.write(f'    "{l}\\n\\t"' + (" \\\n" if i + 1 < len(lines) else "\n"))
            regs = ", ".join(f'"v{i}"' for i in range(40, 64))
            f.write(f"#define ENTRY_ROUND_CLOBBERS {regs}\n")


if __name__ == "__main__":
    main()
