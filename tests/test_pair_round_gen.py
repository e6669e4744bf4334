"""CPU: the generated rounds of the back end's term loop (jpegsnoop_amd/csrc/jsnoop_pair_round.h, tools/gen/gen_pair_round.py).

The LDS returns in order and every `s_waitcnt lgkmcnt(N)` in the rounds is a COUNT of younger LDS instructions: a wrong count reads a register whose data
has not landed -- garbage that a parity test may or may not catch.  Here the text is replayed against an in-order queue: no instruction may read a register
an LDS read still in flight writes, and no LDS read may land in a register a pending instruction... (the latter cannot happen in order).  Also: the
committed header is what the generator writes."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "jpegsnoop_amd", "csrc", "jsnoop_pair_round.h")


def rounds():
    txt = open(HDR).read()
    for m in re.finditer(r"#define PAIR_ROUND_ASM_(\d) \\\n((?:    \".*\n)+)", txt):
        lines = [re.match(r'\s*"(.*?)\\n\\t"', l).group(1) for l in m.group(2).strip().split("\n")]
        yield int(m.group(1)), lines


def regs_of(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return ["v%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1)]
    if re.match(r"v\d+$", tok) or tok in ("%[rw]", "%[ey]"):
        return [tok]
    return []


def check(lines, steps_taken):
    """Replays the round up to the exit of step `steps_taken` (None: all sixteen)."""
    inflight = []                                                # [(set of dest registers)] in issue order
    step = 0
    for l in lines:
        op, _, rest = l.partition(" ")
        args = [a for a in re.split(r",\s*", rest.split(" row_newbcast")[0].split(" offset")[0]) if a]
        if op == "s_waitcnt":
            n = int(re.search(r"lgkmcnt\((\d+)\)", l).group(1))
            assert n <= 15
            while len(inflight) > n:
                inflight.pop(0)
        elif op.startswith("ds_read"):
            dst = regs_of(args[0])
            for r in regs_of(args[1]):
                assert all(r not in q for q in inflight), (l, "address register in flight")
            inflight.append(set(dst))
        elif op.startswith("v_"):
            for a in args[1:]:                                   # sources
                for r in regs_of(a):
                    assert all(r not in q for q in inflight), (l, "reads a register whose LDS read is in flight", inflight)
            for r in regs_of(args[0]):                           # a destination must not be overwritten by a read that lands later
                assert all(r not in q for q in inflight), (l, "destination has an LDS read in flight")
        elif op == "s_cmp_le_u32":
            step = int(args[1])
            if steps_taken is not None and step == steps_taken:
                return                                            # the exit is taken: .Lpe waits for lgkmcnt(0)
        elif op in ("s_cbranch_scc1", ".Lpe%=:"):
            pass
        else:
            raise AssertionError("unexpected line " + l)
    assert not inflight or lines[-1].startswith("s_waitcnt lgkmcnt(0)")


def test_every_wait_of_the_generated_rounds_covers_the_registers_read_behind_it():
    n = 0
    for r, lines in rounds():
        assert lines[-1] == "s_waitcnt lgkmcnt(0)" and lines[-2] == ".Lpe%=:"
        for taken in [None] + list(range(1, 16)):
            check(lines, taken)
        # sixteen steps: 16 table reads, 8 coefficient pairs, the row words
        assert sum(1 for l in lines if l.startswith("ds_read_b64") and "%[ad]" in l) == 16
        assert sum(1 for l in lines if l.startswith("ds_read_b64") and "%[ah]" in l) == 8
        assert sum(1 for l in lines if l.startswith("v_mul_f32 ")) == 32 and sum(1 for l in lines if l.startswith("v_add_f32 ")) == 32
        # the coefficient of step s is the low / high word of pair s / 2, read at list offset round * 64 + (s / 2) * 8
        offs = sorted(int(re.search(r"offset:(\d+)", l).group(1)) for l in lines if "%[ah]" in l)
        assert offs == [r * 64 + j * 8 for j in range(8)]
        n += 1
    assert n == 4


def test_the_committed_header_is_the_generators_output(tmp_path):
    out = tmp_path / "gen.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_pair_round.py"), str(out)],
                          env={k: v for k, v in os.environ.items() if not k.startswith("GEN_")})
    assert out.read_text() == open(HDR).read()


def test_no_instruction_of_the_built_back_end_copies_a_register_in_flight():
    """The one-layout back-end kernels issue their row / DC loads by inline asm and wait for them by hand (back_end_pairs): the compiler does not know those
    registers are in flight.  tools/check_inflight_regs.py compiles the kernels to assembly (hipcc cross-compiles without a GPU) and lists every instruction
    that reads a register the MCU loop's loads write: only the mask (v_and_b32) and the DC add (v_add_u32) may -- a copy would read it before the data lands."""
    import shutil
    import pytest
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight_regs.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and out.count("-> ok") == 4, out
