"""Replays case K of tools/fuzz_damaged_log.py (seed S, small bases), writes the file and both logs to gpurun_out/: python tools/dbg_dump_log_case.py S K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seed, K = int(sys.argv[1]), int(sys.argv[2])
src = open(os.path.join(ROOT, "tools", "fuzz_damaged_log.py")).read()
src = src.replace('n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200', 'n_cases = %d' % (K + 1)).replace('rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)', 'rng = np.random.default_rng(%d)' % seed)
src = src.replace('big = len(sys.argv) > 3 and sys.argv[3] == "1"', 'big = False')
src = src.replace('    if ref is not None:\n        H.drive(ref, data, q, quiet=0); want = ref.log_lines()', '    if ref is not None:\n        H.drive(ref, data, q, quiet=0); want = ref.log_lines()\n        if k == %d:\n            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)\n            open(os.path.join(ROOT, "gpurun_out", "case_%d_%d.jpg"), "wb").write(data); open(os.path.join(ROOT, "gpurun_out", "case_%d_%d_got.txt"), "w").write("\\n".join(got)); open(os.path.join(ROOT, "gpurun_out", "case_%d_%d_want.txt"), "w").write("\\n".join(want)); print("em", em)' % (K, seed, K, seed, K, seed, K))
exec(compile(src, "fuzz_damaged_log.py", "exec"))
