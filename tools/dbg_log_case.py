"""Replays case K of tools/fuzz_damaged_log.py (seed S, small bases) alone and prints the first differing log line: python tools/dbg_log_case.py S K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]] + sys.argv[1:]
seed, K = int(sys.argv[1]), int(sys.argv[2])
src = open(os.path.join(ROOT, "tools", "fuzz_damaged_log.py")).read()
src = src.replace('n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200', 'n_cases = %d' % (K + 1)).replace('rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)', 'rng = np.random.default_rng(%d)' % seed)
src = src.replace('big = len(sys.argv) > 3 and sys.argv[3] == "1"', 'big = False')
exec(compile(src, "fuzz_damaged_log.py", "exec"))
