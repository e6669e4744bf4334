mkdir -p gpurun_out/r06_f
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06_f/pytest.log 2>&1; tail -3 gpurun_out/r06_f/pytest.log
timeout 900 python tools/fuzz_damaged_log.py 2400 13 > gpurun_out/r06_f/fuzz_small.txt 2>&1; grep -A3 "MISMATCH\|LOG\|exception" gpurun_out/r06_f/fuzz_small.txt | head -30; tail -1 gpurun_out/r06_f/fuzz_small.txt
timeout 900 python tools/fuzz_damaged_log.py 300 9 1 > gpurun_out/r06_f/fuzz_big.txt 2>&1; grep -A3 "MISMATCH\|LOG\|exception" gpurun_out/r06_f/fuzz_big.txt | head; tail -1 gpurun_out/r06_f/fuzz_big.txt
timeout 900 python tools/fuzz_gpu.py 3000 17 > gpurun_out/r06_f/fuzz_gpu.txt 2>&1; tail -4 gpurun_out/r06_f/fuzz_gpu.txt
