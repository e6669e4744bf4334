mkdir -p gpurun_out/r05_k
timeout 600 python -m pytest tests/test_gpu_damaged.py tests/test_gpu_parity.py tests/test_gpu_batch_results.py -m gpu -q --maxfail=10 > gpurun_out/r05_k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_k/pytest.log; tail -8 gpurun_out/r05_k/pytest.log
timeout 400 python tools/fuzz_1080p_timing.py 1000 23 > gpurun_out/r05_k/fuzz_1080p_timing_s23.log 2>&1; head -5 gpurun_out/r05_k/fuzz_1080p_timing_s23.log
JSNOOP_SUB_WL=5 timeout 250 python tools/fuzz_gpu.py 2500 9105 > gpurun_out/r05_k/fuzz_gpu_wl5.log 2>&1; tail -3 gpurun_out/r05_k/fuzz_gpu_wl5.log
JSNOOP_SUB_WL=7 timeout 250 python tools/fuzz_gpu.py 2500 9107 > gpurun_out/r05_k/fuzz_gpu_wl7.log 2>&1; tail -3 gpurun_out/r05_k/fuzz_gpu_wl7.log
timeout 250 python tools/fuzz_gpu.py 2500 9100 > gpurun_out/r05_k/fuzz_gpu.log 2>&1; tail -3 gpurun_out/r05_k/fuzz_gpu.log
timeout 200 python tools/fuzz_batch.py 32 > gpurun_out/r05_k/fuzz_batch.log 2>&1; tail -2 gpurun_out/r05_k/fuzz_batch.log
