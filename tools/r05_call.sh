mkdir -p gpurun_out/r05_c
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r05_c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_c/pytest.log; tail -5 gpurun_out/r05_c/pytest.log
bash tools/ab_env.sh r05_c "tree_fused|-||" "tree_3pass|-|JSNOOP_UNSTUFF_3PASS=1|" "c_contig|gpurun_variants/lib_c_contig.so||" "c_prefetch|gpurun_variants/lib_c_prefetch.so||" "c_write2|gpurun_variants/lib_c_write2.so||" "c_all3|gpurun_variants/lib_c_all3.so||"  > gpurun_out/r05_c/ab.log 2>&1
cat gpurun_out/r05_c/ab.log | cut -c1-250
timeout 500 python tools/fuzz_1080p_timing.py 1000 17 > gpurun_out/r05_c/fuzz_1080p_timing.log 2>&1; cp gpurun_out/fuzz_1080p_timing.json gpurun_out/r05_c/ 2>/dev/null; tail -30 gpurun_out/r05_c/fuzz_1080p_timing.log
timeout 300 python tools/fuzz_gpu.py 2500 4243 > gpurun_out/r05_c/fuzz_gpu.log 2>&1; tail -3 gpurun_out/r05_c/fuzz_gpu.log
