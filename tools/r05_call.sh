mkdir -p gpurun_out/r05_b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_b/pytest.log; tail -5 gpurun_out/r05_b/pytest.log
bash tools/ab_env.sh r05_b "tree_fused|-||" "tree_3pass|-|JSNOOP_UNSTUFF_3PASS=1|" "c_contig|gpurun_variants/lib_c_contig.so||" "c_prefetch|gpurun_variants/lib_c_prefetch.so||" "c_write2|gpurun_variants/lib_c_write2.so||" "c_all3|gpurun_variants/lib_c_all3.so||" "c_all3_mpw16|gpurun_variants/lib_c_all3.so|JSNOOP_MPW=16|" "c_all3_mpw32|gpurun_variants/lib_c_all3.so|JSNOOP_MPW=32|" > gpurun_out/r05_b/ab.log 2>&1
cat gpurun_out/r05_b/ab.log | cut -c1-250
timeout 500 python tools/fuzz_1080p_timing.py 1000 17 > gpurun_out/r05_b/fuzz_1080p_timing.log 2>&1; cp gpurun_out/fuzz_1080p_timing.json gpurun_out/r05_b/ 2>/dev/null; tail -30 gpurun_out/r05_b/fuzz_1080p_timing.log
timeout 300 python tools/fuzz_gpu.py 2500 4243 > gpurun_out/r05_b/fuzz_gpu.log 2>&1; tail -3 gpurun_out/r05_b/fuzz_gpu.log
