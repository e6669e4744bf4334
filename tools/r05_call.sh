mkdir -p gpurun_out/r05_i
timeout 600 python -m pytest tests/test_gpu_entropy_variants.py tests/test_gpu_damaged.py tests/test_gpu_parity.py tests/test_gpu_batch_large.py -m gpu -q --maxfail=10 > gpurun_out/r05_i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_i/pytest.log; tail -5 gpurun_out/r05_i/pytest.log
bash tools/ab_env.sh r05_i "tree1|-||" "head1|gpurun_variants/lib_head.so||" "tree2|-||" "head2|gpurun_variants/lib_head.so||" "tree3|-||"
for L in - gpurun_variants/lib_head.so gpurun_variants/lib_r04.so; do
  cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/keep.so; [ "$L" != "-" ] && cp $L jpegsnoop_amd/libjsnoop_gpu.so
  echo "dri_batch $L" >> gpurun_out/r05_i/dri_batch.txt; timeout 300 python tools/dri_batch.py 2>/dev/null | tail -1 >> gpurun_out/r05_i/dri_batch.txt
  cp /tmp/keep.so jpegsnoop_amd/libjsnoop_gpu.so
done
cat gpurun_out/r05_i/dri_batch.txt
