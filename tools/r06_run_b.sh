mkdir -p gpurun_out/r06_d
for V in r06_stamps r06_stamps_tile; do
  cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/orig.so; cp gpurun_variants/lib_$V.so jpegsnoop_amd/libjsnoop_gpu.so
  python tools/backend_stamps.py 256 16 > gpurun_out/r06_d/$V.json 2> gpurun_out/r06_d/$V.err; cat gpurun_out/r06_d/$V.json; tail -2 gpurun_out/r06_d/$V.err
  cp /tmp/orig.so jpegsnoop_amd/libjsnoop_gpu.so
done
AB_SKIP_TESTS=1 tools/ab_round.sh r06_d gpurun_variants/lib_r06_listpred.so gpurun_variants/lib_r05base.so
