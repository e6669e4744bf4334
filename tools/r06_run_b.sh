mkdir -p gpurun_out/r06_b
for V in r06_stamps r06_stamps_tile; do
  cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/orig.so; cp gpurun_variants/lib_$V.so jpegsnoop_amd/libjsnoop_gpu.so
  python tools/backend_stamps.py 256 16 > gpurun_out/r06_b/$V.json 2> gpurun_out/r06_b/$V.err; cat gpurun_out/r06_b/$V.json; tail -2 gpurun_out/r06_b/$V.err
  cp /tmp/orig.so jpegsnoop_amd/libjsnoop_gpu.so
done
