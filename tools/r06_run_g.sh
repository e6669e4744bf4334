python tools/dbg_fuzz_gpu_case.py 11 1038
cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/orig.so; cp gpurun_variants/lib_r06_sidemarks.so jpegsnoop_amd/libjsnoop_gpu.so
python tools/dbg_fuzz_gpu_case.py 11 1038
cp /tmp/orig.so jpegsnoop_amd/libjsnoop_gpu.so
