mkdir -p gpurun_out/r05_h
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r05_h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_h/pytest.log; tail -5 gpurun_out/r05_h/pytest.log
python - <<'PY' > gpurun_out/r05_h/mirror_speed.txt 2>&1
import time, sys
sys.path.insert(0, '.')
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
orc = H.oracle_backend()
for kw in (dict(width=1920, height=1080, seed=5), dict(width=1920, height=1080, seed=6, restart_interval=120), dict(width=3840, height=2160, seed=7)):
    f = H.synth_jpeg(**kw)
    b = J.JpegBatch(); b._lib.jsnoop_batch_set_options(b._h, 1, 0, 1)      # force the exact-mirror kernel for the whole image
    b.add_jpeg(f); b.upload(); b.decode(); b.sync()
    t = time.perf_counter(); b.decode(); b.sync(); ms = (time.perf_counter() - t) * 1e3
    H.drive(orc, f)
    print(kw, "whole image through the sequential mirror: %.1f ms, bit-exact %s, path %d" % (ms, int(b.dib_checksums()[0]) == J.dib_checksum_numpy(orc.dib()), b.info(0)["path"]))
    b.close()
PY
cat gpurun_out/r05_h/mirror_speed.txt
timeout 500 python tools/fuzz_1080p_timing.py 1000 23 > gpurun_out/r05_h/fuzz_1080p_timing_s23.log 2>&1; head -8 gpurun_out/r05_h/fuzz_1080p_timing_s23.log
timeout 500 python tools/fuzz_1080p_timing.py 1000 17 > gpurun_out/r05_h/fuzz_1080p_timing_s17.log 2>&1; head -6 gpurun_out/r05_h/fuzz_1080p_timing_s17.log
timeout 300 python tools/fuzz_gpu.py 4000 9002 > gpurun_out/r05_h/fuzz_gpu.log 2>&1; tail -3 gpurun_out/r05_h/fuzz_gpu.log
timeout 200 python tools/fuzz_batch.py 32 > gpurun_out/r05_h/fuzz_batch.log 2>&1; tail -2 gpurun_out/r05_h/fuzz_batch.log
python tools/call_latency.py 2>/dev/null | tail -1 > gpurun_out/r05_h/call_latency.json; cat gpurun_out/r05_h/call_latency.json
