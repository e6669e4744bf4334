"""CPU: bench.py's rank / shard / reduce logic end to end at world size 2 (gloo), decode stubbed (--stub).

`bench.py --gpus 2 --stub` has no RANK in its environment, so it must spawn the two ranks itself -- the path the
driver's `python bench.py --gpus N` takes -- and rank 0 must print one JSON line for the whole job."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return p


def last_json(p):
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout.decode()[-2000:], p.stderr.decode()[-2000:])
    return json.loads(lines[-1])


def test_gpus_2_spawns_two_ranks_and_reduces():
    images, steps, w, h = 8, 3, 64, 48
    out = last_json(run_bench("--gpus", "2", "--stub", "--stub-ms", "5", "--images", str(images), "--distinct", "4", "--steps", str(steps), "--warmup", "1",
                              "--width", str(w), "--height", str(h)))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["parallelism"] == "shard2"
    assert len(out["per_rank_ms_per_step"]) == 2 and all(ms >= 5.0 for ms in out["per_rank_ms_per_step"])
    assert out["bit_exact"] and out["parity_errors"] == 0
    # value = pixels of BOTH ranks / slowest rank's time
    assert abs(out["value"] - 2 * images * w * h * steps / (out["ms_per_step"] * steps * 1e-3) / 1e6) < 0.05 * out["value"] + 0.2
    assert out["ms_per_step"] >= max(out["per_rank_ms_per_step"]) - 1e-3
    # the job checksum is the sum (mod 2^64) of the per-rank XORs of the stub's per-image sums
    import numpy as np
    want = 0
    for r in range(2):
        sums = (np.arange(images, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(r + 1)).astype(np.uint64)
        want = (want + int(np.bitwise_xor.reduce(sums))) & 0xFFFFFFFFFFFFFFFF
    assert int(out["job_checksum"], 16) == want


def test_single_rank_stub_line_has_the_contract_keys():
    out = last_json(run_bench("--stub", "--images", "4", "--distinct", "2", "--steps", "2", "--warmup", "1", "--width", "64", "--height", "48"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out
    assert out["n_gpus"] == 1 and out["roofline"]["bound"] == "hbm" and "workload" in out["config"]


def test_world_size_mismatch_fails_loudly():
    p = run_bench("--gpus", "2", "--stub", env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and b"--gpus 2" in p.stderr


def test_more_gpus_than_devices_fails_loudly():
    p = run_bench("--gpus", "64", "--steps", "1")          # no box has 64 devices: the launcher must refuse before spawning
    assert p.returncode != 0 and b"HIP device" in p.stderr
