"""GPU: the REAL N-rank path of bench.py on a one-GPU box.  `--share-device` (test only) puts every rank on device 0 and swaps RCCL for gloo (RCCL wants
a device per rank); everything else is the path the driver's `python bench.py --gpus N` takes: the launcher spawns the ranks, every rank builds its
shard, decodes it through the C ABI, checks EVERY image of it against the oracle, and the job scalars are reduced.  What is left for the first run on
eight devices to find is RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_two_ranks_on_one_device_decode_the_same_job_as_one_rank():
    job = ("--strong", "--job-images", "24", "--distinct", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--cpu-seconds", "0")
    p1, one = _run(*job)
    assert p1.returncode == 0 and one, p1.stderr.decode()[-1500:]
    p2, two = _run("--gpus", "2", "--share-device", *job)
    assert p2.returncode == 0 and two, p2.stderr.decode()[-1500:]
    for out, w in ((one, 1), (two, 2)):
        assert out["n_gpus"] == w and out["scaling"] == "strong" and out["bit_exact"] and out["parity_errors"] == 0 and out["value"] > 0
        assert out["shards"]["union_is_the_job"] and sum(out["shards"]["images"]) == 24 and len(out["shards"]["images"]) == w
        assert out["images_oracle_checked_per_rank"] >= 24 // w - 12 and out["images_oracle_checked_per_rank"] > 0        # every image of rank 0's shard was compared
    assert one["job_checksum"] == two["job_checksum"] and int(one["job_checksum"], 16) != 0                         # the same job whatever the partition
    assert len(two["per_rank_ms_per_step"]) == 2 and all(ms > 0 for ms in two["per_rank_ms_per_step"])
    assert "share-device" in two["data"]                                                                              # ... and the line says what it is


def test_without_the_switch_the_launcher_refuses_when_devices_are_missing():
    import jpegsnoop_amd as J
    have = int(J.load(require_device=False).jsnoop_device_count())
    if have >= 2:
        pytest.skip("two devices are visible here: the launcher would run the job")
    p, out = _run("--gpus", "2", "--strong", "--job-images", "8", "--steps", "1")
    assert p.returncode != 0 and out is None
    assert (b"bench.py: --gpus 2 requested but only %d HIP device(s) are visible" % have) in p.stderr
