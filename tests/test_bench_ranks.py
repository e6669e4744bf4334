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
    # the job checksum: sum mod 2^64 over ALL images of the job of mix64(image checksum ^ mix64(global index)) -- position-keyed, so
    # replicas cannot cancel; the stub's image checksum is a function of the global index alone
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    gidx = list(range(2 * images))
    sums = (np.array(gidx, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)).astype(np.uint64)
    want = bench.shard_checksum(sums, gidx)
    assert int(out["job_checksum"], 16) == want and want != 0


def test_shard_checksum_does_not_cancel_on_replicas():
    sys.path.insert(0, ROOT)
    import bench
    # 16 identical DIB checksums (a tiled workload): the XOR is 0, the position-keyed sum is not -- and it depends on every member
    sums = [0x1234567890ABCDEF] * 16
    a = bench.shard_checksum(sums, range(16))
    assert a != 0
    b = bench.shard_checksum(sums[:15] + [0x1234567890ABCDEE], range(16))
    assert a != b
    # partition independence: two shards' fingerprints add up to the job's
    assert (bench.shard_checksum(sums[:5], range(5)) + bench.shard_checksum(sums[5:], range(5, 16))) & bench.MIX64_MASK == a


def test_strong_scaling_job_is_partitioned_by_bytes_and_its_union_is_the_job():
    sys.path.insert(0, ROOT)
    import bench
    from jpegsnoop_amd.shard import partition_lpt
    nj, distinct = 203, 5
    one = last_json(run_bench("--stub", "--strong", "--job-images", str(nj), "--distinct", str(distinct), "--steps", "2", "--warmup", "1", "--stub-ms", "3"))
    two = last_json(run_bench("--gpus", "2", "--stub", "--strong", "--job-images", str(nj), "--distinct", str(distinct), "--steps", "2", "--warmup", "1", "--stub-ms", "3"))
    for out, w in ((one, 1), (two, 2)):
        assert out["scaling"] == "strong" and out["n_gpus"] == w and out["bit_exact"]
        sh = out["shards"]
        assert sh["union_is_the_job"] and sum(sh["images"]) == nj and len(sh["images"]) == w
    # the same job whatever the partition: same fingerprint, same pixel count per step
    assert one["job_checksum"] == two["job_checksum"] and int(one["job_checksum"], 16) != 0
    plan = bench.job_plan(nj, distinct)
    px = sum(bench.JOB_KINDS[k][0] * bench.JOB_KINDS[k][1] for k, _ in plan)
    for out in (one, two):
        assert abs(out["value"] - px * 2 / (out["ms_per_step"] * 2 * 1e-3) / 1e6) < 0.05 * out["value"] + 0.2
    # the shards are the LPT bins of the compressed-size list, and they are balanced to within one (largest) file
    costs = [bench.stub_cost(k, sd) for k, sd in plan]
    bins = partition_lpt(costs, 2)
    assert two["shards"]["images"] == [len(b) for b in bins]
    assert two["shards"]["compressed_bytes"] == [sum(costs[i] for i in b) for b in bins]
    assert max(two["shards"]["compressed_bytes"]) - min(two["shards"]["compressed_bytes"]) <= max(costs)
    assert {bench.JOB_KINDS[k] for k, _ in plan} == {(1280, 720), (1920, 1080), (3840, 2160)}


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
    assert p.returncode != 0
    import re
    m = re.search(rb"bench\.py: --gpus 64 requested but only (\d+) HIP device\(s\) are visible", p.stderr)
    assert m and int(m.group(1)) < 64, p.stderr[-500:]
    assert b"torch.distributed" not in p.stderr                # refused before a single rank was spawned


def test_strong_job_at_world_size_2_is_refused_the_same_way_without_devices():
    """`bench.py --gpus 2 --strong` (no --stub) on a box with fewer than two devices: the same refusal, the same text."""
    sys.path.insert(0, ROOT)
    import jpegsnoop_amd as J
    have = int(J.load(require_device=False).jsnoop_device_count())
    if have >= 2:
        import pytest
        pytest.skip("two devices are visible here: the launcher would run the job")
    p = run_bench("--gpus", "2", "--strong", "--steps", "1")
    assert p.returncode != 0 and (b"bench.py: --gpus 2 requested but only %d HIP device(s) are visible" % have) in p.stderr


def test_gpus_8_end_to_end_weak_and_strong():
    """The driver's `python bench.py --gpus 8` path end to end on the CPU (--stub: gloo in RCCL's place, a stand-in batch): eight ranks are spawned,
    the weak job gives every rank its shard, the strong job is ONE mixed job LPT-partitioned by compressed bytes (files are independent,
    source/JfifDecode.cpp:7306-7308), both reduce to one JSON line with eight per-rank times -- what is left for the first real run is RCCL itself."""
    sys.path.insert(0, ROOT)
    import bench
    from jpegsnoop_amd.shard import partition_lpt
    images, steps, w, h = 4, 2, 64, 48
    weak = last_json(run_bench("--gpus", "8", "--stub", "--stub-ms", "3", "--images", str(images), "--distinct", "2", "--steps", str(steps), "--warmup", "1",
                               "--width", str(w), "--height", str(h)))
    assert weak["n_gpus"] == 8 and weak["scaling"] == "weak" and weak["config"]["parallelism"] == "shard8" and weak["steps"] == steps
    assert len(weak["per_rank_ms_per_step"]) == 8 and all(ms >= 3.0 for ms in weak["per_rank_ms_per_step"])
    assert weak["bit_exact"] and weak["parity_errors"] == 0
    assert weak["ms_per_step"] >= max(weak["per_rank_ms_per_step"]) - 1e-3
    assert abs(weak["value"] - 8 * images * w * h * steps / (weak["ms_per_step"] * steps * 1e-3) / 1e6) < 0.05 * weak["value"] + 0.2
    nj, distinct = 131, 5
    strong = last_json(run_bench("--gpus", "8", "--stub", "--strong", "--job-images", str(nj), "--distinct", str(distinct), "--steps", "2", "--warmup", "1", "--stub-ms", "3"))
    one = last_json(run_bench("--stub", "--strong", "--job-images", str(nj), "--distinct", str(distinct), "--steps", "2", "--warmup", "1", "--stub-ms", "3"))
    sh = strong["shards"]
    assert strong["n_gpus"] == 8 and strong["scaling"] == "strong" and strong["bit_exact"] and len(strong["per_rank_ms_per_step"]) == 8
    assert sh["union_is_the_job"] and sum(sh["images"]) == nj and len(sh["images"]) == 8
    assert strong["job_checksum"] == one["job_checksum"] and int(one["job_checksum"], 16) != 0
    plan = bench.job_plan(nj, distinct)
    bins = partition_lpt([bench.stub_cost(k, sd) for k, sd in plan], 8)
    assert sh["images"] == [len(b) for b in bins]
