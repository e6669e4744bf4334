#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the MI355X scan-decode path on BASELINE.json's headline workload.

Workload (N = 1): BASELINE config 3, a batch of 1024 baseline 4:2:0 1920x1080 JPEGs per GPU
(`--distinct` different synthetic pictures, physically replicated to `--images` files, so HBM
holds -- and the timed kernels read -- all 0.6 GB of compressed bytes), compressed bytes already
resident in HBM when the timed region starts, DIBs left in HBM (timing scope T1 of SURVEY.md 8d).
N > 1: one rank per GPU, every rank owns its own 1024-image shard (weak scaling, BASELINE
config 4 at N = 8); no data-path collective, RCCL only for the barrier and the all-reduce /
all-gather of the job scalars.  `python bench.py --gpus N` without RANK in the environment
spawns the N ranks itself (torch.distributed.run); under torch.distributed.run it is a rank.

A "step" is one decode of the whole resident batch: unstuff -> sub-sequence synchronisation ->
block scan -> coefficient write -> DC scan -> IDCT + colour -> DIB.  Prints ONE JSON line
(rank 0).  `value` counts SOF pixels (X*Y) of every image of every rank per second, and is only
reported when every DIB checksum equals the oracle's (bit-exact gate).

`--strong` (opt-in; the default line stays the weak-scaling one): ONE job of `--job-images` files of mixed size (1280x720,
1920x1080, 3840x2160) -- the loop of CJPEGsnoopCore::DoBatchFileProcess (source/JPEGsnoopCore.cpp:765-845) made parallel: every
rank derives the same job list, `partition_lpt` balances it by compressed bytes, each rank decodes its shard, `value` is the
job's pixels over the slowest rank's time ("scaling": "strong"), the per-rank times and shard costs are reported beside it.

`--stub` replaces the GPU batch by a CPU stand-in (gloo instead of RCCL): the rank / shard /
reduce logic of this file runs unchanged, which is what tests/test_bench_ranks.py drives at
world size 2.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "Mpixels/sec decoded (baseline 4:2:0 JPEG) at 1/2/4/8 GPUs; bit-exact vs ref"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=1024, help="images per GPU")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic pictures per GPU (replicated to --images)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-extras", action="store_true", help="skip the BASELINE config 2 / config 5 / staging-pipeline extras (rank 0, N=1)")
    ap.add_argument("--no-split", action="store_true", help="decode the batch on ONE stream (profiling runs: keeps the per-kernel averages those of whole-batch launches); "
                    "default: the library's own choice, two halves on two streams for a batch this size")
    ap.add_argument("--strong", action="store_true", help="strong scaling: one job of --job-images mixed-size files, LPT-partitioned over the ranks")
    ap.add_argument("--job-images", type=int, default=8192, help="--strong: files in the whole job (BASELINE config 4 names 8192)")
    ap.add_argument("--stub", action="store_true", help="CPU dry run of the rank logic: stand-in batch, gloo backend (tests)")
    ap.add_argument("--stub-ms", type=float, default=2.0, help="--stub: pretended decode time per step")
    ap.add_argument("--share-device", action="store_true", help="TEST ONLY: all ranks decode on device 0 (gloo instead of RCCL: RCCL wants one device per rank) -- runs the real "
                    "N-rank path on a one-GPU box; its value is meaningless as a scaling figure and the line says so")
    return ap.parse_args(argv)


def visible_gpus():
    import jpegsnoop_amd as J
    return int(J.load(require_device=False).jsnoop_device_count())


def spawn_ranks(args, argv):
    """`bench.py --gpus N` outside torch.distributed.run: become the launcher of N ranks (one per GPU)."""
    if not args.stub:
        have = visible_gpus()
        if have < (1 if args.share_device else args.gpus):
            sys.exit(f"bench.py: --gpus {args.gpus} requested but only {have} HIP device(s) are visible")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


class StubBatch:
    """CPU stand-in for jpegsnoop_amd.JpegBatch (--stub): same surface bench.py uses, no device.  `dims` = (width, height) per image,
    `gidx` = the global index of each image in the job (the pretended DIB checksum depends on it alone: whatever the partition,
    the job's fingerprint is the same)."""

    def __init__(self, dims, gidx, step_ms):
        import numpy as np
        self.dims, self.step_ms = list(dims), step_ms
        self._sums = (np.array(list(gidx), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)).astype(np.uint64)
    def __len__(self): return len(self.dims)
    def pixels(self): return sum(w * h for w, h in self.dims)
    def algorithmic_bytes(self): return sum((w * ((h + 15) // 16 * 16)) * 4 + int(0.285 * w * h) for w, h in self.dims)
    def upload(self): pass
    def decode(self): time.sleep(self.step_ms * 1e-3)
    def sync(self): pass
    def dib_checksums(self): return self._sums.copy()
    def info(self, i): return {"flags": 0, "path": 1}
    def decode_timed(self, reps):
        st = {"clear": 0.0, "unstuff": 0.1, "sync": 0.3, "blockscan": 0.0, "write": 0.5, "dcscan": 0.0, "exact": 0.0, "idct_color": 1.0}
        k = self.step_ms / sum(st.values())
        return self.step_ms, {a: b * k for a, b in st.items()}
    def close(self): pass


MIX64_MASK = 0xFFFFFFFFFFFFFFFF


def mix64(z):
    """splitmix64 finaliser (the same mixer as k_dib_checksum)."""
    z = (z + 0x9E3779B97F4A7C15) & MIX64_MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MIX64_MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MIX64_MASK
    return z ^ (z >> 31)


def shard_checksum(sums, global_index):
    """Fingerprint of a shard: sum mod 2^64 of mix64(dib_checksum_i ^ mix64(global index of image i)).  Position-keyed, so the
    replicas of a tiled workload do not cancel (a plain XOR of 16 identical values is 0) and the all-reduced sum over the ranks
    is the fingerprint of the whole job whatever the partition."""
    tot = 0
    for s, g in zip(sums, global_index):
        tot = (tot + mix64(int(s) ^ mix64(int(g)))) & MIX64_MASK
    return tot


# --strong: the job list.  Entry i = (kind, seed): kind indexes JOB_KINDS, 5/8 of the files are 1080p, 2/8 720p, 1/8 4K.
JOB_KINDS = [(1280, 720), (1920, 1080), (3840, 2160)]
JOB_PATTERN = [1, 1, 0, 1, 1, 2, 1, 0]


def job_plan(job_images, distinct):
    return [(JOB_PATTERN[i % 8], (i // 8) % max(1, distinct)) for i in range(job_images)]


def job_seed(kind, seed):
    return 50000 + 1000 * kind + seed + 1


def stub_cost(kind, seed):
    """--stub: a stand-in for the compressed size of file (kind, seed): proportional to its pixels, +-12 % by seed."""
    w, h = JOB_KINDS[kind]
    return int(w * h * 0.285 * (0.88 + 0.24 * ((seed * 2654435761) % 97) / 96.0))


def _cpu_worker_init(width, height, seed_base):
    """Worker process of the all-cores cpu_baseline figure: own oracle instance, own synthetic image (no data crosses processes)."""
    from oracle import harness as H
    global _ORC, _IMG, _H
    _H = H
    _ORC = H.oracle_backend()
    _IMG = H.synth_jpeg(width=width, height=height, hs=2, vs=2, quality=85, seed=seed_base + os.getpid() % 64 + 1)
    H.drive(_ORC, _IMG)                                           # warm-up


def _cpu_worker(reps):
    t, c = time.perf_counter(), time.process_time()
    for _ in range(reps):
        _H.drive(_ORC, _IMG)
    return reps, time.perf_counter() - t, time.process_time() - c


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_all_cores(args, rank):
    """The same CPU path on every host core: independent decoder instances, one per process (the reference is single-threaded,
    DoBatchFileProcess run N-wide is N instances) -- bounded to a few seconds."""
    try:
        import concurrent.futures as cf
        import multiprocessing as mp
        ncore = len(os.sched_getaffinity(0))
        try:                                                               # a container's CPU quota, not the visible core count, is what it can use
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                ncore = max(1, min(ncore, int(q) // int(per)))
        except (OSError, ValueError):
            pass
        if ncore <= 1:
            return None
        reps = 16
        with cf.ProcessPoolExecutor(max_workers=ncore, mp_context=mp.get_context("spawn"), initializer=_cpu_worker_init,
                                    initargs=(args.width, args.height, 1000 * rank)) as ex:
            list(ex.map(_cpu_worker, [0] * ncore))                         # every worker is up (library loaded, image made, one decode done)
            t1 = time.perf_counter(); res = list(ex.map(_cpu_worker, [reps] * ncore)); dt = time.perf_counter() - t1
        n_done = sum(r[0] for r in res)
        return {"value": round(n_done * args.width * args.height / dt / 1e6, 1), "unit": "Mpixels/s", "cores": ncore,
                "cpu_seconds_per_wall_second": round(sum(r[2] for r in res) / dt, 1),
                "note": "one oracle instance per usable core (affinity capped by the cgroup CPU quota; separate processes), %d decodes in %.2f s wall; "
                        "cpu_seconds_per_wall_second is the CPU time the box actually granted" % (n_done, dt)}
    except Exception as e:                                       # a baseline figure, never a reason to fail the bench
        return {"error": str(e)}


def staging_pipeline(J, files, images):
    """Timing scopes T1 / T2 / T3 from one run of the overlapped staging pipeline (two slots, each holding the whole workload):
    H2D of batch k+1 -- and for T3 the D2H of batch k-1's DIBs -- while batch k decodes."""
    pipe = J.JpegPipeline(2)
    try:
        for b in pipe.slots:
            for f in files:
                b.add_jpeg(f)
            b.tile(images)
        t2 = pipe.run(8, d2h=False)
        t3 = pipe.run(2, d2h=True)
        px = pipe.slots[0].pixels()
        bound2 = max(t2["decode_ms"], t2["h2d_ms"])
        return {"slots": 2, "T1_decode_ms": round(t2["decode_ms"], 3), "h2d_ms": round(t2["h2d_ms"], 3), "h2d_GBps": round(t2["compressed_bytes"] / t2["h2d_ms"] / 1e6, 1),
                "T2_ms_per_batch_overlapped": round(t2["ms_per_batch"], 3), "T2_mpix_per_s": round(px / t2["ms_per_batch"] / 1e3, 1),
                "T2_over_max_T1_h2d": round(t2["ms_per_batch"] / bound2, 3),
                "d2h_ms": round(t3["d2h_ms"], 2), "d2h_GBps": round(t3["dib_bytes"] / t3["d2h_ms"] / 1e6, 1),
                "T3_ms_per_batch_overlapped": round(t3["ms_per_batch"], 2), "T3_mpix_per_s": round(px / t3["ms_per_batch"] / 1e3, 1),
                "compressed_bytes": t2["compressed_bytes"], "dib_bytes": t3["dib_bytes"],
                "note": "8 batches for T2, 2 for T3 (PCIe D2H of 8.9 GB of DIBs per batch bounds T3); reported beside, never as, value"}
    finally:
        pipe.close()


def _plain_ms(b, reps=20):
    """ms per decode of `reps` back-to-back decodes of a resident batch, wall clock around a device synchronise -- the headline's own way of timing (decode_timed puts
    a hipEvent behind every stage: that is where stages_ms comes from, and it costs a job of a few hundred microseconds 5-10 %)."""
    import torch
    b.decode(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        b.decode()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / reps


def extras_single_gpu(J, H, orc, np):
    """BASELINE configs 1, 2 and 5 beside the headline (rank 0, N = 1): small, each < 1 s of GPU time."""
    extra = {}
    # config 2: one 3840x2160 4:2:0 image end to end on the device
    one = J.JpegBatch()
    f4k = H.synth_jpeg(width=3840, height=2160, hs=2, vs=2, quality=85, seed=77)
    one.add_jpeg(f4k); one.upload(); one.decode(); one.sync()
    ms1t, st1 = one.decode_timed(10)
    ms1 = _plain_ms(one, 50)
    H.drive(orc, f4k)
    alg1 = one.algorithmic_bytes()
    extra["config2_single_3840x2160"] = {"ms": round(ms1, 4), "mpix_per_s": round(3840 * 2160 / ms1 / 1e3, 1),
                                          "bit_exact": bool(int(one.dib_checksums()[0]) == J.dib_checksum_numpy(orc.dib())),
                                          "roofline_frac": round(alg1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "algorithmic_bytes": alg1,
                                          "ms_with_stage_events": round(ms1t, 4), "stages_ms": {k: round(v, 4) for k, v in st1.items()}}
    one.close()
    # small jobs between config 2 and config 3: N distinct 1920x1080 images resident in HBM, ms per decode (what a caller that cannot batch a
    # thousand files sees; jobs this size synchronise by candidates), every DIB checked against the oracle
    small = {}
    fs = [H.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=100 + i) for i in range(16)]
    ws = []
    for f in fs:
        H.drive(orc, f); ws.append(J.dib_checksum_numpy(orc.dib()))
    for nsm in (1, 4, 16, 64, 128):                               # (64 and 128: the sixteen pictures four / eight times over, every one with its own bytes)
        sb = J.JpegBatch()
        for f in fs[:min(nsm, 16)]:
            sb.add_jpeg(f)
        if nsm > 16:
            sb.tile(nsm)
        sb.upload(); sb.decode(); sb.sync()
        msst, sts = sb.decode_timed(10)
        mss = _plain_ms(sb, 20)
        small[str(nsm)] = {"ms": round(mss, 4), "mpix_per_s": round(nsm * 1920 * 1080 / mss / 1e3, 1), "bit_exact": bool(all(int(a) == ws[i % 16] for i, a in enumerate(sb.dib_checksums()))),
                           "ms_with_stage_events": round(msst, 4), "sync_ms": round(sts["sync"], 4)}
        sb.close()
    extra["small_jobs_1080p"] = small
    # config 5: progressive multi-scan 4:2:2 with RSTn every MCU row; parity is transitive (same coefficients as the baseline encoding)
    kw5 = dict(width=1920, height=1080, hs=2, vs=1, restart_interval=120, quality=85, seed=55)
    base5, prog5 = H.synth_jpeg(progressive=0, **kw5), H.synth_jpeg(progressive=2, **kw5)
    dec = J.CimgDecode()
    nsc = dec.DecodeProgressive(prog5)
    H.drive(orc, base5)
    ok5 = bool(np.array_equal(dec.GetBitmapPtr(), orc.dib()))
    t5 = time.perf_counter()
    for _ in range(10):
        dec.DecodeProgressive(prog5)
    ms5 = (time.perf_counter() - t5) * 100.0
    extra["config5_progressive_1920x1080_422_rst"] = {"scans": nsc, "ms_end_to_end": round(ms5, 3), "mpix_per_s": round(1920 * 1080 / ms5 / 1e3, 1),
                                                      "bit_exact_vs_baseline_encoding": ok5,
                                                      "note": "single file, per call: host parse + H2D + scan launches (one wave per restart interval) + back end"}
    dec.close()
    # ... and as a batch citizen: N such files, every scan of every image decoded together (one launch per dependency level).  64 files
    # decode one restart interval per wave; from ~100 files on the library switches to one interval per LANE (k_prog_scan_lanes)
    want5 = J.dib_checksum_numpy(orc.dib())
    for nb5 in (64, 512):
        p5 = J.JpegBatch(); p5.add_jpeg(prog5); p5.tile(nb5); p5.upload(); p5.decode(); p5.sync()
        ok5b = bool(all(int(s) == want5 for s in p5.dib_checksums()))
        ms5b, st5b = p5.decode_timed(3)
        extra["config5_progressive_batch%d" % nb5] = {"ms_per_batch": round(ms5b, 3), "ms_per_image": round(ms5b / nb5, 4), "mpix_per_s": round(nb5 * 1920 * 1080 / ms5b / 1e3, 1),
                                                      "bit_exact_vs_baseline_encoding": ok5b, "speedup_vs_single_call": round(ms5 / (ms5b / nb5), 1),
                                                      "stages_ms": {"scans": round(st5b["write"], 3), "finalize": round(st5b["dcscan"], 3), "idct_color": round(st5b["idct_color"], 3)}}
        p5.close()
    # ... and as most progressive files in the wild come: no restart markers at all (libjpeg-turbo's file, tests/golden/pillow/: the baseline
    # encoding of the same picture is the oracle's input).  Every scan is ONE interval: the scan kinds decode on one lane / one wave each.
    try:
        pdir = os.path.join(ROOT, "tests", "golden", "pillow")
        pprog = open(os.path.join(pdir, "p422_nodri_1920x1080_prog.jpg"), "rb").read(); pbase = open(os.path.join(pdir, "p422_nodri_1920x1080_base.jpg"), "rb").read()
        H.drive(orc, pbase)
        decn = J.CimgDecode()
        nscn = decn.DecodeProgressive(pprog)
        okn = bool(np.array_equal(decn.GetBitmapPtr()[-1080:, :1920], orc.dib()[-1080:, :1920]))
        tn0 = time.perf_counter(); decn.DecodeProgressive(pprog); msn = (time.perf_counter() - tn0) * 1e3
        extra["config5_progressive_no_rst"] = {"scans": nscn, "ms_end_to_end": round(msn, 2), "mpix_per_s": round(1920 * 1080 / msn / 1e3, 2), "bit_exact_vs_libjpeg_baseline_encoding": okn,
                                               "note": "libjpeg-turbo progressive 1920x1080 4:2:2 without DRI: restart intervals are the parallel grain of the progressive path, a scan without markers is one interval"}
        wantn = J.dib_checksum_numpy(decn.GetBitmapPtr())
        decn.close()
        # ... the form in which such files get their throughput: many of them in one batch, a scan of every file per launch
        for nbn in (64, 512):
            pn = J.JpegBatch(); pn.add_jpeg(pprog); pn.tile(nbn); pn.upload(); pn.decode(); pn.sync()
            okb = bool(all(int(x) == wantn for x in pn.dib_checksums()))
            msnb, _ = pn.decode_timed(1)
            extra["config5_progressive_no_rst_batch%d" % nbn] = {"ms_per_batch": round(msnb, 2), "ms_per_image": round(msnb / nbn, 3), "mpix_per_s": round(nbn * 1920 * 1080 / msnb / 1e3, 1),
                                                                 "same_pixels_as_the_single_call": okb}
            pn.close()
    except Exception as e:
        extra["config5_progressive_no_rst"] = {"error": repr(e)}
    # damaged files (JPEGsnoop's daily input): one 1080p file each, decode + sync of a resident batch of one, DIB checked against the oracle.
    # "flip": the scan bit flips of tools/corrupt_timing.py that leave a trace (coefficient-index overflow); "cut": truncated at half its
    # scan (the rest decodes as the zero bytes CwindowBuf::Buf returns past the end); "marker": two bytes overwritten by a stray marker
    dmg = {}
    gcall = None
    for label, kwd in (("1080p", dict(width=1920, height=1080)), ("1080p_rst", dict(width=1920, height=1080, restart_interval=120))):
        based = H.synth_jpeg(seed=9, **kwd)
        pd = H.parse_jpeg(based)
        kinds = [("flip", 0.95), ("cut", 0.5), ("marker", 0.3)]
        # the cases the parallel path hands to the sequential mirror for most of the file: 256 random bytes (FF among them) at 30 % of a scan
        # without restart markers; a restart marker that falls INSIDE a block early in the file (two bytes deleted in front of the second RSTn)
        # ... and round 4's worst class (0.7-2.4 s then): behind a marker met inside a block, the value bits of a symbol run past the end of a later restart
        # interval -- the reference's register over-reads and its decode of the image is over (source/ImgDecode.cpp:1229-1282, :3623-3625)
        kinds += [("garbage", 0.3)] if label == "1080p" else [("rst_in_block", 0.0), ("overrun", 0.0)]
        for kind, frac in kinds:
            d = bytearray(based)
            i = pd.scan_start + int((pd.scan_end - pd.scan_start) * frac)
            if kind == "overrun":
                j = bytes(d).index(b"\xff\xd1", pd.scan_start)
                del d[j - 2:j]
                marks = [k for k in range(pd.scan_start, len(d) - 1) if d[k] == 0xFF and 0xD0 <= d[k + 1] <= 0xD7]
                found = None
                probe = J.JpegBatch()
                for which in (len(marks) // 3, len(marks) // 2, 2 * len(marks) // 3):
                    for cut in range(1, 7):
                        e = bytearray(d); k = marks[which]
                        if 0xFF in e[k - cut - 1:k]:
                            continue
                        del e[k - cut:k]
                        probe.clear(); probe.add_jpeg(bytes(e)); probe.upload(); probe.decode(); probe.sync()
                        if probe.info(0)["flags"] & 0x0002:
                            found = e
                            break
                    if found is not None:
                        break
                probe.close()
                if found is None:
                    continue
                d = found
            elif kind == "garbage":
                d[i:i + 256] = np.random.RandomState(5).randint(0, 256, 256).astype(np.uint8).tobytes()
            elif kind == "rst_in_block":
                j = bytes(d).index(b"\xff\xd1", pd.scan_start)
                del d[j - 2:j]
            elif kind == "flip":
                while d[i] == 0xFF or d[i - 1] == 0xFF or (d[i] ^ 0x10) == 0xFF:
                    i += 1
                d[i] ^= 0x10
            elif kind == "cut":
                d = d[:i]
            else:
                d[i:i + 2] = b"\xff\xe3"
            d = bytes(d)
            bd = J.JpegBatch(); bd.add_jpeg(d); bd.upload(); bd.decode(); bd.sync()
            t0 = time.perf_counter()
            reps = 0
            while reps < 3 and (reps == 0 or time.perf_counter() - t0 < 2.0):       # (a case on the slow path takes seconds: one repetition)
                bd.decode(); bd.sync(); reps += 1
            msd = (time.perf_counter() - t0) / reps * 1e3
            H.drive(orc, d)
            inf = bd.info(0)
            dmg["%s_%s" % (label, kind)] = {"ms": round(msd, 3), "path": int(inf["path"]), "flags": "0x%04x" % inf["flags"],
                                            "bit_exact": bool(int(bd.dib_checksums()[0]) == J.dib_checksum_numpy(orc.dib()))}
            bd.close()
            # ... and the call a CjfifDecode makes (source/JfifDecode.cpp:5299): DecodeScanImg with a log callback -- upload, decode, side outputs,
            # messages and report of this one file, wall clock; side outputs and status words against the oracle's
            try:
                if gcall is None:
                    gcall = H.Backend(J.load(), "jsnoop_", "hip")
                qd = H.parse_jpeg(d)                                 # (the header walk is the caller's, in Python here: not part of the call)
                H.drive(gcall, d, qd, quiet=0)
                tc = time.perf_counter(); H.drive(gcall, d, qd, quiet=0); call_ms = (time.perf_counter() - tc) * 1e3
                import importlib.util as _iu
                _fu = _iu.spec_from_file_location("fuzz_util", os.path.join(ROOT, "tests", "fuzz_util.py")); _fm = _iu.module_from_spec(_fu); _fu.loader.exec_module(_fm)
                dmg["%s_%s" % (label, kind)].update({"call_ms": round(call_ms, 3), "side_mode": int(gcall.lib.jsnoop_last_side_mode(gcall.h)), "log_lines": len(gcall.log_lines()),
                                                     "side_outputs_equal_oracle": _fm.differs(orc, gcall) is None})
            except Exception as e:
                dmg["%s_%s" % (label, kind)]["call_error"] = repr(e)
    # ... the same call on CLEAN files (what JPEGsnoop does with one file: decode + side outputs + messages + report through the drop-in API, log callback installed):
    # ms per call, the file in pageable host memory, the header walk not included; side outputs and status words against the oracle's
    calls = {}
    try:
        import ctypes as _C, importlib.util as _iu
        _fu = _iu.spec_from_file_location("fuzz_util", os.path.join(ROOT, "tests", "fuzz_util.py")); _fm = _iu.module_from_spec(_fu); _fu.loader.exec_module(_fm)
        if gcall is None:
            gcall = H.Backend(J.load(), "jsnoop_", "hip")
        for name, kwc in (("640x480_444", dict(width=640, height=480, hs=1, vs=1, seed=3)), ("1080p_420", dict(width=1920, height=1080, hs=2, vs=2, seed=100)),
                          ("2160p_420", dict(width=3840, height=2160, hs=2, vs=2, seed=77)), ("1080p_422_rst", dict(width=1920, height=1080, hs=2, vs=1, restart_interval=120, seed=55))):
            fc = H.synth_jpeg(quality=85, **kwc)
            pc = H.drive(gcall, fc, quiet=0)
            H.drive(orc, fc)
            okc = _fm.differs(orc, gcall) is None
            buf = (_C.c_uint8 * len(fc)).from_buffer_copy(fc)
            for _ in range(3):
                gcall.decode_scan_img(_C.cast(buf, _C.c_void_p), len(fc), pc.scan_start, 1, 1)
            nrep = 30; tc = time.perf_counter()
            for _ in range(nrep):
                gcall.decode_scan_img(_C.cast(buf, _C.c_void_p), len(fc), pc.scan_start, 1, 1)
            calls[name] = {"call_ms": round((time.perf_counter() - tc) / nrep * 1e3, 3), "side_outputs_equal_oracle": okc}
    except Exception as e:
        calls["error"] = repr(e)
    extra["drop_in_call_clean_files"] = calls
    if gcall is not None:
        gcall.close()
    extra["damaged_files"] = dmg
    # the per-image results of a batch (side outputs of every image behind one decode: what DoBatchFileProcess does per file): ms per image
    try:
        nbr = 128
        fsb = [H.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=100 + i) for i in range(16)]
        bb = J.JpegBatch(want_planes=True)
        for f in fsb:
            bb.add_jpeg(f)
        bb.tile(nbr); bb.upload(); bb.decode(); bb.sync()
        tb = time.perf_counter()
        for i in range(nbr):
            bb.side_outputs(i)
        extra["batch_side_outputs_128x1080p"] = {"ms_per_image": round((time.perf_counter() - tb) * 1e3 / nbr, 3), "note": "jsnoop_batch_side_outputs of every image behind one decode, Python call included"}
        bb.close()
    except Exception as e:
        extra["batch_side_outputs_128x1080p"] = {"error": repr(e)}
    # the baseline (SOF0) form of config 5 through the batch path, oracle-checked directly
    b5 = J.JpegBatch(); b5.add_jpeg(base5); b5.upload(); b5.decode(); b5.sync()
    msb, _ = b5.decode_timed(10)
    extra["config5_baseline_form_422_rst"] = {"ms": round(msb, 4), "bit_exact": bool(int(b5.dib_checksums()[0]) == want5)}     # (want5: the oracle's DIB of this file, taken above -- the oracle has decoded the damaged files since)
    b5.close()
    return extra


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args, argv))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")

    import numpy as np
    import torch
    import torch.distributed as dist
    import jpegsnoop_amd as J

    stub = args.stub
    dev = None
    if not stub:
        import __graft_entry__ as G
        from oracle import harness as H            # checker + input generator only (cpu_baseline leg, parity gate)
        if not (os.path.exists(J.LIB_PATH) and os.path.exists(H.ORC_SO) and os.path.exists(H.SYNTH_SO)):
            if local_rank == 0:
                G.build()
            else:                                        # other ranks wait for rank 0's build instead of racing it
                for _ in range(600):
                    if os.path.exists(J.LIB_PATH) and os.path.exists(H.ORC_SO) and os.path.exists(H.SYNTH_SO):
                        break
                    time.sleep(0.5)
                time.sleep(1.0)
        dev_index = 0 if args.share_device else local_rank
        if torch.cuda.device_count() <= dev_index:
            sys.exit(f"bench.py: rank {rank} needs GPU {dev_index}, {torch.cuda.device_count()} visible")
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    if world > 1:
        if stub or args.share_device:
            dist.init_process_group("gloo")
            dev = None if stub else dev                     # (gloo reduces CPU tensors: `rdev` below)
        else:
            dist.init_process_group("nccl", device_id=dev)
        dist.barrier()
    rdev = dev if (world > 1 and not stub and not args.share_device) else None      # where the job scalars are reduced: the GPU under RCCL, the host under gloo

    def device_sync():
        if not stub:
            torch.cuda.synchronize()

    # ---- this rank's shard ----------------------------------------------------------------------------
    #  weak (default): every rank owns --images pictures of its own, seeded per (rank, index): global index = rank * images + i
    #  strong:         ONE job list, the same on every rank; partition_lpt by compressed bytes; this rank takes bin `rank`
    t_gen = t_upload_first = t_upload = 0.0
    files = []                    # weak: the rank's distinct files
    job_files = {}                # strong: (kind, seed) -> file bytes, every distinct file of the job
    keys = []                     # per image of this rank's batch: what it is a copy of (weak: index into files; strong: (kind, seed))
    shard_info = None
    if args.strong:
        plan = job_plan(args.job_images, args.distinct)
        distinct_keys = sorted(set(plan))
        if stub:
            costs = [stub_cost(k, sd) for k, sd in plan]
        else:
            t0 = time.time()
            for k, sd in distinct_keys:                                   # deterministic: every rank makes the same files, and so the same costs
                w, h = JOB_KINDS[k]
                job_files[(k, sd)] = H.synth_jpeg(width=w, height=h, hs=2, vs=2, quality=85, seed=job_seed(k, sd))
            t_gen = time.time() - t0
            costs = [len(job_files[e]) for e in plan]
        bins = J.partition_lpt(costs, world)
        gidx = bins[rank]
        keys = [plan[g] for g in gidx]
        shard_info = {"images": len(gidx), "compressed_bytes": int(sum(costs[g] for g in gidx)),
                      "index_sums": [len(gidx), int(sum(gidx)), int(sum(g * g for g in gidx))]}
    else:
        gidx = list(range(rank * args.images, (rank + 1) * args.images))
        keys = [i % args.distinct for i in range(args.images)]
    if stub:
        dims = [JOB_KINDS[k[0]] for k in keys] if args.strong else [(args.width, args.height)] * args.images
        batch = StubBatch(dims, gidx, args.stub_ms)
    else:
        lib = J.load()
        assert lib.jsnoop_set_device(0 if args.share_device else local_rank) == 0, J.last_error()
        batch = J.JpegBatch(want_planes=False)
        if args.strong:
            for k in keys:
                batch.add_jpeg(job_files[k])                 # every image has its own bytes in the raw arena
        else:
            t0 = time.time()
            files = [H.synth_jpeg(width=args.width, height=args.height, hs=2, vs=2, quality=85, seed=1000 * rank + i + 1)
                     for i in range(args.distinct)]
            t_gen = time.time() - t0
            for f in files:
                batch.add_jpeg(f)
            batch.tile(args.images)                          # physical replication: every image has its own bytes in the raw arena
        t0 = time.perf_counter()
        batch.upload()                                   # pinned host -> HBM, outside the timed region
        t_upload_first = time.perf_counter() - t0        # includes the one-time hipMalloc of the arenas
        lib.jsnoop_batch_set_options(batch._h, 1, 0, 0)  # marks the batch dirty: the next upload() repeats only the H2D copies
        t0 = time.perf_counter()
        batch.upload()                                   # returns after the copy stream has drained
        t_upload = time.perf_counter() - t0
    pixels = batch.pixels()
    alg_bytes = batch.algorithmic_bytes()

    # The decode form is the library's default (jsnoop_batch_set_split(0): two halves of a batch this size on two streams) unless --no-split.
    if not stub:
        batch.set_split(1 if args.no_split else 0)
    # ---- parity gate + cpu_baseline sample (the oracle is only the checker / the baseline here) ----
    batch.decode(); batch.sync()
    sums = batch.dib_checksums()
    flags = [batch.info(i)["flags"] for i in range(len(batch))]
    paths = [batch.info(i)["path"] for i in range(len(batch))]
    n_cpu, cpu_time, errors = 0, 0.0, 0
    budget = args.cpu_seconds if (rank == 0 and world == 1 and not stub and not args.strong) else 0.0
    all_cores = ref_info = cfg1 = None
    orc = None
    n_checked = 0
    if not stub:
        # EVERY rank checks EVERY distinct picture of its shard against the oracle (ranks work side by side; 64 x 1080p = ~9 s):
        # no image is reported bit-exact on the strength of its flags alone.  The cpu_baseline figure is the time of the first
        # `budget` seconds of these same oracle decodes (rank 0, N = 1).
        orc = H.oracle_backend()
        members = {}
        for i, k in enumerate(keys):
            members.setdefault(k, []).append(i)
        for k in sorted(members):
            t1 = time.perf_counter()
            H.drive(orc, job_files[k] if args.strong else files[k])
            dt = time.perf_counter() - t1
            want = J.dib_checksum_numpy(orc.dib())
            for i in members[k]:
                errors += int(int(sums[i]) != want)
                n_checked += 1
            if budget > 0 and cpu_time <= budget:
                n_cpu += 1
                cpu_time += dt
        assert n_checked == len(batch), "parity gate: every image of the shard must have been compared"
    errors += sum(1 for f in flags if f)
    if budget > 0:
        all_cores = cpu_all_cores(args, rank)
        if H.have_ref():                  # the compiled reference, when its .so travelled (never reads /root/reference)
            ref = H.ref_backend()
            nref = min(16, args.distinct)
            t1 = time.perf_counter()
            for j in range(nref):
                H.drive(ref, files[j])
            dt = time.perf_counter() - t1
            ref_info = {"value": round(nref * args.width * args.height / dt / 1e6, 2), "images": nref, "seconds": round(dt, 2)}
            f1 = H.synth_jpeg(width=640, height=480, hs=1, vs=1, quality=85, seed=11)     # BASELINE config 1: the reference's own CPU case
            t1 = time.perf_counter()
            for _ in range(20):
                H.drive(ref, f1)
            dref = (time.perf_counter() - t1) / 20
            t1 = time.perf_counter()
            for _ in range(20):
                H.drive(orc, f1)
            dorc = (time.perf_counter() - t1) / 20
            cfg1 = {"workload": "single 640x480 baseline 4:4:4 q85, CPU", "reference_ms": round(dref * 1e3, 2), "reference_mpix_per_s": round(640 * 480 / dref / 1e6, 2),
                    "port_ms": round(dorc * 1e3, 2), "port_mpix_per_s": round(640 * 480 / dorc / 1e6, 2)}
            ref.close()

    # ---- timed region -----------------------------------------------------------------------
    for _ in range(args.warmup):
        batch.decode()
    batch.sync()
    if world > 1:
        dist.barrier()
    device_sync()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        batch.decode()
    device_sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    batch.sync()
    flags2 = [batch.info(i)["flags"] for i in range(len(batch))]
    errors += sum(1 for f in flags2 if f)
    sums2 = batch.dib_checksums()
    errors += int((sums2 != sums).sum())

    # per-stage device time (hipEvents on the batch stream, same resident batch) -- taken in the ONE-stream form whatever form was timed above:
    # in a split decode every kernel shares the chip with a kernel of the other half, its launch duration describes the sharing, not the kernel
    decode_form = "stub" if stub else ("two halves on two streams (library default)" if batch.split_parts() == 2 else "one stream")
    if not stub:
        batch.set_split(1)
    ms_whole, stages = batch.decode_timed(max(3, min(10, args.steps)))
    dom = max(stages, key=stages.get)

    my_ck = shard_checksum(sums, gidx)
    tot_px, max_el, job_ck, tot_err = J.reduce_job_stats(pixels * args.steps, elapsed, my_ck, errors, rdev)
    if len(batch) and job_ck == 0:
        tot_err += 1                                      # a fingerprint of 0 carries no information (what the XOR of replicas used to give)
    per_rank_ms = [round(elapsed / args.steps * 1e3, 4)]
    if world > 1:
        t = torch.tensor([elapsed / args.steps * 1e3], dtype=torch.float64, device=rdev if rdev is not None else "cpu")
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        per_rank_ms = [round(float(g[0]), 4) for g in got]
    shards = None
    if args.strong:
        # the union of the shards must be the job: count, sum and sum of squares of the global indices, all-reduced, against the closed forms
        t = torch.tensor(shard_info["index_sums"] + [shard_info["compressed_bytes"]], dtype=torch.int64, device=rdev if rdev is not None else "cpu")
        mine = t.clone()
        if world > 1:
            got = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(got, mine)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            got = [mine]
        nj = args.job_images
        union_ok = [int(t[0]), int(t[1]), int(t[2])] == [nj, nj * (nj - 1) // 2, (nj - 1) * nj * (2 * nj - 1) // 6]
        if not union_ok:
            tot_err += 1
        per_bytes = [int(g[3]) for g in got]
        shards = {"images": [int(g[0]) for g in got], "compressed_bytes": per_bytes, "union_is_the_job": union_ok,
                  "byte_imbalance": round((max(per_bytes) - min(per_bytes)) / max(1, max(per_bytes)), 4),
                  "ms_spread": round((max(per_rank_ms) - min(per_rank_ms)) / max(per_rank_ms), 4) if max(per_rank_ms) > 0 else 0.0}

    extra = {}
    if rank == 0 and world == 1 and not stub and not args.strong and not args.no_split:
        # the same resident batch in the one-stream form, beside the headline (which ran the library's default form)
        try:
            batch.set_split(1)
            batch.decode(); batch.sync()
            device_sync(); t2 = time.perf_counter()
            for _ in range(args.steps):
                batch.decode()
            device_sync(); el2 = time.perf_counter() - t2
            batch.sync()
            ok2 = bool((batch.dib_checksums() == sums).all()) and not any(batch.info(i)["flags"] for i in range(len(batch)))
            extra["one_stream"] = {"ms_per_step": round(el2 / args.steps * 1e3, 4), "mpix_per_s": round(pixels * args.steps / el2 / 1e6, 1) if ok2 else 0.0,
                                   "bit_exact": ok2, "note": "same batch, same arenas, jsnoop_batch_set_split(1); the form the stage times and the roofline object are measured in"}
        except Exception as e:
            extra["one_stream"] = {"error": repr(e)}
    if not stub:
        batch.set_split(0)
    if rank == 0 and world == 1 and not stub and not args.no_extras and not args.strong:
        try:
            extra.update(extras_single_gpu(J, H, orc, np))
        except Exception as e:                                       # beside the headline, never a reason to lose it
            extra["extras_error"] = repr(e)
        try:
            extra["staging_pipeline"] = staging_pipeline(J, files, args.images)
        except Exception as e:
            extra["staging_pipeline"] = {"error": repr(e)}

    if rank == 0:
        value = tot_px / max_el / 1e6 if tot_err == 0 else 0.0
        out = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(max_el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "u8/i16 entropy+DIB, f32 IDCT+colour", "data": "synthetic" + (" (stub: no decode, rank logic only)" if stub else "") + (" (share-device: every rank on device 0 over gloo -- a test of the rank path, not a scaling figure)" if args.share_device else ""),
            "config": {"workload": (f"ONE job of {args.job_images} baseline 4:2:0 q85 JPEGs of mixed size (5/8 1920x1080, 2/8 1280x720, 1/8 3840x2160; "
                                    f"{args.distinct} distinct seeds per size), LPT-partitioned by compressed bytes over the ranks, HBM->HBM (T1)") if args.strong else
                                   (f"{args.images} x {args.width}x{args.height} baseline 4:2:0 q85 JPEG per GPU "
                                    f"({args.distinct} distinct seeds replicated; BASELINE config 3, config 4 at 8 GPUs), HBM->HBM (T1)"),
                       "images_per_gpu": (args.job_images // world) if args.strong else args.images, "distinct": args.distinct, "subsampling": "4:2:0", "quality": 85,
                       "parallelism": f"shard{world}" if world > 1 else "single", "entropy_path": "parallel" if all(p == 1 for p in paths) else "mixed",
                       "decode_form": decode_form},
            "bit_exact": tot_err == 0, "parity_errors": tot_err,
            "per_rank_ms_per_step": per_rank_ms, "job_checksum": "%016x" % job_ck, "images_oracle_checked_per_rank": n_checked,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(alg_bytes / (stages[dom] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(alg_bytes / (stages[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(stages[dom], 4),
                         "pipeline": {"ms": round(ms_whole, 4), "achieved": round(alg_bytes / (ms_whole * 1e-3) / 1e9, 1),
                                      "frac": round(alg_bytes / (ms_whole * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
                         "stages_ms": {k: round(v, 4) for k, v in stages.items()},
                         "measured_in": "one-stream decodes of the same resident batch (hipEvents between the stages)"},
        }
        if budget > 0 and n_cpu:
            out["cpu_baseline"] = {"value": round(n_cpu * args.width * args.height / cpu_time / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "cpu_model": cpu_model(),
                                   "kind": "port", "sample": f"{n_cpu} x {args.width}x{args.height} 4:2:0 images of this workload, oracle/oracle_imgdecode.c, "
                                                             f"1 thread, {cpu_time:.1f} s"}
            if ref_info:
                out["cpu_baseline"]["compiled_reference"] = dict(ref_info, unit="Mpixels/s", cores=1, kind="reference",
                                                                 sample=f"{ref_info['images']} of the same images through oracle/_ref (unmodified ImgDecode.cpp), 1 thread")
            if cfg1:
                out["cpu_baseline"]["config1_640x480_444"] = cfg1
            if all_cores:
                out["cpu_baseline"]["all_cores"] = all_cores
        # measured HBM traffic of the dominant kernel (rocprofv3 PMC passes of this same workload, tools/pmc_collect.sh)
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and not stub:
            try:
                tj = json.load(open(tf))
                if tj.get("kernel") == dom and tj.get("images_per_launch"):
                    out["roofline"]["traffic"] = int(tj["hbm_bytes_per_launch"] * args.images / tj["images_per_launch"])
                    out["roofline"]["traffic_source"] = tj.get("source")
            except Exception:
                pass
        if shards:
            out["shards"] = shards
        if not stub and not args.strong:
            comp = int(sum(len(f) for f in files) * (args.images / args.distinct))
            out["setup_s"] = {"synth": round(t_gen, 1), "first_upload_with_alloc": round(t_upload_first, 3)}
            out["pcie_inclusive_T2"] = {"h2d_ms": round(t_upload * 1e3, 3), "compressed_bytes": comp, "h2d_GBps": round(comp / t_upload / 1e9, 1),
                                        "mpix_per_s_serial": round(pixels / (t_upload + max_el / args.steps) / 1e6, 1),
                                        "note": "timing scope T2, serial form: pinned H2D of the whole compressed batch (stream drained), then decode; "
                                                "reported beside, never as, value"}
        out.update(extra)
        print(json.dumps(out))
    batch.close()
    if orc is not None:
        orc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
