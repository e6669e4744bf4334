"""Batch partitioning across the GPUs of one node.

Images are fully independent (the reference resets every decoder field per file,
source/JfifDecode.cpp:7306-7308), so the batch shards with NO data-path collective:
every rank owns its inputs, tables and DIB arena.  torch.distributed (RCCL over xGMI on
the GPU box, gloo in CPU tests) is used only for the completion barrier and the all-reduce
of a few scalars: pixels decoded, max elapsed time, XOR/sum of per-image DIB checksums and
the error count.
"""
from __future__ import annotations

from typing import Sequence


def partition_lpt(costs: Sequence[int], world_size: int) -> list[list[int]]:
    """Greedy longest-processing-time partition of image indices by cost (compressed scan bytes:
    entropy decode dominates).  Deterministic: ties broken by index."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    bins = [[] for _ in range(world_size)]
    load = [0] * world_size
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += costs[i]
    for b in bins:
        b.sort()
    return bins


def partition_contiguous(n: int, world_size: int) -> list[range]:
    """Plain contiguous split for iso-sized images (BASELINE config 4: 1024 per GPU)."""
    base, rem = divmod(n, world_size)
    out, start = [], 0
    for r in range(world_size):
        cnt = base + (1 if r < rem else 0)
        out.append(range(start, start + cnt))
        start += cnt
    return out


def reduce_job_stats(pixels: int, elapsed_s: float, checksum: int, errors: int, device=None):
    """All-reduce of the job scalars.  Returns (total_pixels, max_elapsed_s, checksum_sum, total_errors).
    With no initialised process group (single GPU) returns the inputs."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return pixels, elapsed_s, checksum & 0xFFFFFFFFFFFFFFFF, errors
    dev = device if device is not None else "cpu"
    # 64-bit checksums are summed mod 2^64 as two 32-bit halves in int64 lanes
    t = torch.tensor([pixels, errors, checksum & 0xFFFFFFFF, (checksum >> 32) & 0xFFFFFFFF], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    e = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    dist.all_reduce(e, op=dist.ReduceOp.MAX)
    lo, hi = int(t[2]), int(t[3])
    total = (lo + (hi << 32)) & 0xFFFFFFFFFFFFFFFF
    return int(t[0]), float(e[0]), total, int(t[1])
