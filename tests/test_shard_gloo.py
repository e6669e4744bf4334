"""CPU: the multi-GPU sharding logic with a real world_size-2 process group (gloo)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from jpegsnoop_amd.shard import partition_lpt, partition_contiguous, reduce_job_stats
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%d" %% int(os.environ["PORT"]), rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
costs = [(i * 7919) %% 1000 + 1 for i in range(37)]
mine = partition_lpt(costs, 2)[r]
cont = partition_contiguous(37, 2)[r]
px, el, ck, err = reduce_job_stats(sum(costs[i] for i in mine), 1.0 + r, (0xFFFFFFFFFFFFFF00 + r * 0x1234), r, None)
dist.barrier()
if r == 0:
    print(json.dumps(dict(px=px, el=el, ck=ck, err=err, n0=len(mine), c0=len(cont))))
dist.destroy_process_group()
''' % ROOT


def test_partition_is_balanced_and_complete():
    from jpegsnoop_amd.shard import partition_contiguous, partition_lpt
    costs = [(i * 7919) % 1000 + 1 for i in range(1000)]
    for w in (1, 2, 4, 8):
        bins = partition_lpt(costs, w)
        assert sorted(i for b in bins for i in b) == list(range(1000))
        loads = [sum(costs[i] for i in b) for b in bins]
        assert max(loads) - min(loads) <= max(costs)
        cont = partition_contiguous(1000, w)
        assert sum(len(c) for c in cont) == 1000 and max(len(c) for c in cont) - min(len(c) for c in cont) <= 1
    assert partition_lpt([], 4) == [[], [], [], []]


def test_world_size_2_reduce():
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    costs = [(i * 7919) % 1000 + 1 for i in range(37)]
    assert res["px"] == sum(costs)
    assert res["el"] == 2.0 and res["err"] == 1
    assert res["ck"] == (0xFFFFFFFFFFFFFF00 + 0xFFFFFFFFFFFFFF00 + 0x1234) & 0xFFFFFFFFFFFFFFFF
    assert res["c0"] == 19
