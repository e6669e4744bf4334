"""Probe: does decoding the 1024-image workload as TWO concurrent half batches (own streams) beat one batch?
usage: python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
files = [H.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=s + 1) for s in range(16)]
def make(n):
    b = J.JpegBatch()
    for f in files: b.add_jpeg(f)
    b.tile(n); b.upload(); b.decode(); b.sync()
    return b
def timeit(batches, reps=8):
    for b in batches: b.decode()
    for b in batches: b.sync()
    t = time.perf_counter()
    for _ in range(reps):
        for b in batches: b.decode()
        for b in batches: b.sync()
    return (time.perf_counter() - t) / reps * 1e3
one = make(1024); print("one batch of 1024: %.3f ms" % timeit([one])); one.close()
two = [make(512), make(512)]; print("two batches of 512, concurrent: %.3f ms" % timeit(two)); [b.close() for b in two]
four = [make(256) for _ in range(4)]; print("four batches of 256, concurrent: %.3f ms" % timeit(four)); [b.close() for b in four]
