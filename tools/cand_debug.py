"""One decode of N distinct 1080p images with JSNOOP_DEBUG_CAND=2 (open links after the candidate chain, per image).  usage: python tools/cand_debug.py N"""
import os, sys
os.environ["JSNOOP_DEBUG_CAND"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegsnoop_amd as J
from oracle import harness as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
b = J.JpegBatch()
for i in range(n):
    b.add_jpeg(H.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=100 + i))
b.upload(); b.decode(); b.sync(); b.close()
