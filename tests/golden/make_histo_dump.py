#!/usr/bin/env python3
"""Golden log text with bDumpHistoY on (ReportHistogramY, source/ImgDecode.cpp:3845-3868), written by the COMPILED REFERENCE
(oracle/_ref, needs /root/reference) for a few committed golden cases -> tests/golden/histo_dump.json.  The GPU box only reads it."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import harness as H          # noqa: E402
from golden_util import load_case        # noqa: E402

CASES = ["c1_444_160x120", "420_odd_141x93", "bad_flip_gray_101x77"]


def main():
    H.build(["ref"])
    ref = H.ref_backend()
    out = {}
    import glob
    names = [n for n in CASES if glob.glob(os.path.join(HERE, n + ".jpg"))]
    if len(names) < 2:                     # fall back to whatever golden cases exist
        names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "*.jpg")))[:3]
    for name in names:
        data = load_case(name)
        ref.set_options(decode_ac=1, histo_en=1)
        ref.set_dump_histo_y(1)
        H.drive(ref, data, quiet=0)
        out[name] = ref.log_lines()
        assert any("Y Histogram in DC" in l for l in out[name]), name
    ref.set_dump_histo_y(0); ref.set_options()
    json.dump(out, open(os.path.join(HERE, "histo_dump.json"), "w"), indent=0)
    print("wrote", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
