"""Shared helpers for the golden-fixture tests."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_case(name):
    with open(os.path.join(GOLDEN, name + ".jpg"), "rb") as f:
        return f.read()


def record(H, b, side=True):
    """Same record tests/golden/make_golden.py stores, taken from any backend."""
    dib = b.dib()
    if dib is None:
        return {"preview": False, "size": list(b.image_size())}
    r = {"preview": bool(b.is_preview_ready()), "size": list(b.image_size()), "dib": H.hash_bytes(dib)}
    r["planes"] = [H.hash_bytes(p) if p is not None else None for p in b.planes()]
    if side:
        r["mcu_map"] = H.hash_bytes(b.mcu_map())
        r["blk_dc"] = [H.hash_bytes(p) if p is not None else None for p in b.blk_dc()]
        r["histo"] = H.hash_bytes(b.dht_histo())
        r["status"] = {k: int(v) for k, v in b.status().items()}
        r["bright_avg"] = [int(v) for v in b.bright_avg()]
    return r


def stats_record(H, b):
    import numpy as np
    st = b.color_stats()
    words = np.concatenate([st["histo"].view(np.uint32), np.array([st["count"]], np.uint32), st["clip"], st["rgb"].ravel(), st["yfull"]])
    return {"sha256": H.hash_bytes(words), "count": st["count"], "clip": [int(v) for v in st["clip"]], "histo": [int(v) for v in st["histo"]]}


def histo_record(H, b, data):
    """Same bHistoEn record tests/golden/make_golden.py stores, taken from any backend."""
    b.set_options(decode_ac=1, histo_en=1)
    try:
        H.drive(b, data)
        if b.dib() is None:
            return {"preview": False}
        r = {"preview": True, "dib": H.hash_bytes(b.dib()), "stats": stats_record(H, b)}
        b.set_preview_mode(6)
        b.set_preview_ycc_offset(1, 1, 500, -200, 100)
        r["dib_rerender"] = H.hash_bytes(b.dib())
        r["stats_rerender"] = stats_record(H, b)
        b.set_preview_ycc_offset(0, 0, 0, 0, 0)
        b.set_preview_mode(1)
        return r
    finally:
        b.set_options(decode_ac=1)
