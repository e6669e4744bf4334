"""Shared generator of hostile inputs for the parity sweeps: corrupted scan bytes and mutated header fields
(dimensions, sampling factors, precision, restart interval, table selectors, single-component scans of a
three-component frame, shifted scan start, damaged Huffman tables).  Deterministic for a given seed."""
import copy

import numpy as np


def bases(H):
    specs = [(160, 96, {}), (141, 93, dict(hs=2, vs=1, restart_interval=3)), (128, 64, dict(hs=1, vs=1, restart_interval=1)), (97, 61, dict(gray=1)),
             (200, 120, dict(quality=25, restart_interval=7)), (96, 96, dict(quality=97, optimize_huffman=1)), (64, 48, dict(hs=1, vs=2))]
    return [H.synth_jpeg(width=w, height=h, seed=s + 1, **kw) for s, (w, h, kw) in enumerate(specs)]


def mutate(H, rng, base):
    """Returns (data, parsed): one randomly damaged variant of `base`."""
    p = H.parse_jpeg(base)
    d = bytearray(base)
    s, e = p.scan_start, p.scan_end
    mode = int(rng.integers(14))
    if mode == 0:
        for _ in range(int(rng.integers(1, 5))):
            d[int(rng.integers(s, e))] ^= 1 << int(rng.integers(8))
    elif mode == 1:
        d = d[: int(rng.integers(s + 1, len(d)))]
    elif mode == 2:
        i = int(rng.integers(s, e)); d[i:i] = bytes([0xFF, int(rng.integers(0xC0, 0xFF))])
    elif mode == 3:
        i = int(rng.integers(s, e)); d[i:i] = b"\xff" * int(rng.integers(2, 5))
    elif mode == 4:
        i = int(rng.integers(s, max(s + 1, e - 8))); del d[i:i + int(rng.integers(1, 6))]
    elif mode == 5:
        i = int(rng.integers(s, e)); d[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
    data = bytes(d)
    q = H.parse_jpeg(data) if mode != 1 else copy.deepcopy(p)
    if mode == 6:                                                # sampling factors
        q.comps = [(c[0], int(rng.integers(1, 5)), int(rng.integers(1, 5)), c[3]) for c in p.comps]
    elif mode == 7:                                              # dimensions
        q.x = max(1, int(p.x * rng.uniform(0.3, 1.8))); q.y = max(1, int(p.y * rng.uniform(0.3, 1.8)))
    elif mode == 8:                                              # restart interval announced != used
        q.rst_en = bool(rng.integers(2)); q.rst_interval = int(rng.integers(0, 12))
    elif mode == 9:                                              # precision
        q.precision = int(rng.choice([0, 2, 7, 9, 10, 12, 16]))
    elif mode == 10 and len(p.comps) == 3:                       # one-component scan of a three-component frame
        q.scan_comps = [p.scan_comps[int(rng.integers(3))]]
    elif mode == 11:                                             # table selectors swapped around
        q.scan_comps = [(c[0], int(rng.integers(2)), int(rng.integers(2))) for c in p.scan_comps] if len(p.comps) == 3 else p.scan_comps
    elif mode == 12:                                             # scan start off by a few bytes
        q.scan_start = max(2, p.scan_start + int(rng.integers(-3, 6)))
    elif mode == 13:                                             # a Huffman table with codes removed / symbols changed
        key = list(p.dht.keys())[int(rng.integers(len(p.dht)))]
        counts, vals = p.dht[key]
        counts, vals = list(counts), list(vals)
        if rng.integers(2) and sum(counts) > 2:
            ln = max(i for i, c in enumerate(counts) if c); counts[ln] -= 1; vals = vals[:-1]
        else:
            vals[int(rng.integers(len(vals)))] = int(rng.integers(256))
        q.dht = dict(p.dht); q.dht[key] = (counts, vals)
    return data, q, mode


def differs(a, b, stats=False):
    """None when backend b reproduces backend a on everything the decoder exposes, else the name of the first difference."""
    da, db = a.dib(), b.dib()
    if (da is None) != (db is None):
        return "preview"
    if da is None:
        return None if a.image_size() == b.image_size() else "size"
    if a.image_size() != b.image_size():
        return "size"
    if not np.array_equal(da, db):
        return "dib"
    for x, y in zip(a.planes(), b.planes()):
        if x is not None and not np.array_equal(x, y):
            return "planes"
    if not np.array_equal(a.mcu_map(), b.mcu_map()):
        return "mcu_map"
    for x, y in zip(a.blk_dc(), b.blk_dc()):
        if x is not None and not np.array_equal(x, y):
            return "blk_dc"
    if not np.array_equal(a.dht_histo(), b.dht_histo()):
        return "dht_histo"
    if a.status() != b.status():
        return "status %s vs %s" % (a.status(), b.status())
    if a.bright_avg() != b.bright_avg():
        return "bright_avg"
    if stats:
        sa, sb = a.color_stats(), b.color_stats()
        for k in sa:
            if (sa[k] != sb[k]) if k == "count" else (not np.array_equal(sa[k], sb[k])):
                return "color_stats." + k
    return None
