"""Turns a tools/pmc_collect.sh summary into profiles/traffic.json: measured HBM bytes per launch of the dominant
kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
FETCH_SIZE and WRITE_SIZE are in KiB and come from separate --pmc passes; FETCH_SIZE reports half of the bytes of a
coalesced streaming read, so it is doubled (calibrated here on a known byte count: the IDCT kernel reads the 1.605 GB
coefficient arena + 25 MB of DC words per 256 images and FETCH_SIZE*1024 reads 0.831 GB)."""
import json, re, sys
summary, kernel, images, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
txt = open(summary).read()
blk = re.search(r"^%s\n((?:   .*\n)+)" % re.escape(kernel), txt, re.M).group(1)
vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+([0-9.]+) per dispatch", blk, re.M)}
hbm = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
json.dump({"kernel": {"k_idct_color": "idct_color", "k_write": "write", "k_sync": "sync"}.get(kernel.split("<")[0], kernel), "images_per_launch": images,
           "hbm_bytes_per_launch": int(hbm), "fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), 2*FETCH+WRITE, %s" % summary}, open(out, "w"), indent=1)
print(open(out).read())
