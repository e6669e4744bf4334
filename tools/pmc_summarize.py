"""Aggregates rocprofv3 --pmc CSV output (one directory per pass) into per-kernel averages per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")      # "void k_idct_color<1>(...)" -> "k_idct_color<1>"
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        n = cnt[k][c]
        print("   %-40s %16.1f per dispatch (%d dispatches)" % (c, agg[k][c] / n, n))
