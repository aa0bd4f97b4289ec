"""Summarise rocprofv3 --pmc csv output directories produced by tools/pmc_gemm.sh:
per kernel name, the mean of each counter over dispatches."""
import csv, glob, os, sys, collections
root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if filt in k:
                res[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in res.items():
    print(k[:110])
    for c, v in sorted(cs.items()):
        print(f"    {c:40s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
