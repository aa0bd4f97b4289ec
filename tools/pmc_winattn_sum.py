"""Per-dispatch counters of the window-attention backward launches collected by tools/pmc_winattn.sh: python tools/pmc_winattn_sum.py <outdir>"""
import collections
import csv
import glob
import sys

out = sys.argv[1]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]) if "Dispatch_Id" in r else 0)
        rows[(f.split("/")[-3] if f.count("/") > 2 else f, key)][r["Counter_Name"]] = float(r["Counter_Value"])
        rows[(f.split("/")[-3] if f.count("/") > 2 else f, key)]["grid"] = r.get("Grid_Size", "")
for k in sorted(rows):
    print(k, rows[k])
