"""Average per-dispatch PMC values per kernel from the rocprofv3 --pmc csv outputs of tools/pmc.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
files = glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True) + \
    glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(s in k for s in ("likelihood", "octree", "population", "aabb", "resample")):
        continue
    print("==", k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        tail = v[len(v) // 2:]  # steady-state half
        print("   %-32s n=%3d  avg=%.4g" % (c, len(v), sum(tail) / len(tail)))
