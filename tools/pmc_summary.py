#!/usr/bin/env python3
"""Mean per launch of every counter in the rocprofv3 --pmc csv files below a directory, grouped by kernel.
Usage: python tools/pmc_summary.py <dir> [kernel-name-substring ...]"""
import csv, glob, os, sys
from collections import defaultdict
root, want = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]
        if want and not any(w in k for w in want):
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        tot, cnt = acc[k][c]
        print(f"    {c:40s} launches {cnt:5d}  mean per launch {tot / cnt:18.1f}")
