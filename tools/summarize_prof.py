#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel-trace stats + PMC passes) into a short per-kernel table.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 bytes; on gfx950 FETCH_SIZE is
TCC_EA0_RDREQ x 64 B although the requests are 128-byte line fills (MI355X_MICROARCH.md, HBM section; confirmed for
the gather kernels here by TCC_EA0_RDREQ_128B, profiles/r01_v6_gl7d19_pmc_l2_fabric.txt), so the raw value and the
doubled value are both shown; the doubled one is the byte count.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, suffix):
    hits = glob.glob(os.path.join(root, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


st = find("stats", "kernel_stats.csv")
if st:
    print("== kernel-trace stats (rocprofv3 --kernel-trace --stats) ==")
    print(f"{'kernel':70s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for r in csv.DictReader(open(st)):
        print(f"{short(r['Name']):70s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.1f} "
              f"{float(r['TotalDurationNs'])/1e6:10.2f} {float(r['Percentage']):6.1f}")
traffic = defaultdict(dict)
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    print(f"== {ctr} per launch (rocprofv3 --pmc {ctr}); value x 1024 bytes ==")
    for k, (tot, cnt) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        mean = tot / cnt
        extra = f"  (x2 = bytes: {mean*2*1024/1e6:10.1f} MB)" if ctr == "FETCH_SIZE" else ""
        print(f"{k:70s} launches {cnt:6d}  mean {mean*1024/1e6:10.1f} MB{extra}")
        traffic[k][ctr + "_bytes_per_launch"] = mean * 1024
        traffic[k]["launches"] = cnt
if traffic:
    with open(os.path.join(root, "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
