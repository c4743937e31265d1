#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc ... --output-format csv counter_collection file per kernel: mean counter value per launch.
usage: python tools/pmc_agg.py <dir> [kernel substring]"""
import collections, csv, glob, os, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k: continue
        k = k.split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, cs in acc.items():
    print(k, "launches", len(n[k]))
    for c, v in sorted(cs.items()): print(f"   {c:32s} {v / len(n[k]):16.0f}")
