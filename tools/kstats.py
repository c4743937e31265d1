#!/usr/bin/env python3
"""Print per-kernel average durations from a rocprofv3 --kernel-trace --stats --output-format csv directory.
usage: python tools/kstats.py <dir> [substring]"""
import csv, glob, os, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Name"]:
            print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:8.2f} us  {float(r['Percentage']):5.1f} %")
