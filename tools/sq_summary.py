#!/usr/bin/env python3
"""Condense a rocprofv3 --pmc SQ_* counter_collection.csv into profiles/<tag>_sq_counters.csv (per kernel: launches,
SQ_WAVE_CYCLES per launch, the SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_* shares of it, MFMA-busy cycles per wave
cycle, LDS bank-conflict cycles per launch).
usage: python tools/sq_summary.py <counter_collection.csv> [<second pass csv> ...] <tag>"""
import collections, csv, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).replace("avf::", "")
    return n.split("(")[0]


def main():
    *paths, tag = sys.argv[1:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in paths:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k.startswith("at::") or k.startswith("Cijk") or k.startswith("__amd"):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]].add((path, r["Dispatch_Id"]))
    out = os.path.join(ROOT, "profiles", f"{tag}_sq_counters.csv")
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "SQ_WAVE_CYCLES_per_launch", "WAIT_ANY_pct", "WAIT_INST_ANY_pct", "ACTIVE_INST_ANY_pct",
                    "ACTIVE_INST_VALU_pct", "ACTIVE_INST_LDS_pct", "VALU_MFMA_BUSY_CYCLES_per_WAVE_CYCLE", "LDS_BANK_CONFLICT_per_launch"])
        def per(k, c):
            n = len(launches[k][c])
            return acc[k][c] / n if n else float("nan")
        for k in sorted(acc, key=lambda k: -per(k, "SQ_WAVE_CYCLES") * len(launches[k]["SQ_WAVE_CYCLES"])):
            wc = per(k, "SQ_WAVE_CYCLES")
            pct = lambda c: f"{100.0 * per(k, c) / wc:.1f}" if wc == wc and wc > 0 else ""
            w.writerow([k, len(launches[k]["SQ_WAVE_CYCLES"]), f"{wc:.0f}", pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"),
                        pct("SQ_ACTIVE_INST_ANY"), pct("SQ_ACTIVE_INST_VALU"), pct("SQ_ACTIVE_INST_LDS"),
                        f"{per(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / wc:.3f}" if wc > 0 else "", f"{per(k, 'SQ_LDS_BANK_CONFLICT'):.0f}"])
    print("wrote", out)


if __name__ == "__main__":
    main()
