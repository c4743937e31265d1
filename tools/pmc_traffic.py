#!/usr/bin/env python3
"""Condense rocprofv3 outputs into the small summaries committed under profiles/.

  python tools/pmc_traffic.py <kernel_stats.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> <tag> [workload]

Writes profiles/<tag>_kernel_stats.csv (verbatim copy of rocprofv3 --kernel-trace --stats), profiles/<tag>_pmc_summary.csv
(per kernel: launches, mean FETCH_SIZE / WRITE_SIZE in KiB as reported, corrected HBM bytes per launch) and
profiles/<tag>_traffic.json (kernel class -> HBM bytes per launch, read by bench.py for roofline.traffic).

Correction (MI355X_MICROARCH.md, section HBM): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads,
WRITE_SIZE is exact; both are in KiB.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  FETCH and WRITE are collected in
separate passes (TCC slot budget)."""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).replace("avf::", "")
    return n.split("(")[0]


def klass(k):
    for c in ("gemm_bf16_nt", "gemm_bf16_tn", "gemm_f32", "attn_fwd", "attn_dq", "attn_dkv", "attn_bwd_m4", "ln_fwd", "ln_bwd"):
        if k.startswith(c):
            return c
    return None


def mean_counter(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def main():
    stats, fetch, write, tag = sys.argv[1:5]
    workload = sys.argv[5] if len(sys.argv) > 5 else "c2"
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
    f, w = mean_counter(fetch, "FETCH_SIZE"), mean_counter(write, "WRITE_SIZE")
    rows = []
    per_class = collections.defaultdict(lambda: [0, 0.0])
    for k in sorted(set(f) | set(w)):
        n = f.get(k, w.get(k))[0]
        fk, wk = f.get(k, (0, 0.0))[1], w.get(k, (0, 0.0))[1]
        b = (2 * fk + wk) * 1024
        rows.append((k, n, fk, wk, b))
        c = klass(k)
        if c:
            per_class[c][0] += n
            per_class[c][1] += n * b
    with open(os.path.join(out, f"{tag}_pmc_summary.csv"), "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["kernel", "launches", "mean_FETCH_SIZE_KiB_raw", "mean_WRITE_SIZE_KiB", "hbm_bytes_per_launch_corrected"])
        for r in sorted(rows, key=lambda r: -r[4] * r[1]):
            wr.writerow([r[0], r[1], f"{r[2]:.1f}", f"{r[3]:.1f}", f"{r[4]:.0f}"])
    traffic = {c: v / n for c, (n, v) in per_class.items()}
    sys.path.insert(0, ROOT)
    import subprocess
    from bench import kernel_source_hash
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    # the stamp bench.py checks: the traffic is quoted only while the kernel sources are the ones profiled
    json.dump({"unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)", "workload": workload,
               "kernel_sources_sha": kernel_source_hash(), "commit": head or "(not a git checkout: GPU box snapshot)",
               "per_class": traffic},
              open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
    for c, v in traffic.items():
        print(f"{c:14s} {v / 1e6:8.1f} MB/launch")


if __name__ == "__main__":
    main()
