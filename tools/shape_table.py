#!/usr/bin/env python3
"""Per-shape kernel table for profiles/ (VERDICT r02, weak item 6c): one row per distinct (kernel template, grid) of the hot
path with its operand shape, launches, mean duration from a rocprofv3 kernel trace, TFLOP/s, fraction of the bf16 MFMA roof,
algorithmic bytes and - when the FETCH_SIZE / WRITE_SIZE passes are given - measured HBM bytes per launch.

  python tools/shape_table.py <kernel_trace.csv> <shape_log.csv> <out.csv> [fetch_counter_collection.csv write_counter_collection.csv]

<kernel_trace.csv>: rocprofv3 --kernel-trace --output-format csv (one row per dispatch).  <shape_log.csv>: written by the library
under AVF_SHAPE_LOG=<file> during an EAGER run of the same workload (python bench.py --launch eager ...): lines
"class,kernel template,grid (workgroups),M,N,K,epilogue,flops,bytes".  Kernels are joined on (template name, workgroups);
shapes that share both (same tile grid, different K) are told apart by the order of their first appearance within a step.
HBM bytes: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of wide reads; MI355X_MICROARCH.md, HBM)."""
import collections
import csv
import re
import sys

PEAK_TFLOPS = 2500.0


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).replace("avf::", "")
    return n.split("(")[0].strip()


def main():
    trace, shapes, out = sys.argv[1:4]
    pmc = sys.argv[4:6] if len(sys.argv) >= 6 else None
    # ---- shape log: (template, wgs) -> ordered list of distinct (M, N, K, epi, flops, bytes)
    by_key = collections.OrderedDict()
    for line in open(shapes):
        f = line.strip().split(",")
        if len(f) < 9:
            continue
        # the template name itself contains commas: class, <template...>, wgs, M, N, K, epi, flops, bytes
        cls, rest = f[0], f[1:]
        flops, byts = float(rest[-2]), float(rest[-1])
        epi, K, N, M, wgs = rest[-3], rest[-4], rest[-5], rest[-6], int(rest[-7])
        tmpl = ",".join(rest[:-7])
        key = (tmpl, wgs)
        ent = (cls, M, N, K, epi, flops, byts)
        by_key.setdefault(key, [])
        if ent not in by_key[key]:
            by_key[key].append(ent)
    # ---- trace: group dispatches by (template, wgs)
    durs = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        name = short(r["Kernel_Name"])
        wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        durs[(name, grid // max(wg, 1))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    # ---- optional PMC passes
    hbm = {}
    if pmc:
        acc = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
        for path, cname, slot in ((pmc[0], "FETCH_SIZE", 0), (pmc[1], "WRITE_SIZE", 2)):
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] != cname:
                    continue
                wg = int(r.get("Workgroup_Size", 0) or 0) or 1
                key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // wg)
                acc[key][slot] += 1
                acc[key][slot + 1] += float(r["Counter_Value"])
        for k, (nf, f, nw, w) in acc.items():
            hbm[k] = (2.0 * (f / nf if nf else 0.0) + (w / nw if nw else 0.0)) * 1024.0
    def norm(t):
        return t.replace(" ", "")

    def lookup(name, wgs):
        for (tmpl, w2), e in by_key.items():
            if w2 != wgs:
                continue
            if norm(tmpl) == norm(name) or ("<" not in tmpl and name.startswith(tmpl)) or name.startswith(tmpl):
                return e
        return None

    rows = []
    for (name, wgs), lst in durs.items():
        ents = lookup(name, wgs)
        lst.sort()
        n = len(lst)
        if not ents:
            continue
        # several shapes on one (template, grid): launches alternate in a fixed per-step order -> split round-robin
        k = len(ents)
        for i, (cls, M, N, K, epi, flops, byts) in enumerate(ents):
            sub = [d for j, (_, d) in enumerate(lst) if j % k == i]
            if not sub:
                continue
            us = sum(sub) / len(sub) / 1e3
            tf = flops / (us * 1e-6) / 1e12 if flops > 0 else 0.0
            rows.append([cls, name, wgs, M, N, K, epi, len(sub), round(us, 2), round(tf, 1), round(tf / PEAK_TFLOPS, 4),
                         round(byts / 1e6, 2), "" if (name, wgs) not in hbm else round(hbm[(name, wgs)] / 1e6, 2),
                         round(byts / (us * 1e-6) / 1e9, 1)])
    # attention launches are logged per CALL ("attn_fwd", "attn_dq+attn_dkv"), whatever kernel(s) the call dispatches to (head-
    # resident, merged or the two-kernel / streaming forms, whose grids differ from the logged B*H): kernels not matched above
    # are joined by name prefix, a call's row = the sum of the mean durations of its kernels
    matched = {r[1] for r in rows}
    for (tmpl, wgs), ents in by_key.items():
        if not tmpl.startswith("attn"):
            continue
        parts = tmpl.split("+")
        ks = [(name, w) for (name, w) in durs if name not in matched and any(name.startswith(pp) for pp in parts)]
        if tmpl == "attn_dq+attn_dkv":
            ks += [(name, w) for (name, w) in durs if name not in matched and name.startswith("attn_bwd") and (name, w) not in ks]
        if not ks:
            continue
        cls, M, N, K, epi, flops, byts = ents[0]
        us = sum(sum(d for _, d in durs[k]) / len(durs[k]) for k in ks) / 1e3
        n = min(len(durs[k]) for k in ks)
        tf = flops / (us * 1e-6) / 1e12
        hb = sum(hbm.get(k, 0.0) for k in ks)
        rows.append([cls, " + ".join(k[0] for k in ks), "/".join(str(k[1]) for k in ks), M, N, K, epi, n, round(us, 2), round(tf, 1),
                     round(tf / PEAK_TFLOPS, 4), round(byts / 1e6, 2), round(hb / 1e6, 2) if hb else "", round(byts / (us * 1e-6) / 1e9, 1)])
        matched.update(k[0] for k in ks)
    rows.sort(key=lambda r: -r[7] * r[8])
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["class", "kernel", "workgroups", "M (or B)", "N", "K (or H*dh)", "epilogue", "launches", "mean_us", "TFLOP/s",
                    "frac_of_bf16_mfma_peak", "algorithmic_MB", "hbm_MB_measured", "algorithmic_GB/s"])
        w.writerows(rows)
    for r in rows:
        print(",".join(str(v) for v in r))


if __name__ == "__main__":
    main()
