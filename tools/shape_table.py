#!/usr/bin/env python3
"""Per-shape kernel table for profiles/: one row per distinct hot-path launch (kernel template, grid, operand shape, epilogue)
with its launches, mean duration from a rocprofv3 kernel trace, TFLOP/s, fraction of the bf16 MFMA roof, algorithmic bytes and -
when the FETCH_SIZE / WRITE_SIZE passes are given - measured HBM bytes per launch OF THAT SHAPE.

  python tools/shape_table.py <kernel_trace.csv> <shape_log.csv> <out.csv> [fetch_counter_collection.csv write_counter_collection.csv]

<kernel_trace.csv>: rocprofv3 --kernel-trace --output-format csv (one row per dispatch).  <shape_log.csv>: written by the library
under AVF_SHAPE_LOG=<file> during ONE eager step of the same workload (python bench.py --steps 1 --warmup 0 --launch eager ...):
lines "class,kernel template,grid (workgroups),M,N,K,epilogue,flops,bytes" in launch order.

Join (round 4; the round-3 tool joined by kernel NAME, which gave every shape of a template the template's mean): the shape log
is the launch sequence of one step; the hot-path dispatches of the trace (and of each PMC pass), in time / dispatch order, repeat
that sequence step after step, so every dispatch is matched to its log entry by POSITION with the kernel name as a check (a
dispatch the current entry does not accept moves the pointer on).  An attention call that dispatches to several kernels
(dQ + dK/dV, delta) is one entry: its duration and traffic are the sums over its kernels.
HBM bytes: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of wide reads; MI355X_MICROARCH.md, HBM)."""
import collections
import csv
import re
import sys

PEAK_TFLOPS = 2500.0
HOT = ("gemm_bf16_nt", "gemm_bf16_tn_group", "gemm_mx8_nt", "attn_")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).replace("avf::", "")
    return n.split("(")[0].strip()


def norm(t):
    return t.replace(" ", "")


class Entry:
    def __init__(self, cls, tmpl, wgs, M, N, K, epi, flops, byts):
        self.cls, self.tmpl, self.wgs, self.M, self.N, self.K, self.epi, self.flops, self.byts = cls, tmpl, wgs, M, N, K, epi, flops, byts
        self.multi = tmpl.startswith("attn")

    def key(self):
        return (self.cls, self.tmpl, self.wgs, self.M, self.N, self.K, self.epi)

    def accepts(self, name, wgs):
        if self.multi:
            if self.tmpl == "attn_fwd":
                return name.startswith("attn_fwd")
            return name.startswith(("attn_dq", "attn_dkv", "attn_bwd", "attn_delta"))
        if norm(self.tmpl) == norm(name):
            # (the grouped weight-gradient launch pads its grid for the XCD-aware block order: name only)
            return wgs == self.wgs or self.tmpl.startswith("gemm_bf16_tn_group")
        # fold / helper launches are not logged; a template logged without arguments matches by prefix
        return "<" not in self.tmpl and name.startswith(self.tmpl) and wgs == self.wgs


def read_log(path):
    seq = []
    for line in open(path):
        f = line.strip().split(",")
        if len(f) < 9:
            continue
        cls, rest = f[0], f[1:]
        flops, byts = float(rest[-2]), float(rest[-1])
        epi, K, N, M, wgs = rest[-3], rest[-4], rest[-5], rest[-6], int(rest[-7])
        seq.append(Entry(cls, ",".join(rest[:-7]), wgs, M, N, K, epi, flops, byts))
    return seq


def walk(seq, dispatches):
    """dispatches: iterable of (name, workgroups, payload) in launch order -> list of (entry index, call number, name, payload);
    a call = one pass of the pointer over an entry (the kernels of a multi-kernel attention call share a call number)"""
    out, ptr, call, open_multi, skipped = [], 0, 0, False, 0
    n = len(seq)
    for name, wgs, payload in dispatches:
        if not name.startswith(HOT):
            continue
        tries = 0
        while tries <= n:
            e = seq[ptr % n]
            if e.accepts(name, wgs):
                out.append((ptr % n, call, name, payload))
                if e.multi:
                    open_multi = True
                else:
                    ptr += 1
                    call += 1
                break
            if open_multi:
                open_multi = False
            ptr += 1
            call += 1
            tries += 1
        else:
            skipped += 1
    return out, skipped


def main():
    trace, shapes, out = sys.argv[1:4]
    pmc = sys.argv[4:6] if len(sys.argv) >= 6 else None
    seq = read_log(shapes)
    if not seq:
        sys.exit("empty shape log")
    disp = []
    for r in csv.DictReader(open(trace)):
        name = short(r["Kernel_Name"])
        wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        disp.append((int(r["Start_Timestamp"]), name, grid // max(wg, 1), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    disp.sort()
    matched, skipped = walk(seq, [(n, w, d) for _, n, w, d in disp])
    # per entry KEY (the six layers of a stack log the same entry six times): per call the summed duration of its kernels
    per_call = collections.OrderedDict()
    names = collections.defaultdict(list)
    for idx, call, name, d in matched:
        k = seq[idx].key()
        per_call.setdefault(k, collections.OrderedDict())
        per_call[k][call] = per_call[k].get(call, 0) + d
        if name not in names[k]:
            names[k].append(name)
    hbm = {}
    if pmc:
        for path, cname, mult in ((pmc[0], "FETCH_SIZE", 2.0), (pmc[1], "WRITE_SIZE", 1.0)):
            rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == cname]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            ds = []
            for r in rows:
                wg = int(r.get("Workgroup_Size", 0) or 0) or 1
                ds.append((short(r["Kernel_Name"]), int(r["Grid_Size"]) // wg, float(r["Counter_Value"])))
            m, _ = walk(seq, ds)
            acc = collections.defaultdict(lambda: collections.OrderedDict())
            for idx, call, _, v in m:
                k = seq[idx].key()
                acc[k][call] = acc[k].get(call, 0.0) + v
            for k, calls in acc.items():
                hbm[k] = hbm.get(k, 0.0) + mult * (sum(calls.values()) / len(calls)) * 1024.0
    rows = []
    info = {e.key(): e for e in seq}
    for k, calls in per_call.items():
        e = info[k]
        us = sum(calls.values()) / len(calls) / 1e3
        tf = e.flops / (us * 1e-6) / 1e12 if e.flops > 0 else 0.0
        rows.append([e.cls, " + ".join(names[k]), e.wgs, e.M, e.N, e.K, e.epi, len(calls), round(us, 2), round(tf, 1),
                     round(tf / PEAK_TFLOPS, 4), round(e.byts / 1e6, 2), round(hbm[k] / 1e6, 2) if k in hbm else "",
                     round(e.byts / (us * 1e-6) / 1e9, 1)])
    rows.sort(key=lambda r: -r[7] * r[8])
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["class", "kernel", "workgroups", "M (or B)", "N", "K (or H*dh)", "epilogue", "launches", "mean_us", "TFLOP/s",
                    "frac_of_bf16_mfma_peak", "algorithmic_MB", "hbm_MB_measured", "algorithmic_GB/s"])
        w.writerows(rows)
    for r in rows:
        print(",".join(str(v) for v in r))
    if skipped:
        print(f"# {skipped} hot-path dispatches matched no log entry", file=sys.stderr)


if __name__ == "__main__":
    main()
