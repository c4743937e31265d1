"""Same-box A/B of two builds (or two environments) of the library on the bench step; writes the evidence as JSON.

    python tools/ab_bench.py --name ws_stagger_prio --a-lib <pkg>/lib/ab_prev.so --rounds 3 [--a-env AVF_NT_WS=0] [--b-env ...]

Arm A and arm B alternate (A B A B ...), each run is one `python bench.py --steps N` process on this GPU; the JSON
(profiles/ab/<name>.json, or gpurun_out/ab/<name>.json on a GPU box - copy it into profiles/ab/ to keep it) records the box
(GPU uuid / hostname), every run's ms per step for the main workload (C2) and the north-star shape (C3), and the medians.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def box_id():
    try:
        out = subprocess.run(["rocm-smi", "--showuniqueid", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        return {"host": socket.gethostname(), "gpu": next(iter(j.values())).get("Unique ID", "?")}
    except Exception:
        return {"host": socket.gethostname(), "gpu": "?"}


def run(env_extra, steps, extra):  # extra: bench.py arguments of this arm
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps)] + extra, capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not line:
        raise RuntimeError(f"bench.py printed no JSON line:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
    d = json.loads(line[-1])
    ns = d.get("north_star_shape") or {}
    return {"c2_ms": d["ms_per_step"], "c3_ms": ns.get("ms_per_step"), "value": d["value"]}  # (c2_ms = the main workload's)


def parse_env(items):
    out = {}
    for it in items or []:
        k, v = it.split("=", 1)
        out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", required=True)
    ap.add_argument("--a-lib", default=None, help="AVF_LIB_PATH of arm A (default: the built library)")
    ap.add_argument("--b-lib", default=None)
    ap.add_argument("--a-env", action="append")
    ap.add_argument("--b-env", action="append")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--note", default="")
    ap.add_argument("--a-args", default="", help="bench.py arguments of arm A only (one string, e.g. '--dropout 0')")
    ap.add_argument("--b-args", default="")
    args, rest = ap.parse_known_args()  # everything this tool does not know goes to bench.py (e.g. --config c5 --dtype mx8)
    args.bench_args = rest
    ea, eb = parse_env(args.a_env), parse_env(args.b_env)
    if args.a_lib:
        ea["AVF_LIB_PATH"] = os.path.abspath(args.a_lib)
    if args.b_lib:
        eb["AVF_LIB_PATH"] = os.path.abspath(args.b_lib)
    for e in (ea, eb):  # the library honours its tuning switches under AVF_TUNING=1 only
        if any(k.startswith("AVF_") and k != "AVF_LIB_PATH" for k in e):
            e.setdefault("AVF_TUNING", "1")
    runs = {"A": [], "B": []}
    for i in range(args.rounds):
        for arm, env in (("A", ea), ("B", eb)):
            runs[arm].append(run(env, args.steps, args.bench_args + (args.a_args if arm == "A" else args.b_args).split()))
            print(f"round {i + 1} arm {arm}: {runs[arm][-1]}", flush=True)  # (a silent GPU-box call is taken for a hung one)
    med = lambda arm, k: statistics.median([r[k] for r in runs[arm] if r[k] is not None]) if any(r[k] is not None for r in runs[arm]) else None
    out = {"name": args.name, "note": args.note, "box": box_id(), "alternations": args.rounds, "steps": args.steps,
           "arm_A": {"env": ea, "args": args.a_args, "runs": runs["A"], "median_c2_ms": med("A", "c2_ms"), "median_c3_ms": med("A", "c3_ms")},
           "arm_B": {"env": eb, "args": args.b_args, "runs": runs["B"], "median_c2_ms": med("B", "c2_ms"), "median_c3_ms": med("B", "c3_ms")}}
    a2, b2, a3, b3 = out["arm_A"]["median_c2_ms"], out["arm_B"]["median_c2_ms"], out["arm_A"]["median_c3_ms"], out["arm_B"]["median_c3_ms"]
    out["B_over_A_c2"] = round(b2 / a2, 4) if a2 and b2 else None
    out["B_over_A_c3"] = round(b3 / a3, 4) if a3 and b3 else None
    on_box = os.path.isdir(os.path.join(ROOT, "gpurun_out"))
    d = os.path.join(ROOT, "gpurun_out" if on_box else "profiles", "ab")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, args.name + ".json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("name", "B_over_A_c2", "B_over_A_c3")}), "->", path)
    print("A c2:", [r["c2_ms"] for r in runs["A"]], "c3:", [r["c3_ms"] for r in runs["A"]])
    print("B c2:", [r["c2_ms"] for r in runs["B"]], "c3:", [r["c3_ms"] for r in runs["B"]])


if __name__ == "__main__":
    main()
