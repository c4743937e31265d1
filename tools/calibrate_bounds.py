#!/usr/bin/env python3
"""tests/golden/lowp_measured.json from a calibration run of the GPU tests.

    AVF_RECORD_ERRORS=gpurun_out/errors.json python -m pytest tests -m gpu -q        (on the MI355X box)
    python tools/calibrate_bounds.py gpurun_out/errors.json [--merge | --keep-raised | --new-only] [--allow-regress "<reason>"]

Every low-precision assertion in tests/ (gpu_util.check*) is then held to 3x the value recorded here (and to its
stated cap).  --merge keeps entries of the existing file that the run did not touch; --keep-raised (implies --merge) records
new tags and lower values only and leaves every entry whose new measurement is higher untouched - no bound gets looser; --new-only records tags the file does not hold yet and nothing else.

The file is FROZEN against regressions: a tag whose new measurement is HIGHER than the recorded one is a looser bound, and
a kernel regression that lands just before a re-calibration would be baked in.  Without --allow-regress the tool refuses
(exit 2, nothing written) and lists those tags; with it, the reason, the commit and every raised tag (old -> new) are
appended to the file's "regress_log", which the commit message of the re-calibration should quote.  New tags and LOWER
values are always accepted."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "lowp_measured.json")
RAISE_TOL = 1e-6  # relative; the kernels are bitwise deterministic, so an unchanged case reproduces its value


def raised_tags(old, new):
    """tags present in both whose new value exceeds the recorded one -> {tag: (old, new)}"""
    return {k: (old[k], v) for k, v in new.items() if k in old and v > old[k] * (1.0 + RAISE_TOL) + 1e-30}


def main(argv=None, out_path=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    out_path = out_path or OUT
    reason = None
    if "--allow-regress" in argv:
        i = argv.index("--allow-regress")
        if i + 1 >= len(argv) or argv[i + 1].startswith("--") or not argv[i + 1].strip():
            print("--allow-regress needs a reason", file=sys.stderr)
            return 2
        reason = argv[i + 1]
        del argv[i:i + 2]
    keep_raised = "--keep-raised" in argv  # take new tags and LOWER values only; a higher measurement leaves the record as it is
    new_only = "--new-only" in argv        # take new tags only: every recorded value stays (nothing gets looser OR tighter)
    merge = "--merge" in argv or keep_raised or new_only
    src = [a for a in argv if not a.startswith("--")][0]
    new = json.load(open(src))
    doc = {}
    if os.path.exists(out_path):
        doc = json.load(open(out_path))
    old = doc.get("measured", {})
    if new_only:
        skipped = sum(1 for k in new if k in old)
        new = {k: v for k, v in new.items() if k not in old}
        print(f"--new-only: {skipped} already recorded tag(s) left as they are")
    up = raised_tags(old, new)
    if keep_raised and up:
        print(f"--keep-raised: {len(up)} higher measurement(s) ignored (their recorded values stay)")
        new = {k: v for k, v in new.items() if k not in up}
        up = {}
    if up and reason is None:
        print(f"REFUSED: {len(up)} recorded value(s) would be RAISED (looser bounds).  Fix the regression, or re-run with "
              f"--allow-regress \"<reason>\":", file=sys.stderr)
        for k, (a, b) in sorted(up.items(), key=lambda t: -t[1][1] / max(t[1][0], 1e-30))[:40]:
            print(f"  {k}: {a:.3e} -> {b:.3e}  (x{b / max(a, 1e-30):.2f})", file=sys.stderr)
        return 2
    cur = dict(old) if merge else {}
    cur.update(new)
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except Exception:
        head = ""
    log = list(doc.get("regress_log", []))
    if up:
        log.append({"commit_before": head, "reason": reason,
                    "raised": {k: [a, b] for k, (a, b) in sorted(up.items())}})
    json.dump({"note": "errors measured on one MI355X by the -m gpu tests (gpu_util.check); bounds = 3x these; frozen: "
                       "values only go up through tools/calibrate_bounds.py --allow-regress (see regress_log)",
               "commit_before": head, "regress_log": log, "measured": dict(sorted(cur.items()))}, open(out_path, "w"), indent=0)
    lowered = sum(1 for k, v in new.items() if k in old and v < old[k])
    print(f"{len(new)} measured ({len(new) - len([k for k in new if k in old])} new tags, {lowered} lowered, {len(up)} raised), "
          f"{len(cur)} total -> {out_path}")
    for k, (a, b) in sorted(up.items()):
        print(f"  raised {k}: {a:.3e} -> {b:.3e}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
