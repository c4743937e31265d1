#!/usr/bin/env python3
"""tests/golden/lowp_measured.json from a calibration run of the GPU tests.

    AVF_RECORD_ERRORS=gpurun_out/errors.json python -m pytest tests -m gpu -q        (on the MI355X box)
    python tools/calibrate_bounds.py gpurun_out/errors.json [--merge]

Every low-precision assertion in tests/ (gpu_util.check*) is then held to 3x the value recorded here (and to its
stated cap).  --merge keeps entries of the existing file that the run did not touch."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "lowp_measured.json")


def main():
    src = sys.argv[1]
    new = json.load(open(src))
    cur = {}
    if "--merge" in sys.argv and os.path.exists(OUT):
        cur = json.load(open(OUT))["measured"]
    cur.update(new)
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except Exception:
        head = ""
    json.dump({"note": "errors measured on one MI355X by the -m gpu tests (gpu_util.check); bounds = 3x these",
               "commit_before": head, "measured": dict(sorted(cur.items()))}, open(OUT, "w"), indent=0)
    print(f"{len(new)} measured, {len(cur)} total -> {OUT}")


if __name__ == "__main__":
    main()
