"""us per step per kernel from a rocprofv3 kernel_stats.csv of a bench.py run (steps = launches of au_loss_kernel)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = next((int(r["Calls"]) for r in rows if "au_loss_kernel" in r["Name"]), 1)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.replace("avf::", "")
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:100]


tot = 0.0
out = []
for r in rows:
    us = float(r["TotalDurationNs"]) / 1e3 / steps
    tot += us
    out.append((us, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, short(r["Name"])))
out.sort(reverse=True)
print(f"steps {steps}  total {tot:.1f} us/step")
for us, cps, avg, n in out:
    if us >= 0.5:
        print(f"{us:9.1f} us/step  {cps:6.1f} x {avg:8.2f} us  {n}")
