"""Margins of a VCR_GRAD_REPORT log (tests/util.py::_report) under the acceptance rules the tree holds NOW: every statistic as a
fraction of its tolerance, worst first, and how many lie within 1.3 x of it (those are findings).
    python profiles/grad_report_margins.py profiles/r5_grad_report_final_build.txt > profiles/r5_grad_report_margins.txt
The log was written before the p99.9 floor was raised; the floor of tests/util.py::_FLOOR[2] is applied to its p99.9 columns here
(every other rule is as logged)."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util  # noqa: E402

rows = []
for line in open(sys.argv[1]):
    cols = [c.strip() for c in line.split("|")]
    stats = {q: (float(v), float(t)) for q, v, t in re.findall(r"(maxnorm|p99|p999) ([0-9.e+-]+)/([0-9.e+-]+)", line)}
    if "p999" in stats:
        v, t = stats["p999"]
        stats["p999"] = (v, max(t, util._FLOOR[2]))
    for q, (v, t) in stats.items():
        rows.append((v / t, q, v, t, cols[0].split("::")[-1], cols[1]))
rows.sort(reverse=True)
print(f"# {len(rows)} statistics of {sys.argv[1]}; value / tolerance, worst first")
for r in rows[:25]:
    print("%.2f  %-7s %.2e / %.1e  %s  %s" % r)
print(f"# within 1.3 x of the tolerance (value / tolerance > {1 / 1.3:.2f}): {sum(r[0] > 1 / 1.3 for r in rows)}")
