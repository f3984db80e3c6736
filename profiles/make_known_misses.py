"""tests/golden/grad_known_misses.json from a VCR_GRAD_REPORT log: the gradient comparisons that sit at or above 0.8 x the
contract figure (max-norm relative error 1e-4 on the rows the oracle does not mark fragile), BY NAME, each with the figure the log
holds and the stage whose adjoint produces the tensor.  tests/util.py holds exactly these to KNOWN_MISS_BOUND instead of the
contract; everything else must meet 1e-4.  Comparisons that carry their own documented `maxnorm_tol` (the depth-variance and the
curvature step tests) are not listed: their bound is in the test.
    python profiles/make_known_misses.py profiles/r6_grad_report_mixed_k4.txt > tests/golden/grad_known_misses.json"""
import json
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from profiles.grad_vs_contract import STAGE  # noqa: E402

CONTRACT, NEAR = 1e-4, 0.8
# Several logs (one per run of the suite): a comparison is listed when ANY run has it at or above NEAR x CONTRACT, with the LARGEST
# figure seen and how many of the runs had it there -- the fp32 atomics of the compositing backward arrive in a different order
# every run, and the max-norm of a tensor is ONE row: `c1 view 2:scales` measured 2.8e-5 in one run and 1.06e-4 in the next
# (profiles/r6_run8_pytest_raster_blk.txt), so a single log under-lists.
out, seen = {}, {}
for path in sys.argv[1:]:
    for line in open(path):
        cols = [c.strip() for c in line.split("|")]
        m = re.match(r"(.*) \[non-fragile (\d+)/(\d+)\]", cols[1]) if len(cols) > 3 else None
        mx = re.search(r"maxnorm ([0-9.e+-]+)/([0-9.e+-]+)", line)
        if not m or not mx:
            continue
        val, tol = float(mx.group(1)), float(mx.group(2))
        if tol > 3.5e-4:                   # the test passes its own documented bound
            continue
        key = f"{cols[0].split('::')[-1]}|{m.group(1)}"
        seen.setdefault(key, []).append(val)
        if val >= NEAR * CONTRACT and val >= out.get(key, {"measured": 0.0})["measured"]:
            out[key] = {"measured": val, "rows": f"{m.group(2)}/{m.group(3)}", "stage": STAGE.get(m.group(1).split(":")[-1], "-")}
for key, e in out.items():
    v = seen[key]
    e["runs_at_or_above_listing"] = f"{sum(x >= NEAR * CONTRACT for x in v)}/{len(v)}"
    e["smallest_seen"] = min(v)
    e["status"] = "misses 1e-4" if e["measured"] >= CONTRACT else "within 20 % of 1e-4"
print(json.dumps({"source": sys.argv[1:], "contract": CONTRACT, "listed_from": NEAR * CONTRACT, "bound_for_listed": 3e-4,
                  "count": len(out), "misses": dict(sorted(out.items(), key=lambda kv: -kv[1]["measured"]))}, indent=1))
