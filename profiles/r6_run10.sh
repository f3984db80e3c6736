#!/bin/bash
# round 6, call 10: kernel mix of the visibility renders of a densification event; quad lists at the metric scene end to end
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_run10
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vt -o v -- python $R/profiles/r6_visibility_trace.py 3 > $OUT/visibility_trace.txt 2>&1
cd $R
cat $OUT/visibility_trace.txt | grep "visibility batch"
cp $(ls $OUT/vt/*/*kernel_stats.csv $OUT/vt/*kernel_stats.csv 2>/dev/null | head -1) $OUT/visibility_kernel_stats.csv
# timeline of cameras 100..103 of the last batch: start / duration / stream of every kernel
python - <<'PY'
import csv, glob, os
f = (glob.glob(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r6_run10/vt/**/*kernel_trace.csv", recursive=True) or [None])[0]
if f:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last 200 composite_fwd_v2 launches with FC = 4 end the last batch; take a window of ~3 ms in its middle
    comp = [i for i, r in enumerate(rows) if "composite_fwd_v2" in r["Kernel_Name"]]
    mid = comp[-100]
    t0 = int(rows[mid]["Start_Timestamp"])
    with open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r6_run10/visibility_timeline.txt", "w") as out:
        for r in rows[mid - 150: mid + 150]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            out.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'].split('(')[0][-60:]}\n")
PY
rm -rf $OUT/vt
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r6_run10/visibility_kernel_stats.csv")))
for r in rows[:22]:
    print(r["Name"].split("(")[0][-70:].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
for q in 2.7 3.2; do
  timeout 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-context --quad-below $q > $OUT/bench_q$q.json 2> $OUT/bench_q$q.err
  python - $OUT/bench_q$q.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d["config"]["quad_lists"], round(d["ms_per_step"], 4), d["stage_ms"], round(d["roofline"]["frac"], 4))
PY
done
