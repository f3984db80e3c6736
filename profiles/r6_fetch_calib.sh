#!/bin/bash
# FETCH_SIZE on a gather of known byte count (profiles/microbench/fetch_calib.hip) -> gpurun_out/r6_fetch_calib.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -o f -- $R/profiles/microbench/fetch_calib > /tmp/fc.log 2>&1
python - <<'PY' > $R/gpurun_out/r6_fetch_calib.txt
import csv, glob, collections
f = glob.glob('/tmp/fc/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
print(open('/tmp/fc.log').read().strip().splitlines()[-1])
N = 4 << 20
need = {"gather_full": N * 68 / 1024, "gather_half": N * 36 / 1024, "stream_full": N * 64 / 1024}
print("# kernel | FETCH_SIZE per launch (counter units, KiB per the counter's definition) | bytes that must be fetched (KiB) | factor needed")
for k, v in acc.items():
    m = sum(v) / len(v)
    print(f"{k} | {m:.0f} | {need.get(k, 0):.0f} | {need.get(k, 0) / m:.3f}")
PY
cat $R/gpurun_out/r6_fetch_calib.txt
