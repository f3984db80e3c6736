"""A/B helper for kernel variants: run bench.py on several workloads (optionally against another build of the library
given with VCR_LIB=...) and print one compact line each.  Usage: python profiles/ab_bench.py [tag] [workload ...]"""
import json
import os
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
loads = sys.argv[2:] or ["metric_1m_1080p", "c5_360_5m_1600x1200", "c2_dtu_300k_800x600"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for wl in loads:
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--steps", "30",
                          "--warmup", "8"], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(tag, wl, "FAILED", out.stderr[-400:])
        continue
    d = json.loads(line[-1])
    st = d.get("stage_ms", {})
    print(tag, wl, round(d["ms_per_step"], 3), "fwd", st.get("composite_fwd"), "bwd", st.get("composite_bwd"), "bin",
          st.get("binning"), "pre", st.get("preprocess"), "preb", st.get("preprocess_bwd"), flush=True)
