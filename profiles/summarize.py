"""Summarise rocprofv3 CSV output (kernel stats / PMC counter collection) into small tracked files.
Usage: python profiles/summarize.py counters <run_counter_collection.csv> <out.csv> [kernel-substr ...]
"""
import csv
import re
import sys
from collections import defaultdict


def counters(path, out, filt):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if filt and not any(f in k for f in filt):
            continue
        short = re.split(r"[(]", k.replace("(anonymous namespace)::", "").replace("void ", ""))[0][:70].replace(",", ";")
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        n[short][r["Counter_Name"]] += 1
    with open(out, "w") as f:
        f.write("kernel,counter,dispatches,avg_per_dispatch\n")
        for k in sorted(acc):
            for c in sorted(acc[k]):
                f.write(f"{k},{c},{n[k][c]},{acc[k][c] / n[k][c]:.6g}\n")
    print(open(out).read())


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3], sys.argv[4:])
