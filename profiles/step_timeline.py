"""One training step as a kernel timeline from a rocprofv3 --kernel-trace CSV.
Usage: python profiles/step_timeline.py <kernel_trace.csv> [step_index_from_end=3]
Prints start [us after the previous composite_bwd ended], duration [us], queue and kernel name of every dispatch up to and
including the next composite_bwd."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
name = lambda r: r.get("Kernel_Name") or r.get("kernel_name")
st = lambda r: int(r.get("Start_Timestamp") or r.get("start_timestamp"))
en = lambda r: int(r.get("End_Timestamp") or r.get("end_timestamp"))
q = lambda r: r.get("Queue_Id") or r.get("queue_id") or "?"
rows.sort(key=st)
bw = [i for i, r in enumerate(rows) if "composite_bwd" in name(r)]
a, b = bw[-back - 1], bw[-back]
t0 = en(rows[a])
queues = {}
for r in rows[a + 1:b + 1]:
    qq = queues.setdefault(q(r), f"s{len(queues)}")
    short = re.sub(r"\(anonymous namespace\)::|void ", "", name(r)).split("(")[0][:48]
    print(f"{(st(r) - t0) / 1e3:9.1f} {(en(r) - st(r)) / 1e3:8.1f}  {qq}  {short}")
print(f"# step = {(en(rows[b]) - t0) / 1e3:.1f} us from end of composite_bwd to end of the next composite_bwd")
