"""Reads a rocprofv3 kernel-trace csv and prints, for the last steps of the run, each kernel's start/end
relative to the first and the time with no extraction kernel (kpe_bs) in flight.
usage: python tools/trace_timeline.py <kernel_trace.csv> [n_last_kernels]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-last:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{a / 1e3:9.1f} {b / 1e3:9.1f} {(b - a) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3} {r['Kernel_Name'][:70]}")
# time without kpe_bs over the last 10 kpe launches
kp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "kpe_bs" in r["Kernel_Name"]][-11:]
if len(kp) > 2:
    span = kp[-1][0] - kp[0][0]
    busy, cur_e = 0, kp[0][0]
    for s, e in kp[:-1]:
        s = max(s, cur_e)
        if e > s:
            busy += e - s
            cur_e = e
    print(f"over the last {len(kp) - 1} steps: {span / (len(kp) - 1) / 1e3:.1f} us per step, kpe_bs in flight {busy / (len(kp) - 1) / 1e3:.1f} us per step")
