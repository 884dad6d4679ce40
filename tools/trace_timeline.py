"""Timeline of the last blocks of a rocprofv3 --kernel-trace CSV: start / end (us) per kernel and queue."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
last = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -90:]
base = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = (int(r["Start_Timestamp"]) - base) / 1e3, (int(r["End_Timestamp"]) - base) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'][:60]}")
