"""Prints the kernels around the end of one training step from a rocprofv3 kernel trace (production schedule):
python scripts/tail_view.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
a = idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a - 14:a + 6]:
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:48]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f}  q{r.get('Queue_Id', '?'):>3} {n}")
