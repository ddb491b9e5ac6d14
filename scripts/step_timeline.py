"""Prints every kernel of one training step (from one loss reduction to the next: a backward pass, the optimizer, the
next forward pass) from a rocprofv3 kernel trace, with
start / end in microseconds from the step's first kernel and its queue, then the busy time per queue and the time in
which only one queue runs:  python scripts/step_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "loss_reduce" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
busy = {}
for r in step:
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:56]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r.get("Queue_Id", "?")
    busy.setdefault(q, []).append((s, e))
    print(f"{s:9.1f} {e:9.1f} {e - s:7.1f}  q{q:>3} {n}")
end = max(e for v in busy.values() for _, e in v)
print(f"step span {end:.1f} us")
for q, v in busy.items():
    print(f"queue {q}: {len(v)} kernels, busy {sum(e - s for s, e in v):.1f} us, first {v[0][0]:.1f}, last end {max(e for _, e in v):.1f}")
