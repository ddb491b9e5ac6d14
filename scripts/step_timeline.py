"""Timeline of one training step from a rocprofv3 kernel trace (production schedule): per kernel start / end relative to
the step start, the queue it ran on, and how much of the step had only one queue busy.
python scripts/step_timeline.py <kernel_trace.csv> [--all]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
qs = sorted({r["Queue_Id"] for r in step})
main_q = max(qs, key=lambda q: sum(1 for r in step if r["Queue_Id"] == q))
ev = []
for r in step:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    ev.append((s, e, r["Queue_Id"] == main_q, r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]))
end = max(e for _, e, _, _ in ev)
# occupancy by sampling
N = 4000
both = only_main = only_side = idle = 0
for i in range(N):
    t = end * (i + 0.5) / N
    m = any(s <= t < e for s, e, mq, _ in ev if mq)
    sd = any(s <= t < e for s, e, mq, _ in ev if not mq)
    if m and sd: both += 1
    elif m: only_main += 1
    elif sd: only_side += 1
    else: idle += 1
print(f"step {end/1e3:.1f} us: both queues {100*both/N:.1f}%  main only {100*only_main/N:.1f}%  side only {100*only_side/N:.1f}%  idle {100*idle/N:.1f}%")
if "--all" in sys.argv:
    for s, e, mq, n in ev:
        print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f}  {'M' if mq else 's'}  {n}")
