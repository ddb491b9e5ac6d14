"""Idle time between consecutive kernels of a rocprofv3 kernel trace: python scripts/gap_stats.py <kernel_trace.csv>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
rows = rows[len(rows) // 2:]          # second half: steady state
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
pos = sorted(g for g in gaps if g > 0)
print(f"kernels {len(rows)}  span {span/1e3:.1f} us  busy(sum of durations) {busy/1e3:.1f} us  idle between kernels {sum(pos)/1e3:.1f} us "
      f"({100*sum(pos)/span:.1f}% of span)  overlapping pairs {sum(1 for g in gaps if g < 0)}")
if pos:
    print(f"gap median {pos[len(pos)//2]/1e3:.2f} us  p90 {pos[int(len(pos)*0.9)]/1e3:.2f} us  max {pos[-1]/1e3:.2f} us")
durs = sorted(e - s for s, e, _ in rows)
print(f"kernel duration median {durs[len(durs)//2]/1e3:.2f} us  min {durs[0]/1e3:.2f} us")
