#!/bin/bash
# kernel trace of the 3-D training step: per-dispatch durations of the last step in launch order (gpurun_out/trace3d_last.txt)
# usage: scripts/trace_3d.sh [extra bench args]
set -o pipefail
OUT=gpurun_out/prof3d
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
MI_NO_SIDE_STREAM=1 timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -o r --output-format csv -- python3 bench.py --dims 3 --size 96 --batch 4 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-reference-network "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last c1_s2_fwd forward launch whose predecessor chain starts a step (the memset precedes it)
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "c1_s2_fwd" in n]
# every step launches c1_s2_fwd twice (forward, backward); the forward one opens the step
i0 = starts[-2]
sel = rows[i0:]
t0 = int(sel[0]["Start_Timestamp"])
out = []
prev_end = t0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    out.append(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  grid {r.get('Grid_Size_X','?'):>8} wg {r.get('Workgroup_Size_X','?'):>4} lds {r.get('LDS_Block_Size','?'):>7}  {n[:90]}")
    prev_end = e
open("gpurun_out/trace3d_last.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:8])); print("...", len(out), "dispatches, last ends at", (prev_end - t0) / 1e3, "us")
PY
