#!/bin/bash
# kernel trace of the norm microbenchmark: reduce / apply passes separately, per shape
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in 16 32 64 128; do
  rm -rf gpurun_out/pn$c
  timeout -k 10 150 rocprofv3 --kernel-trace --stats -d gpurun_out/pn$c -o pn --output-format csv -- python3 scripts/bench_norm2.py $c > /dev/null 2>&1 || exit 1
  f=$(find gpurun_out/pn$c -name '*kernel_stats.csv' | head -1)
  echo "== c$c"; grep -i "norm_" "$f" | cut -d, -f1-4
done
