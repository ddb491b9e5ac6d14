#!/bin/bash
# usage: scripts/gpu_check.sh TAG  -- GPU tests, then the benchmark with the per-layer table (outputs under gpurun_out/)
TAG=${1:-x}
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_$TAG.log 2>&1
rc=$?
tail -5 gpurun_out/t_$TAG.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
BENCH_LAYERS=1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?
cat gpurun_out/bench_$TAG.json
exit $rc
