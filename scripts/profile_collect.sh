#!/bin/bash
# GPU-side collection for profiles/: usage scripts/profile_collect.sh STAGE   (outputs under gpurun_out/prof/)
#   pmc   : FETCH_SIZE and WRITE_SIZE passes (one counter per pass)
#   sq    : one pass of eight SQ counters on a single stream
#   trace : kernel traces (single stream and production schedule) and bench.py's own tables
#   alt   : the secondary workloads (3-D training, sliding-window inference, the 32-512 channel widths of
#           reference/train.py:82, 8 slices of 256 x 256): kernel stats, bench JSON, per-layer table
set -o pipefail
STAGE=${1:-trace}
OUT=gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$STAGE" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_$c
    timeout -s KILL 200 rocprofv3 --pmc $c -d $OUT/pmc_$c -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-reference-network --no-secondary > $OUT/pmc_$c.log 2>&1 || exit 1
    cp "$(find $OUT/pmc_$c -name '*counter_collection.csv' | head -1)" $OUT/pmc_$c.csv || exit 1
    rm -rf $OUT/pmc_$c
  done
  ls -la $OUT
elif [ "$STAGE" = sq ]; then
  rm -rf $OUT/sq
  MI_NO_SIDE_STREAM=1 timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/sq -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-reference-network --no-secondary > $OUT/sq.log 2>&1 || exit 1
  cp "$(find $OUT/sq -name '*counter_collection.csv' | head -1)" $OUT/pmc_sq.csv || exit 1
  rm -rf $OUT/sq
elif [ "$STAGE" = level0 ]; then
  # issue / wait / store-path counters per kernel (three passes: SQ, TCC write side, TCP stalls); read with
  # scripts/pmc_by_kernel.py for the first-layer convolution and the 16-channel scatter layer against the norm kernels
  P="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-reference-network --no-secondary"
  for pass in sq tcc tcp; do
    if [ $pass = sq ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD";
    elif [ $pass = tcc ]; then C="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum";
    else C="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum"; fi
    rm -rf $OUT/l0_$pass
    MI_NO_SIDE_STREAM=1 timeout -s KILL 200 rocprofv3 --pmc $C -d $OUT/l0_$pass -o pmc --output-format csv -- python3 bench.py $P > $OUT/l0_$pass.log 2>&1 || { tail -5 $OUT/l0_$pass.log; exit 1; }
    python3 scripts/pmc_by_kernel.py "$(find $OUT/l0_$pass -name '*counter_collection.csv' | head -1)" > $OUT/pmc_l0_$pass.csv || exit 1
    rm -rf $OUT/l0_$pass
  done
  head -30 $OUT/pmc_l0_sq.csv
elif [ "$STAGE" = alt ]; then
  for w in 3d infer ref b8; do
    if [ $w = 3d ]; then ARGS="--dims 3 --size 96 --batch 4"; elif [ $w = ref ]; then ARGS="--channels 32,64,128,256,512"; elif [ $w = b8 ]; then ARGS="--size 256 --batch 8"; else ARGS="--mode infer"; fi
    rm -rf $OUT/a_$w
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/a_$w -o r --output-format csv -- python3 bench.py $ARGS --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-reference-network --no-secondary > $OUT/trace_$w.log 2>&1 || exit 1
    cp "$(find $OUT/a_$w -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_$w.csv || exit 1
    rm -rf $OUT/a_$w
    BENCH_LAYERS=1 timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-reference-network > $OUT/bench_$w.json 2> $OUT/per_layer_$w.txt || exit 1
    cat $OUT/bench_$w.json
  done
else
  rm -rf $OUT/t1 $OUT/t2
  MI_NO_SIDE_STREAM=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/t1 -o r --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-reference-network --no-secondary > $OUT/trace1.log 2>&1 || exit 1
  cp "$(find $OUT/t1 -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/t2 -o r --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-reference-network --no-secondary > $OUT/trace2.log 2>&1 || exit 1
  cp "$(find $OUT/t2 -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_overlapped.csv || exit 1
  rm -rf $OUT/t1 $OUT/t2
  BENCH_LAYERS=1 timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/per_layer.txt || exit 1
  cat $OUT/bench.json
fi
