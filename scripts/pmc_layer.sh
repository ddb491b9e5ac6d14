#!/bin/bash
# SQ counters of the conv kernels of one layer: scripts/pmc_layer.sh "<bench_ops filter>"  (output gpurun_out/pmc_layer/)
set -o pipefail
OUT=gpurun_out/pmc_layer
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf $OUT/p1 $OUT/p2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM -d $OUT/p1 -o pmc --output-format csv -- python3 scripts/bench_ops.py "$1" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
python3 scripts/pmc_by_kernel.py "$(find $OUT/p1 -name '*counter_collection.csv' | head -1)" > $OUT/sq1.csv
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU -d $OUT/p2 -o pmc --output-format csv -- python3 scripts/bench_ops.py "$1" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 scripts/pmc_by_kernel.py "$(find $OUT/p2 -name '*counter_collection.csv' | head -1)" > $OUT/sq2.csv
rm -rf $OUT/p1 $OUT/p2
cat $OUT/sq1.csv $OUT/sq2.csv
