#!/bin/bash
# where do the waves of the decode mat-vec kernels spend their cycles?  SQ counters per wave: single row (variant 2), 2 rows (12), 4 rows (14) on the w1|w3 shape
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02_46; mkdir -p $OUT
( cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc1 -- python $GRAFT_REPO_ROOT/tools/timeline.py q5_k 13824 5120 2 2 q5_k 13824 5120 2 12 q5_k 13824 5120 2 14 > $GRAFT_REPO_ROOT/$OUT/01.log 2>&1 )
( cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc2 -- python $GRAFT_REPO_ROOT/tools/timeline.py q5_k 13824 5120 2 2 q5_k 13824 5120 2 12 q5_k 13824 5120 2 14 > $GRAFT_REPO_ROOT/$OUT/02.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
for p in pmc1 pmc2; do python tools/pmc_per_wave.py $OUT/$p/*/*counter_collection.csv matvec; tail -2 $OUT/0${p#pmc}.log | cut -c1-200; done
