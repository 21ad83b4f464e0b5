#!/bin/bash
# config #5 (13B f16 weights, 512-token prefill): MFMA-busy counters per kernel (SQ_VALU_MFMA_BUSY_CYCLES / per-XCD GRBM_GUI_ACTIVE), in their own pass with --kernel-trace only
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02_45; mkdir -p $OUT
( cd /tmp && timeout -k 5 330 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_mfma_f16 -- python $GRAFT_REPO_ROOT/bench_prefill.py --reps 1 > $GRAFT_REPO_ROOT/$OUT/01_pmc.log 2>&1 )
tail -2 $OUT/01_pmc.log | cut -c1-300
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
ls $OUT/pmc_mfma_f16/*/ | head
