#!/bin/bash
# round-5 run W: HBM traffic (FETCH_SIZE) and matrix-pipe busy counters of the batched (B = 4) decode step's kernels
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r05w
mkdir -p $OUT
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/tools/batch_decode.py 4 > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1 )
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_mfma -- python $GRAFT_REPO_ROOT/tools/batch_decode.py 4 > $GRAFT_REPO_ROOT/$OUT/pmc_mfma.log 2>&1 )
f=$(ls $OUT/pmc_fetch/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py "$f" $OUT/pmc_fetch_b4_summary.csv "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/batch_decode.py 4 (13B Q5_K_M, eager steps), round 5" && head -12 $OUT/pmc_fetch_b4_summary.csv | cut -c1-170
f=$(ls $OUT/pmc_mfma/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py "$f" $OUT/pmc_mfma_b4_summary.csv "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- tools/batch_decode.py 4, round 5" && head -12 $OUT/pmc_mfma_b4_summary.csv | cut -c1-220
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
