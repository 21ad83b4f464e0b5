#!/bin/bash
# round-5 run H: tests after the reverts; kernel table of the batched decode step (B = 4) under rocprofv3
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q --tb=short -k "row_interleaved" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --tb=short -k "decode_loop or chat_flow or temperature" 2>&1 | tail -4
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_b4 -- python $GRAFT_REPO_ROOT/tools/batch_decode.py 4 > $GRAFT_REPO_ROOT/$OUT/prof_b4.log 2>&1 )
f=$(ls $OUT/prof_b4/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then cp "$f" $OUT/b4_kernel_stats.csv; head -25 "$f" | cut -c1-200; fi
tail -3 $OUT/prof_b4.log
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
