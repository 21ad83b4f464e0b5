#!/bin/bash
# round-5 run O: kernel table of the 13B f16 512-token prefill (configs[4]) + the same for the Q5_K_M 142-row image turn
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r05o
mkdir -p $OUT
timeout 300 python bench_prefill.py --reps 3 > $OUT/prefill_f16.json 2> $OUT/prefill_f16.err; tail -1 $OUT/prefill_f16.json | cut -c1-300
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_f16 -- python $GRAFT_REPO_ROOT/bench_prefill.py --reps 7 > $GRAFT_REPO_ROOT/$OUT/prof_f16.log 2>&1 )
f=$(ls $OUT/prof_f16/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/prefill_f16_kernel_stats.csv && head -24 "$f" | cut -c1-220
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
