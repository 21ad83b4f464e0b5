#!/bin/bash
# round 3: prefill measurements with the fp16-MFMA prompt attention: kernel trace of the 142-row and 512-row Q5_K_M passes, and BASELINE configs[4] (13B f16, 512 tokens)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03p
mkdir -p $OUT
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_q5k_512 -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b --tokens 512 --reps 4 > $GRAFT_REPO_ROOT/$OUT/q5k_512.log 2>&1 )
timeout 300 python bench_prefill.py --config 13b --tokens 142 2>/dev/null | tail -1 > $OUT/q5k_142.json
timeout 900 python bench_prefill.py --config 13b-f16 --tokens 512 2> $OUT/f16_512.err | tail -1 > $OUT/f16_512.json
MINIGPT4_ATTN_PREFILL_F16=0 timeout 600 python bench_prefill.py --config 13b-f16 --tokens 512 2>/dev/null | tail -1 > $OUT/f16_512_attn_f32.json
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
cut -c1-400 $OUT/q5k_142.json $OUT/f16_512.json $OUT/f16_512_attn_f32.json; tail -2 $OUT/f16_512.err
