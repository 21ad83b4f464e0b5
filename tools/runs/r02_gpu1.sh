#!/bin/bash
# Round-2 first GPU call: full GPU suite (incl. the headline-shape parity tests), round-1's waiting A/B arms, bench + kernel trace, prefill baselines.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r02_gpu1.sh'
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02a
mkdir -p $OUT
L=minigpt4.cpp_amd
date +%s > $OUT/t0
echo "== 1. GPU suite" | tee $OUT/00_order.txt
timeout 900 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^\[" | tail -40 > $OUT/01_pytest_gpu.log
echo "== 2. timelines" | tee -a $OUT/00_order.txt
MINIGPT4_LIBRARY=$L/libminigpt4_tl.so timeout 60 python tools/timeline.py q5_k 5120 5120 3 2 q5_k 13824 5120 2 2 q5_k 5120 13824 1 1 q5_k 13824 5120 2 3 > $OUT/02_timeline_default.log 2>&1
MINIGPT4_LIBRARY=$L/libminigpt4_p2tl.so timeout 60 python tools/timeline.py q5_k 5120 5120 3 2 q5_k 13824 5120 2 2 q5_k 5120 13824 1 1 q5_k 13824 5120 2 3 > $OUT/02_timeline_prime2.log 2>&1
echo "== 3. decode / prefill A/B" | tee -a $OUT/00_order.txt
timeout 500 python tools/ab_decode.py base tailq2:MINIGPT4_TAILQ=2 p2:LIB=$L/libminigpt4_p2.so p3:LIB=$L/libminigpt4_p3.so \
    p2f95:LIB=$L/libminigpt4_p2.so,MINIGPT4_FUSE=95 p2f127:LIB=$L/libminigpt4_p2.so,MINIGPT4_FUSE=127 p3f95:LIB=$L/libminigpt4_p3.so,MINIGPT4_FUSE=95 \
    tt1:LIB=$L/libminigpt4_tt1.so tt1apf:LIB=$L/libminigpt4_tt1apf.so base > $OUT/03_ab_decode.log 2>&1
echo "== 4. encode A/B" | tee -a $OUT/00_order.txt
timeout 200 python tools/ab_encode.py base v2:MINIGPT4_ATTN_MFMA=2 base > $OUT/04_ab_encode.log 2>&1
echo "== 5. bench + kernel traces" | tee -a $OUT/00_order.txt
timeout 400 python bench.py > $OUT/05_bench.json 2> $OUT/05_bench.err
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_decode -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --no-cpu-baseline --conversations 0 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_bench.log 2>&1 )
echo "== 6. prefill baselines (int8 MFMA, 142 / 512 rows) + trace" | tee -a $OUT/00_order.txt
timeout 200 python bench_prefill.py --config 13b --tokens 142 > $OUT/06_prefill_q_142.json 2> $OUT/06_prefill_q_142.err
timeout 200 python bench_prefill.py --config 13b --tokens 512 > $OUT/06_prefill_q_512.json 2> $OUT/06_prefill_q_512.err
( cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_prefill -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b --tokens 142 --reps 2 > $GRAFT_REPO_ROOT/$OUT/06_rocprof_prefill.log 2>&1 )
echo "== 7. configs[4]: 13B f16 512-token prefill" | tee -a $OUT/00_order.txt
timeout 500 python bench_prefill.py > $OUT/07_prefill_f16.json 2> $OUT/07_prefill_f16.err
date +%s > $OUT/t1
# keep the merged output small: the stats CSVs only
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
tail -n +1 $OUT/01_pytest_gpu.log $OUT/03_ab_decode.log $OUT/04_ab_encode.log $OUT/05_bench.json $OUT/06_prefill_q_142.json $OUT/06_prefill_q_512.json $OUT/07_prefill_f16.json | tail -120
