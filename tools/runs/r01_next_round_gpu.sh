#!/bin/bash
# One GPU call that answers the questions round 1 left open (run as the gpurun command, from the repo root; ~6 minutes of box time):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/next_round_gpu.sh'
# BEFORE the call, in the dev container:  python -c "import __graft_entry__ as g; g.build()" && make -C minigpt4.cpp_amd/csrc -j8 variants   (~8 min: start it in the background)
# Everything lands in gpurun_out/next_round/ (copy what should be judged into profiles/).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/next_round
mkdir -p $OUT
L=minigpt4.cpp_amd
echo "== 1. GPU parity suite (incl. the never-run tests/test_gpu_zq3k.py)" | tee $OUT/00_order.txt
timeout 300 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $OUT/01_pytest_gpu.log
echo "== 2. in-kernel timelines: default, PRIME2, batched mat-vec, cache-resident" | tee -a $OUT/00_order.txt
MINIGPT4_LIBRARY=$L/libminigpt4_tl.so timeout 60 python tools/timeline.py q5_k 5120 5120 3 2 q5_k 13824 5120 2 2 q5_k 5120 13824 1 1 q5_k 13824 5120 2 3 > $OUT/02_timeline_default.log 2>&1
MINIGPT4_LIBRARY=$L/libminigpt4_p2tl.so timeout 60 python tools/timeline.py q5_k 5120 5120 3 2 q5_k 13824 5120 2 2 q5_k 5120 13824 1 1 q5_k 13824 5120 2 3 > $OUT/02_timeline_prime2.log 2>&1
TL_SETS=1 MINIGPT4_LIBRARY=$L/libminigpt4_tl.so timeout 60 python tools/timeline.py q5_k 13824 5120 2 2 > $OUT/02_timeline_cache_resident.log 2>&1
echo "== 3. decode / prefill A/B (tok_s, prefill_ms, sig must match base)" | tee -a $OUT/00_order.txt
timeout 400 python tools/ab_decode.py base tailq2:MINIGPT4_TAILQ=2 p2:LIB=$L/libminigpt4_p2.so p3:LIB=$L/libminigpt4_p3.so \
    p2f95:LIB=$L/libminigpt4_p2.so,MINIGPT4_FUSE=95 p2f127:LIB=$L/libminigpt4_p2.so,MINIGPT4_FUSE=127 \
    tt1:LIB=$L/libminigpt4_tt1.so tt1apf:LIB=$L/libminigpt4_tt1apf.so base > $OUT/03_ab_decode.log 2>&1
echo "== 4. image encode A/B (embedding signature must match)" | tee -a $OUT/00_order.txt
timeout 200 python tools/ab_encode.py base v2:MINIGPT4_ATTN_MFMA=2 base > $OUT/04_ab_encode.log 2>&1
MINIGPT4_ATTN_MFMA=2 timeout 200 python -m pytest tests -q -m gpu -k "encode or image or vision or chat_flow" 2>&1 | tail -5 > $OUT/04_pytest_attn_v2.log
echo "== 5. headline bench + kernel trace" | tee -a $OUT/00_order.txt
timeout 300 python bench.py > $OUT/05_bench.json 2> $OUT/05_bench.err
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --no-cpu-baseline --conversations 0 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_bench.log 2>&1 )
echo "== 6. BASELINE configs[4]: 13B f16, 512-token prefill (26 GB file)" | tee -a $OUT/00_order.txt
timeout 400 python bench_prefill.py > $OUT/06_bench_prefill.json 2> $OUT/06_bench_prefill.err
tail -n +1 $OUT/01_pytest_gpu.log $OUT/03_ab_decode.log $OUT/04_ab_encode.log $OUT/05_bench.json | tail -60
