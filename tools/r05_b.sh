#!/bin/bash
# round-5 run B: fp16-MFMA k-quant prompt kernel (tests, micro-benchmark, whole prompt pass A/B), Q-Former fold + cross K/V hoist (tests, encode A/B)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mmq2.py -x -q -k "mmq2" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mul_mat_matches or activation_quant or encode or llm_logits or chat_flow" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_paritymode.py -x -q -k "image or chat" 2>&1 | tail -4
timeout 300 python tools/mmq2_bench.py 142 512 2>&1 | tee $OUT/mmq_bench.log | tail -6
timeout 300 python tools/ab_encode.py "r4form:MINIGPT4_QF_FOLD=0,MINIGPT4_KV_HOIST=0" "fold_only:MINIGPT4_KV_HOIST=0" "hoist_only:MINIGPT4_QF_FOLD=0" "both" 2>&1 | tee $OUT/ab_encode.log | tail -6
for v in 1 0; do MINIGPT4_MMQH=$v timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --conversations 0 --no-long-context > $OUT/bench_mmqh$v.json 2> $OUT/bench_mmqh$v.err; python -c "
import json;d=json.load(open('$OUT/bench_mmqh$v.json'));print('MMQH=$v', {k:d[k] for k in ['value','prefill_ms','prefill_first_ms','image_encode_device_ms','parity_mode_tokens_per_s']})"; done
