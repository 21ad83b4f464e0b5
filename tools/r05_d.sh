#!/bin/bash
# round-5 run D: pair-packed K = 5120 mat-vec (tests + decode A/B), row-interleaved MFMA batched decode (tests + A/B), parity attention opt-in
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05d
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q --tb=short 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_paritymode.py -x -q --tb=short -k "llm_parity or chat_flow or oracle_order" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --tb=short -k "decode_matvec or llm_logits or odd_vocab or decode_loop" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q --tb=short -k "13b_l2" 2>&1 | tail -8
timeout 600 python tools/ab_decode.py --steps 128 "pack1" "pack0:MINIGPT4_MV_PACK=0" "pack1_again" 2>&1 | tee $OUT/ab_decode_pack.log | tail -5
for v in 1 0; do MINIGPT4_RI=$v timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --no-long-context > $OUT/bench_ri$v.json 2> $OUT/bench_ri$v.err; python -c "
import json;d=json.load(open('$OUT/bench_ri$v.json'));print('RI=$v', d['value'], d['batched_decode'])"; done
