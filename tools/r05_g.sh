#!/bin/bash
# round-5 run G: MFMA batched mat-vec with in-launch row preparation (tests + A/B), argmax + advance in one launch (tests + decode A/B)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q --tb=short 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --tb=short -k "llm_logits or decode_loop or chat_flow or temperature or context_overflow" 2>&1 | tail -5
for cfg in "4 1 1" "4 1 0" "3 1 1" "2 1 1"; do set -- $cfg; MINIGPT4_RI=$2 MINIGPT4_RI_FUSE=$3 timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --no-long-context --conversations $1 > $OUT/bench_B$1_ri$2_f$3.json 2> $OUT/bench_B$1_ri$2_f$3.err; python -c "
import json;d=json.load(open('$OUT/bench_B$1_ri$2_f$3.json'));b=d['batched_decode'];print('B=$1 RI=$2 FUSE=$3', round(d['value'],1), round(b['tokens_per_s_per_gpu'],1), round(b['ms_per_step'],3))"; done
timeout 600 python tools/ab_decode.py --steps 128 "tail_fused" "tail_3launch:MINIGPT4_FUSE_TAIL=0" "tail_fused_again" 2>&1 | tee $OUT/ab_decode_tail.log | tail -4
