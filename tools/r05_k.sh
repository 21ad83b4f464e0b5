#!/bin/bash
# round-5 run K: exp / SiLU computed instead of gathered in the decode step (tests + A/B), B = 4 bench
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05k
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --tb=short -k "not mul_mat_matches and not gemm" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q --tb=short -k "13b_l2 or 7b" 2>&1 | tail -5
timeout 600 python tools/ab_decode.py --steps 128 "computed" "tables:MINIGPT4_COMPUTED_TABLES=0" "computed_again" "tables_again:MINIGPT4_COMPUTED_TABLES=0" 2>&1 | tee $OUT/ab_decode_tables.log | tail -5
timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --no-long-context --conversations 4 > $OUT/bench_B4.json 2> $OUT/bench_B4.err; python -c "
import json;d=json.load(open('$OUT/bench_B4.json'));b=d['batched_decode'];print('B=4', round(d['value'],1), round(b['tokens_per_s_per_gpu'],1), round(b['ms_per_step'],3))"
