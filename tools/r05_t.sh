#!/bin/bash
# round-5 run T: ViT / Q-Former attention with computed exponentials (no table DMA): vision parity tests, encode A/B
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r05t
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py tests/test_gpu_quantized_vision.py tests/test_gpu_goldens.py -x -q --tb=short -k "vision or encode or image or chat or embedding or vit or qformer" 2>&1 | tail -4
timeout 600 python tools/ab_encode.py "computed_exp" "table_exp:MINIGPT4_COMPUTED_TABLES=0" "computed_exp_again" "table_exp_again:MINIGPT4_COMPUTED_TABLES=0" 2>&1 | tee $OUT/ab_encode_exp.log
