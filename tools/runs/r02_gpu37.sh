#!/bin/bash
# k_matvec_tn_mix (mixed-type qkv of the batched step in one launch): tests, then B = 2 / 3 / 4 on the 13B file with and without it
set -u
timeout 600 python -m pytest tests/test_gpu_batch.py -q -m gpu -x -s 2>&1 | grep -E "passed|failed|Error|B=" | tail -12
for B in 2 3 4; do timeout 200 python tools/batch_decode.py $B 64 2>/dev/null | tail -1; done
for B in 2 4; do MINIGPT4_BATCH_MIX=0 timeout 200 python tools/batch_decode.py $B 64 2>/dev/null | tail -1; done
