#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02x
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mmq2.py tests/test_gpu_batch.py tests/test_gpu_serve.py -q -m gpu > $OUT/01_pytest.log 2>&1; grep -E "passed|failed" $OUT/01_pytest.log | tail -2; grep "^FAILED" $OUT/01_pytest.log | head -5
timeout 300 python tools/mmq2_bench.py --child 5 8 32 2>&1 | tee $OUT/02_mmq_small_n.log
MINIGPT4_MMQ2_W1=0 timeout 300 python tools/mmq2_bench.py --child 8 2>&1 | tee -a $OUT/02_mmq_small_n.log
for B in 4 5 8 16 32; do timeout 300 python tools/batch_decode.py $B 48 2>&1 | tail -1 | tee -a $OUT/03_batch.log; done
MINIGPT4_BATCH_ROWS_MAX=0 timeout 300 python tools/batch_decode.py 4 48 2>&1 | tail -1 | tee -a $OUT/03_batch.log
MINIGPT4_BATCH_ROWS_MAX=0 timeout 300 python tools/batch_decode.py 3 48 2>&1 | tail -1 | tee -a $OUT/03_batch.log
