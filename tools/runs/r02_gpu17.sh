#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02p
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mmq2.py -q -m gpu -x > $OUT/01_pytest_mmq2.log 2>&1; tail -4 $OUT/01_pytest_mmq2.log
timeout 300 python tools/mmq2_bench.py --child 142 512 2>&1 | tee $OUT/02_mmq_bench.log
MINIGPT4_MMQ2_TT=2 timeout 300 python tools/mmq2_bench.py --child 142 512 2>&1 | tee $OUT/02_mmq_bench_tt2.log
for n in 142 512; do timeout 200 python bench_prefill.py --config 13b --tokens $n > $OUT/03_prefill_q_$n.json 2> $OUT/03_prefill_q_$n.err; cut -c1-120 $OUT/03_prefill_q_$n.json; done
MINIGPT4_MMQ_SCALED=0 timeout 200 python bench_prefill.py --config 13b --tokens 142 2>/dev/null | cut -c1-120
MINIGPT4_MMQ2_TT=2 timeout 200 python bench_prefill.py --config 13b --tokens 142 2>/dev/null | cut -c1-120
