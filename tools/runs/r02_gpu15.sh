#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02o
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu > $OUT/01_pytest_gpu.log 2>&1; tail -8 $OUT/01_pytest_gpu.log
for n in 142 512; do timeout 200 python bench_prefill.py --config 7b --tokens $n > $OUT/02_prefill_7b_$n.json 2> $OUT/02_prefill_7b_$n.err; cut -c1-140 $OUT/02_prefill_7b_$n.json; done
timeout 300 python bench.py --config 7b --no-cpu-baseline --steps 128 > $OUT/04_bench_7b.json 2> $OUT/04_bench_7b.err; python -c "
import json;d=json.load(open('$OUT/04_bench_7b.json'));print('7b', {k:d[k] for k in ['value','prefill_ms','image_encode_ms']}, d.get('batched_decode'))"
