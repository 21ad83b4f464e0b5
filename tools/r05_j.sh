#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q --tb=short -k "row_interleaved" 2>&1 | tail -3
timeout 400 python tools/batched_shapes_bench.py 2>&1 | tee $OUT/batched_shapes.log | tail -9
for B in 4; do timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --no-long-context --conversations $B > $OUT/bench_B$B.json 2> $OUT/bench_B$B.err; python -c "
import json;d=json.load(open('$OUT/bench_B$B.json'));b=d['batched_decode'];print('B=$B', round(d['value'],1), round(b['tokens_per_s_per_gpu'],1), round(b['ms_per_step'],3), d.get('configs3_share_per_gpu'))"; done
