#!/bin/bash
# round-5 run N: first weight fetch ahead of the activation staging + batched staging loads in k_matvec_ri: tests, per-launch table, B = 4 / 3
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r05n
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q --tb=short 2>&1 | tail -3
timeout 300 python tools/batched_shapes_bench.py 2>&1 | tee $OUT/batched_shapes.log | tail -10
for B in 4 3 4; do
timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --no-long-context --conversations $B > $OUT/bench_B$B.json 2> $OUT/bench_B$B.err; python -c "
import json;d=json.load(open('$OUT/bench_B$B.json'));b=d['batched_decode'];print('B=$B', round(d['value'],1), round(b['tokens_per_s_per_gpu'],1), round(b['ms_per_step'],3))" | tee -a $OUT/bench.log
done
