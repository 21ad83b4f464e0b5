#!/bin/bash
# round-5 run F: batched decode with the measured policy (MFMA launches for sets of >= 128 row groups at B >= 3), RI on / off, B = 3 and 4
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05f
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q --tb=short -k "not row_interleaved and not multi_row" 2>&1 | tail -5
for B in 4 3; do for v in 1 0; do MINIGPT4_RI=$v timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --no-long-context --conversations $B > $OUT/bench_B${B}_ri$v.json 2> $OUT/bench_B${B}_ri$v.err; python -c "
import json;d=json.load(open('$OUT/bench_B${B}_ri$v.json'));b=d['batched_decode'];print('B=$B RI=$v', round(d['value'],1), round(b['tokens_per_s_per_gpu'],1), round(b['ms_per_step'],3))"; done; done
