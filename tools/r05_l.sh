#!/bin/bash
# round-5 run L: mixed-type row-interleaved launch (k_matvec_ri_mix): kernel test, batch tests, B = 3 / 4 bench
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r05l
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q --tb=short -rs 2>&1 | tail -8
for B in 4 3; do
timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --no-long-context --conversations $B > $OUT/bench_B$B.json 2> $OUT/bench_B$B.err; python -c "
import json;d=json.load(open('$OUT/bench_B$B.json'));b=d['batched_decode'];print('B=$B', round(d['value'],1), round(b['tokens_per_s_per_gpu'],1), round(b['ms_per_step'],3))"
done
MINIGPT4_BATCH_MIX=0 timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --no-long-context --conversations 4 > $OUT/bench_B4_nomix.json 2> $OUT/bench_B4_nomix.err; python -c "
import json;d=json.load(open('$OUT/bench_B4_nomix.json'));b=d['batched_decode'];print('B=4 BATCH_MIX=0', round(d['value'],1), round(b['tokens_per_s_per_gpu'],1), round(b['ms_per_step'],3))"
