#!/bin/bash
# round-5 run U: 7B-specific experiment: two rows per group (R = 2) for the K = 4096 mat-vecs (NU = 2): configs[1] decode, alternating libraries
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r05u
mkdir -p $OUT
for i in 1 2; do for L in libminigpt4.so libminigpt4_r2.so; do
MINIGPT4_LIBRARY=/root/repo/minigpt4.cpp_amd/$L timeout 300 python bench.py --config 7b --steps 128 --no-cpu-baseline --no-extra-configs --no-long-context --conversations 0 > $OUT/b.json 2> $OUT/b.err; python -c "
import json;d=json.load(open('$OUT/b.json'));print('$L', round(d['value'],1), round(d['ms_per_step'],4))" | tee -a $OUT/ab_7b_r2.log
done; done
