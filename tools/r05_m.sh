#!/bin/bash
# round-5 run M: w2 on the K-split MFMA launch, A/B on one box (B = 4, alternating)
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r05m
mkdir -p $OUT
for i in 1 2 3; do for W2 in 0 1; do
MINIGPT4_RI_W2=$W2 timeout 400 python bench.py --steps 32 --no-cpu-baseline --no-extra-configs --no-long-context --conversations 4 > $OUT/b.json 2> $OUT/b.err; python -c "
import json;d=json.load(open('$OUT/b.json'));b=d['batched_decode'];print('B=4 RI_W2=$W2', round(d['value'],1), round(b['tokens_per_s_per_gpu'],1), round(b['ms_per_step'],3))" | tee -a $OUT/ab_w2.log
done; done
