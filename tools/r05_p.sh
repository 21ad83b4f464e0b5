#!/bin/bash
# round-5 run P: LDS fragment reads of the k-quant prompt kernel issued one step ahead: parity tests, per-launch micro-benchmark, image-turn prefill
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r05p
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mmq2.py -x -q --tb=short 2>&1 | tail -3
GENS=2 timeout 300 python tools/mmq2_bench.py 142 512 2>&1 | tee $OUT/mmq2_bench.log | tail -6
timeout 300 python bench_prefill.py --config 13b --tokens 142 --reps 5 2>$OUT/prefill142.err | tee $OUT/prefill142.json | cut -c1-200
timeout 300 python bench_prefill.py --config 13b --tokens 512 --reps 5 2>$OUT/prefill512.err | tee $OUT/prefill512.json | cut -c1-200
