#!/bin/bash
# round-5 run E: parity opt-in fix, per-shape batched mat-vec timing (dot4 vs mfma), decode sanity after the pair-packing removal
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_paritymode.py -x -q --tb=short 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --tb=short -k "odd_vocab or long_context or key_split" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q --tb=short -k "13b_l2" 2>&1 | tail -6
timeout 400 python tools/batched_shapes_bench.py 2>&1 | tee $OUT/batched_shapes.log | tail -12
