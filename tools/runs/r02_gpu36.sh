#!/bin/bash
# 24-bit scale multiplies in the k-quant unit dots + wide k_batch_finish + Q3_K vision files: parity subset, then decode / batched decode on the 13B file
set -u
OUT=gpurun_out/r02_36; mkdir -p $OUT
timeout 500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_quantized_vision.py tests/test_gpu_zq3k.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));print('decode', d['value'], 'prefill', d['prefill_ms'], 'encode', d['image_encode_ms']); r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r['frac']); print(d.get('batched_decode'))"
for B in 2 3; do timeout 200 python tools/batch_decode.py $B 64 2>/dev/null | tail -1; done
