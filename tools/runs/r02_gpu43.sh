#!/bin/bash
# repacked Q4_K / Q5_K header: decode and batched decode on the 13B file (before: 369.2-369.9 tok/s; B = 2 / 4: 618 / 876)
set -u
OUT=gpurun_out/r02_43; mkdir -p $OUT
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));print('decode', d['value'], 'prefill', d['prefill_ms'], 'encode', d['image_encode_ms']); r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r['frac']); print(d.get('batched_decode'))
for k in r['kernel_table']: print('  ', k['kernel'][:50].ljust(50), k['calls_per_token'], k['avg_us'])"
timeout 200 python tools/batch_decode.py 2 64 2>/dev/null | tail -1
