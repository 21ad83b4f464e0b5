#!/bin/bash
# round-5 run A: the new Q8_0 / F16 decode mat-vec instantiations, the odd-vocabulary chat flow, the 13B-width n_vocab 32001 cut, then the default bench line with the new legs
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_matvec_variants or odd_vocabulary or narrow_unit or full_size_matvec or llm_logits" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_headline.py -x -q -k "v32001" 2>&1 | tail -5
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));print({k:d[k] for k in ['value','ms_per_step','prefill_ms','image_encode_ms','image_encode_device_ms','model_load_s','parity_mode_tokens_per_s']})
for k,v in d.get('configs',{}).items(): print(k, {a:b for a,b in v.items() if a!='workload'})"
