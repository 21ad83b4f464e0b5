#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02s
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -q -m gpu > $OUT/01_pytest.log 2>&1; grep -E "passed|failed" $OUT/01_pytest.log | tail -2; grep "^FAILED" $OUT/01_pytest.log | head -5
for q in 256 0; do echo -n "13b q5k 512 rows QS2=$q: "; MINIGPT4_ATTN_QS2=$q timeout 200 python bench_prefill.py --config 13b --tokens 512 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],2))"; done 2>&1 | tee $OUT/02_qs2.log
echo -n "7b 142: "; timeout 200 python bench_prefill.py --config 7b --tokens 142 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],2))"
for q in 256 0; do echo -n "13b f16 512 QS2=$q: "; MINIGPT4_ATTN_QS2=$q timeout 400 python bench_prefill.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['roofline']['frac'],4))"; done 2>&1 | tee -a $OUT/02_qs2.log
