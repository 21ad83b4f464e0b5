#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02r
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mmq2.py tests/test_gpu_parity.py -q -m gpu > $OUT/01_pytest.log 2>&1; grep -E "passed|failed" $OUT/01_pytest.log | tail -2; grep "^FAILED" $OUT/01_pytest.log | head -5
for f in 200 150 125 100 75; do
  for n in 142 512; do echo -n "13b fill=$f n=$n: "; MINIGPT4_MMQ2_FILL=$f timeout 200 python bench_prefill.py --config 13b --tokens $n 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],2))"; done
done 2>&1 | tee $OUT/02_fill_sweep.log
for f in 200 125 75; do for n in 142; do echo -n "7b fill=$f n=$n: "; MINIGPT4_MMQ2_FILL=$f timeout 200 python bench_prefill.py --config 7b --tokens $n 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],2))"; done; done 2>&1 | tee -a $OUT/02_fill_sweep.log
