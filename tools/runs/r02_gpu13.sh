#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02m
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu > $OUT/01_pytest_gpu.log 2>&1; tail -8 $OUT/01_pytest_gpu.log
timeout 300 python tools/ab_decode.py --steps 96 base > $OUT/02_ab_decode.log 2>&1; cut -c1-200 $OUT/02_ab_decode.log
timeout 400 python bench.py > $OUT/03_bench.json 2> $OUT/03_bench.err; tail -2 $OUT/03_bench.err; python -c "
import json;d=json.load(open('$OUT/03_bench.json'));print({k:d[k] for k in ['value','prefill_ms','image_encode_ms','model_load_s']}); r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r.get('timing'), r['frac'], r.get('kernel_sum_ms_per_token'), d['ms_per_step'], r.get('traffic'))
for k in r['kernel_table']: print('  ', k['kernel'][:50].ljust(50), k['calls_per_token'], k['avg_us'], k.get('timing'))
print(d.get('parity')); print(d.get('cpu_baseline',{}).get('value'), d.get('batched_decode'))"
timeout 300 python bench.py --config 7b --no-cpu-baseline --steps 128 > $OUT/04_bench_7b.json 2> $OUT/04_bench_7b.err; python -c "
import json;d=json.load(open('$OUT/04_bench_7b.json'));print('7b', {k:d[k] for k in ['value','prefill_ms','image_encode_ms']}, d.get('batched_decode'))"
( cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_mfma_encode -- python $GRAFT_REPO_ROOT/bench_encode.py 4 > $GRAFT_REPO_ROOT/$OUT/05_pmc_mfma.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT/pmc_mfma_encode/*/ | head -5
