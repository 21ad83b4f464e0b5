#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02k
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu > $OUT/01_pytest_gpu.log 2>&1; tail -8 $OUT/01_pytest_gpu.log
timeout 400 python bench.py --no-cpu-baseline --conversations 0 > $OUT/02_bench.json 2> $OUT/02_bench.err; tail -2 $OUT/02_bench.err; python -c "
import json;d=json.load(open('$OUT/02_bench.json'));print({k:d[k] for k in ['value','prefill_ms','image_encode_ms']}); r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r.get('timing'), r['frac'], r.get('kernel_sum_ms_per_token'), d['ms_per_step'])
for k in r['kernel_table']: print('  ', k['kernel'][:50].ljust(50), k['calls_per_token'], k['avg_us'], k.get('timing'))"
for B in 2 3 4; do
  timeout 200 python tools/batch_decode.py $B 64 2>&1 | tail -1 | tee -a $OUT/04_batch.log
done
( cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_encode -- python $GRAFT_REPO_ROOT/bench_encode.py 8 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_encode.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r02k/prof_encode/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:16]: print("  ", r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
