#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02j
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/01_pytest_gpu.log 2>&1; tail -6 $OUT/01_pytest_gpu.log
timeout 400 python bench.py --no-cpu-baseline --conversations 0 > $OUT/02_bench.json 2> $OUT/02_bench.err; tail -2 $OUT/02_bench.err; python -c "
import json;d=json.load(open('$OUT/02_bench.json'));print({k:d[k] for k in ['value','prefill_ms','image_encode_ms']}); r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r.get('avg_launch_us_markers'), r.get('timing'), r['frac'], r.get('kernel_sum_ms_per_token'), d['ms_per_step'])
for k in r['kernel_table']: print('  ', k['kernel'][:50].ljust(50), k['calls_per_token'], k['avg_us'], k.get('avg_us_markers'), k.get('timing'))"
timeout 300 python tools/ab_encode.py vit3 mfma1:MINIGPT4_ATTN_MFMA=1 2>&1 | tee $OUT/03_ab_encode.log
for B in 2 4; do
  for F in 1 0; do
    MINIGPT4_BATCH_FUSE=$F timeout 200 python tools/batch_decode.py $B 64 2>&1 | tail -1 | tee -a $OUT/04_batch.log
  done
done
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_batch4 -- python $GRAFT_REPO_ROOT/tools/batch_decode.py 4 48 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_batch4.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
for d in ("prof_batch4",):
    for f in glob.glob(f"gpurun_out/r02j/{d}/*/*kernel_stats.csv"):
        print(d)
        for r in list(csv.DictReader(open(f)))[:14]: print("  ", r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
