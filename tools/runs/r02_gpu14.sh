#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02n
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu > $OUT/01_pytest_gpu.log 2>&1; tail -8 $OUT/01_pytest_gpu.log
for n in 142 512; do timeout 200 python bench_prefill.py --config 7b --tokens $n > $OUT/02_prefill_7b_$n.json 2> $OUT/02_prefill_7b_$n.err; cut -c1-140 $OUT/02_prefill_7b_$n.json; done
MINIGPT4_MMQ=1 timeout 200 python bench_prefill.py --config 7b --tokens 142 2>/dev/null | cut -c1-120
timeout 500 python bench_prefill.py > $OUT/03_prefill_f16.json 2> $OUT/03_prefill_f16.err; cut -c1-500 $OUT/03_prefill_f16.json; tail -2 $OUT/03_prefill_f16.err
( cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_f16 -- python $GRAFT_REPO_ROOT/bench_prefill.py --reps 2 > $GRAFT_REPO_ROOT/$OUT/04_rocprof_f16.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r02n/prof_f16/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:10]: print("  ", r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
