#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02q
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu > $OUT/01_pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/01_pytest_gpu.log | tail -3; grep "^FAILED" $OUT/01_pytest_gpu.log | head
for B in 2 3 4; do timeout 200 python tools/batch_decode.py $B 64 2>&1 | tail -1 | tee -a $OUT/04_batch.log; done
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_batch2 -- python $GRAFT_REPO_ROOT/tools/batch_decode.py 2 48 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_batch2.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r02q/prof_batch2/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:12]: print("  ", r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
