#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02l
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu > $OUT/01_pytest_gpu.log 2>&1; tail -8 $OUT/01_pytest_gpu.log
timeout 300 python tools/mmq2_bench.py --child 142 512 2>&1 | tee $OUT/02_mmq_bench.log
for n in 142 512; do timeout 200 python bench_prefill.py --config 13b --tokens $n > $OUT/03_prefill_q_$n.json 2> $OUT/03_prefill_q_$n.err; cut -c1-160 $OUT/03_prefill_q_$n.json; done
MINIGPT4_MMQ_GEN=2 timeout 200 python bench_prefill.py --config 13b --tokens 142 2>/dev/null | cut -c1-120
timeout 300 python tools/ab_encode.py vit 2>&1 | tee $OUT/04_ab_encode.log
( cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_prefill142 -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b --tokens 142 --reps 2 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_prefill142.log 2>&1 )
( cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_encode -- python $GRAFT_REPO_ROOT/bench_encode.py 8 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_encode.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
for d in ("prof_prefill142","prof_encode"):
    for f in glob.glob(f"gpurun_out/r02l/{d}/*/*kernel_stats.csv"):
        print(d)
        for r in list(csv.DictReader(open(f)))[:12]: print("  ", r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
