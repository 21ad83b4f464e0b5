#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02i
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_serve.py tests/test_gpu_parity.py -q -m gpu -x -k "arena or recv or profile or single_embedding" > $OUT/01_pytest.log 2>&1; tail -4 $OUT/01_pytest.log
timeout 400 python bench.py --no-cpu-baseline --conversations 0 > $OUT/02_bench.json 2> $OUT/02_bench.err; tail -2 $OUT/02_bench.err; python -c "
import json;d=json.load(open('$OUT/02_bench.json'));print({k:d[k] for k in ['value','prefill_ms','image_encode_ms']}); r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r['frac'], r.get('eager_ms_per_token_with_events'))
for k in r['kernel_table']: print('  ', k['kernel'][:50].ljust(50), k['calls_per_token'], k['avg_us'])"
for v in "base" "f119:MINIGPT4_FUSE=119" "f127:MINIGPT4_FUSE=127" "f95:MINIGPT4_FUSE=95"; do :; done
timeout 400 python tools/ab_decode.py --steps 96 base f119:MINIGPT4_FUSE=119 f127:MINIGPT4_FUSE=127 f95:MINIGPT4_FUSE=95 fat768:MINIGPT4_FAT_LB=768 > $OUT/03_ab_decode.log 2>&1; cat $OUT/03_ab_decode.log | cut -c1-200
for B in 2 4; do
  timeout 200 python tools/batch_decode.py $B 64 2>&1 | tail -1 | tee $OUT/04_batch_$B.json
done
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_batch4 -- python $GRAFT_REPO_ROOT/tools/batch_decode.py 4 48 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_batch4.log 2>&1 )
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_batch2 -- python $GRAFT_REPO_ROOT/tools/batch_decode.py 2 48 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_batch2.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
for d in ("prof_batch4","prof_batch2"):
    for f in glob.glob(f"gpurun_out/r02i/{d}/*/*kernel_stats.csv"):
        print(d)
        for r in list(csv.DictReader(open(f)))[:16]: print("  ", r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
