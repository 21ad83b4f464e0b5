#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02h
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu > $OUT/01_pytest_gpu.log 2>&1; tail -12 $OUT/01_pytest_gpu.log
timeout 400 python bench.py > $OUT/02_bench.json 2> $OUT/02_bench.err; tail -3 $OUT/02_bench.err; python -c "
import json;d=json.load(open('$OUT/02_bench.json'));print({k:d[k] for k in ['value','prefill_ms','image_encode_ms','model_load_s']}); print(json.dumps(d['roofline'])[:3000]); print(d.get('parity')); print(d.get('cpu_baseline',{}).get('value'))"
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_decode -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --no-cpu-baseline --conversations 0 > $GRAFT_REPO_ROOT/$OUT/03_rocprof_bench.log 2>&1 )
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --conversations 0 > $GRAFT_REPO_ROOT/$OUT/03_pmc_fetch.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT/pmc_fetch/*/ | head
