#!/bin/bash
# round-4 measurement run: smoke, default bench line (N = 1, incl. the configs[1] / configs[4] legs), eager kernel trace + FETCH_SIZE pass of the decode loop, parity-mode trace
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04z
mkdir -p $OUT
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $OUT/02_bench.json 2> $OUT/02_bench.err; tail -2 $OUT/02_bench.err; python -c "
import json;d=json.load(open('$OUT/02_bench.json'));print({k:d[k] for k in ['value','ms_per_step','prefill_ms','image_encode_ms','image_encode_device_ms','model_load_s','parity_mode_tokens_per_s']}); r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r.get('timing'), r['frac'], r.get('kernel_sum_ms_per_token'), d['ms_per_step'], r.get('traffic'), r['whole_step'])
for k in r['kernel_table']: print('  ', k['kernel'][:50].ljust(50), k['calls_per_token'], k['avg_us'], k.get('timing'))
print(d.get('parity')); print(d.get('cpu_baseline',{}).get('value'), d.get('batched_decode'), d.get('image_encode_batched'), d.get('long_context')); print(d.get('configs'))"
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_decode -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --no-cpu-baseline --no-extra-configs --conversations 0 --no-long-context > $GRAFT_REPO_ROOT/$OUT/03_rocprof_bench.log 2>&1 )
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extra-configs --conversations 0 --no-long-context > $GRAFT_REPO_ROOT/$OUT/03_pmc_fetch.log 2>&1 )
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_parity -- python $GRAFT_REPO_ROOT/tools/parity_speed.py --config 13b --steps 24 > $GRAFT_REPO_ROOT/$OUT/04_parity.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
ls $OUT/prof_decode/*/ $OUT/pmc_fetch/*/ $OUT/prof_parity/*/ 2>/dev/null | head -20
