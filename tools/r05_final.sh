#!/bin/bash
# round-5 measurement run: the complete GPU suite, smoke, default bench line (N = 1, all legs), eager kernel trace + FETCH_SIZE pass of the decode loop,
# kernel tables of the batched step (B = 4) and of one image encode
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05z
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --tb=short -rs 2>&1 | tee $OUT/01_pytest_gpu.log | tail -4
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/02_bench.json 2> $OUT/02_bench.err; tail -2 $OUT/02_bench.err; python -c "
import json;d=json.load(open('$OUT/02_bench.json'));print({k:d[k] for k in ['value','ms_per_step','prefill_ms','image_encode_ms','image_encode_device_ms','model_load_s','parity_mode_tokens_per_s']}); r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r.get('timing'), r['frac'], r.get('kernel_sum_ms_per_token'), d['ms_per_step'], r.get('traffic'), r['whole_step'])
for k in r['kernel_table']: print('  ', k['kernel'][:50].ljust(50), k['calls_per_token'], k['avg_us'], k.get('timing'))
print(d.get('parity')); print(d.get('cpu_baseline',{}).get('value'), d.get('batched_decode'), d.get('image_encode_batched'), d.get('long_context')); print(d.get('configs')); print(d.get('configs3_share_per_gpu'))"
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_decode -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --no-cpu-baseline --no-extra-configs --conversations 0 --no-long-context > $GRAFT_REPO_ROOT/$OUT/03_rocprof_bench.log 2>&1 )
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extra-configs --conversations 0 --no-long-context > $GRAFT_REPO_ROOT/$OUT/03_pmc_fetch.log 2>&1 )
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_b4 -- python $GRAFT_REPO_ROOT/tools/batch_decode.py 4 > $GRAFT_REPO_ROOT/$OUT/04_prof_b4.log 2>&1 )
for d in prof_decode prof_b4; do f=$(ls $OUT/$d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/${d}_kernel_stats.csv; done
f=$(ls $OUT/pmc_fetch/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py "$f" $OUT/pmc_fetch_summary.csv "rocprofv3 --kernel-trace --pmc FETCH_SIZE, eager decode loop of bench.py (13B Q5_K_M), round 5" && head -6 $OUT/pmc_fetch_summary.csv | cut -c1-200
bash tools/encode_trace.sh $OUT/enc | tail -3
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
ls $OUT
