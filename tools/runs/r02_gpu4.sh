#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02d
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mmq2.py -q -x > $OUT/01_pytest_mmq2.log 2>&1; tail -5 $OUT/01_pytest_mmq2.log
timeout 300 python tools/mmq2_bench.py 142 512 > $OUT/02_mmq2_bench.log 2>&1; cat $OUT/02_mmq2_bench.log
timeout 900 python -m pytest tests -q -m gpu > $OUT/03_pytest_gpu.log 2>&1; tail -25 $OUT/03_pytest_gpu.log
for n in 142 512; do timeout 200 python bench_prefill.py --config 13b --tokens $n > $OUT/04_prefill_q_$n.json 2> $OUT/04_prefill_q_$n.err; cut -c1-120 $OUT/04_prefill_q_$n.json; done
( cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_prefill142 -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b --tokens 142 --reps 2 > $GRAFT_REPO_ROOT/$OUT/04_rocprof_prefill142.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
