#!/bin/bash
# Round-2 GPU call 2: new prefill kernels (mmq2 + MFMA prefill attention): parity, full suite, prefill timings + kernel trace.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02b
mkdir -p $OUT
echo "== 1. mmq2 parity" | tee $OUT/00_order.txt
timeout 600 python -m pytest tests/test_gpu_mmq2.py -q -x 2>&1 | tail -15 > $OUT/01_pytest_mmq2.log
echo "== 2. GPU suite" | tee -a $OUT/00_order.txt
timeout 900 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^\[" | tail -40 > $OUT/02_pytest_gpu.log
echo "== 3. prefill timings" | tee -a $OUT/00_order.txt
for n in 142 512; do
  timeout 200 python bench_prefill.py --config 13b --tokens $n > $OUT/03_prefill_q_$n.json 2> $OUT/03_prefill_q_$n.err
  MINIGPT4_MMQ=1 MINIGPT4_ATTN_PREFILL=0 timeout 200 python bench_prefill.py --config 13b --tokens $n > $OUT/03_prefill_q_${n}_old.json 2> /dev/null
  MINIGPT4_ATTN_PREFILL=0 timeout 200 python bench_prefill.py --config 13b --tokens $n > $OUT/03_prefill_q_${n}_oldattn.json 2> /dev/null
done
for ks in 1; do MINIGPT4_MMQ2_KS=$ks timeout 200 python bench_prefill.py --config 13b --tokens 142 > $OUT/03_prefill_q_142_ks$ks.json 2> /dev/null; done
( cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_prefill142 -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b --tokens 142 --reps 2 > $GRAFT_REPO_ROOT/$OUT/03_rocprof_prefill142.log 2>&1 )
( cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_prefill512 -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b --tokens 512 --reps 2 > $GRAFT_REPO_ROOT/$OUT/03_rocprof_prefill512.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
tail -n +1 $OUT/01_pytest_mmq2.log $OUT/02_pytest_gpu.log | tail -60; for f in $OUT/03_prefill_q_*.json; do echo $f; cut -c1-160 $f; done
