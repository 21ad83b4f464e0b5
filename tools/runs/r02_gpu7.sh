#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02g
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mmq2.py -q -x > $OUT/01_pytest_mmq2.log 2>&1; tail -5 $OUT/01_pytest_mmq2.log
timeout 300 python tools/mmq2_bench.py --child 142 512 > $OUT/02_mmq2_bench.log 2>&1
MINIGPT4_MMQ3_TT=2 timeout 300 python tools/mmq2_bench.py --child 142 512 >> $OUT/02_mmq2_bench.log 2>&1
KS=1 timeout 300 python tools/mmq2_bench.py --child 142 >> $OUT/02_mmq2_bench.log 2>&1
cat $OUT/02_mmq2_bench.log
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc1 -- python $GRAFT_REPO_ROOT/tools/mmq2_bench.py --child 142 > $GRAFT_REPO_ROOT/$OUT/03_pmc1.log 2>&1 )
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py tests/test_gpu_batch.py -q -x > $OUT/04_pytest.log 2>&1; tail -4 $OUT/04_pytest.log
for n in 142 512; do timeout 200 python bench_prefill.py --config 13b --tokens $n > $OUT/05_prefill_q_$n.json 2> $OUT/05_prefill_q_$n.err; cut -c1-120 $OUT/05_prefill_q_$n.json; done
( cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_prefill142 -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b --tokens 142 --reps 2 > $GRAFT_REPO_ROOT/$OUT/05_rocprof_prefill142.log 2>&1 )
timeout 400 python bench.py > $OUT/06_bench.json 2> $OUT/06_bench.err; tail -3 $OUT/06_bench.err; python -c "
import json;d=json.load(open('$OUT/06_bench.json'));print({k:d[k] for k in ['value','prefill_ms','image_encode_ms','model_load_s']}); print(json.dumps(d['roofline'])[:1500])"
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
