#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02c
mkdir -p $OUT
echo "== 1. the aborting test, unfiltered" | tee $OUT/00_order.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "mul_mat_matches or gemm_f16" > $OUT/01_pytest_mulmat.log 2>&1
tail -30 $OUT/01_pytest_mulmat.log
echo "== 2. mmq2 micro-benchmark + ablations" | tee -a $OUT/00_order.txt
timeout 300 python tools/mmq2_bench.py 142 512 > $OUT/02_mmq2_bench.log 2>&1
KS=1 timeout 100 python tools/mmq2_bench.py --child 142 >> $OUT/02_mmq2_bench.log 2>&1
cat $OUT/02_mmq2_bench.log
echo "== 3. PMC" | tee -a $OUT/00_order.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > $OUT/03_counters.txt
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc1 -- python $GRAFT_REPO_ROOT/tools/mmq2_bench.py --child 142 > $GRAFT_REPO_ROOT/$OUT/03_pmc1.log 2>&1 )
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc2 -- python $GRAFT_REPO_ROOT/tools/mmq2_bench.py --child 142 > $GRAFT_REPO_ROOT/$OUT/03_pmc2.log 2>&1 )
echo "== 4. f16 prefill with the 128x128 GEMM" | tee -a $OUT/00_order.txt
timeout 400 python bench_prefill.py > $OUT/04_prefill_f16.json 2> $OUT/04_prefill_f16.err
cut -c1-300 $OUT/04_prefill_f16.json
( cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_f16 -- python $GRAFT_REPO_ROOT/bench_prefill.py --reps 1 > $GRAFT_REPO_ROOT/$OUT/04_rocprof_f16.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +30M -delete
