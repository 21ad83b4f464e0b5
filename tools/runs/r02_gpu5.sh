#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02e
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mmq2.py -q -x > $OUT/01_pytest_mmq2.log 2>&1; tail -3 $OUT/01_pytest_mmq2.log
timeout 300 python tools/mmq2_bench.py --child 142 512 > $OUT/02_mmq2_bench.log 2>&1; cat $OUT/02_mmq2_bench.log
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc1 -- python $GRAFT_REPO_ROOT/tools/mmq2_bench.py --child 142 > $GRAFT_REPO_ROOT/$OUT/03_pmc1.log 2>&1 )
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc2 -- python $GRAFT_REPO_ROOT/tools/mmq2_bench.py --child 142 > $GRAFT_REPO_ROOT/$OUT/03_pmc2.log 2>&1 )
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_LEVEL_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc3 -- python $GRAFT_REPO_ROOT/tools/mmq2_bench.py --child 142 > $GRAFT_REPO_ROOT/$OUT/03_pmc3.log 2>&1 )
timeout 900 python -m pytest tests -q -m gpu -x -v > $OUT/04_pytest_gpu.log 2>&1; grep -v PASSED $OUT/04_pytest_gpu.log | tail -25
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
