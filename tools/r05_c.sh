#!/bin/bash
# round-5 run C: broadcast failure paths, fp16-form test, parity attention opt-in, the batched-decode MFMA probe, MFMA-busy counters of the two prompt kernels
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05c
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_serve.py -x -q 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_mmq2.py -x -q -k "fp16_mfma_form" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_paritymode.py -x -q -k "llm_parity or chat_flow" 2>&1 | tail -4
timeout 400 python tools/batched_mfma_probe.py 2>&1 | tee $OUT/batched_mfma_probe.log | tail -12
( cd /tmp && GENS=4,2 timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_mmq -- python $GRAFT_REPO_ROOT/tools/mmq2_bench.py --child 142 > $GRAFT_REPO_ROOT/$OUT/pmc_mmq.log 2>&1 )
f=$(ls $OUT/pmc_mmq/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python tools/pmc_summary.py "$f" $OUT/pmc_mmq_summary.csv "GENS=4,2 tools/mmq2_bench.py --child 142 under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" && grep -E "mmq|kernel" $OUT/pmc_mmq_summary.csv | cut -c1-260; fi
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete; find $OUT -name "*kernel_trace.csv" -delete
