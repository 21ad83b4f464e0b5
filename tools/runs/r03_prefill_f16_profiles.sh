# round 3 (final tree): config #5 evidence -- kernel table and MFMA-busy counters of one 13B f16 512-token prompt pass (separate rocprofv3 passes)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03f16; mkdir -p $OUT
( cd /tmp && timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b-f16 --tokens 512 --reps 3 > $GRAFT_REPO_ROOT/$OUT/bench.log 2>&1 )
tail -1 $OUT/bench.log | cut -c1-120
( cd /tmp && timeout -k 5 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b-f16 --tokens 512 --reps 2 > $GRAFT_REPO_ROOT/$OUT/pmc.log 2>&1 )
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
ls $OUT/prof/*/ $OUT/pmc/*/
