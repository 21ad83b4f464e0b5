# round 3 (final tree): image-encode evidence -- kernel table B = 1 (rocprofv3 --kernel-trace --stats), MFMA-busy counters of a B = 1 + B = 4 run (own pass, --kernel-trace only)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03enc; mkdir -p $OUT
tools/encode_trace.sh $OUT/trace | tail -3
( cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_mfma -- python $GRAFT_REPO_ROOT/bench_encode.py 4 4 > $GRAFT_REPO_ROOT/$OUT/pmc_mfma.log 2>&1 )
tail -2 $OUT/pmc_mfma.log
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
ls $OUT/pmc_mfma/*/
