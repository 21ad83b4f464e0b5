#!/bin/bash
# Round-6 measurement set on one MI355X box (everything the DESIGN.md / BASELINE.md tables of this round cite):
#   1. the whole GPU test tier; 2. the default bench line; 3. rocprofv3 --kernel-trace --stats of the decode loop (kernel durations behind roofline.frac);
#   4. the per-batch-size encode tables (kernel trace + FETCH_SIZE pass, B = 1 only and B = 4 only).
#   tools/r06_final.sh [tag]     -> gpurun_out/<tag>_*
set -u
export TMPDIR=/tmp
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/${TAG}_pytest_gpu.log; tail -3 gpurun_out/${TAG}_pytest_gpu.log
python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; tail -c 600 gpurun_out/${TAG}_bench_n1.json; echo
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_decode_prof -- python $R/bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-extra-configs --conversations 0 --no-long-context > $R/gpurun_out/${TAG}_decode_prof.log 2>&1 )
F=$(find gpurun_out/${TAG}_decode_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" gpurun_out/${TAG}_decode_kernel_stats.csv && head -5 gpurun_out/${TAG}_decode_kernel_stats.csv | cut -c1-160
find gpurun_out/${TAG}_decode_prof -name "*.db" -delete; find gpurun_out/${TAG}_decode_prof -name "*kernel_trace.csv" -delete
tools/encode_pmc.sh gpurun_out/${TAG}_encpmc > gpurun_out/${TAG}_encpmc.log 2>&1
head -12 gpurun_out/${TAG}_encpmc/encode_kernel_table_b1.txt; head -10 gpurun_out/${TAG}_encpmc/encode_kernel_table_b4.txt
