#!/bin/bash
# round 6: FETCH_SIZE pass of the eager decode loop (the record bench.py's roofline.traffic cites: profiles/pmc_traffic.json) + the B = 4 batched step's kernel table
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06pmc; mkdir -p $OUT; R=$GRAFT_REPO_ROOT
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extra-configs --conversations 0 --no-long-context > $R/$OUT/pmc_fetch.log 2>&1 )
f=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py "$f" $OUT/pmc_fetch_summary.csv "rocprofv3 --kernel-trace --pmc FETCH_SIZE, eager decode loop of bench.py (13B Q5_K_M), round 6 (tools/r06_pmc.sh)" && head -6 $OUT/pmc_fetch_summary.csv | cut -c1-220
[ -n "$f" ] && python tools/roofline_from_profile.py pmc "$f" $OUT/pmc_traffic.json "tools/r06_pmc.sh: MINIGPT4_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extra-configs --conversations 0 --no-long-context" 2>&1 | tail -3
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_b4 -- python $R/tools/batch_decode.py 4 > $R/$OUT/prof_b4.log 2>&1 )
g=$(find $OUT/prof_b4 -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp "$g" $OUT/b4_kernel_stats.csv && head -12 $OUT/b4_kernel_stats.csv | cut -c1-160
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
