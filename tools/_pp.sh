export TMPDIR=/tmp
( cd /tmp && MINIGPT4_NO_GRAPH=1 timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_par2 -- python $GRAFT_REPO_ROOT/tools/parity_speed.py --config 13b --steps 24 > $GRAFT_REPO_ROOT/gpurun_out/r04_par2.log 2>&1 )
f=$(ls gpurun_out/r04_par2/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then head -16 "$f" | cut -c1-150; fi
find gpurun_out/r04_par2 -name "*.db" -delete; find gpurun_out/r04_par2 -name "*kernel_trace.csv" -delete
