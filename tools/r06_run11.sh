echo "== MG4_MMQ2_NOSTAGE build (activation staging removed after the first super-block: wrong results, timing only)"
MINIGPT4_LIBRARY=$GRAFT_REPO_ROOT/minigpt4.cpp_amd/libminigpt4_ns_test.so GENS=2 python tools/mmq2_bench.py 142 512 2>&1 | grep "^N="
T0=$(date +%s)
MG4_BENCH_REHEARSAL=1 python bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/r06_rehearsal8.json 2> gpurun_out/r06_rehearsal8.err
echo "rehearsal rc=$? wall_s=$(( $(date +%s) - T0 ))" | tee gpurun_out/r06_rehearsal8.time
tail -c 1200 gpurun_out/r06_rehearsal8.json; tail -5 gpurun_out/r06_rehearsal8.err
