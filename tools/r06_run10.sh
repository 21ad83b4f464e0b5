# round 6: (a) upper bound on what ANY change to the k-quant prompt mat-mul's activation staging could gain (diagnostic build without the staging after the first super-block);
# (b) the 8-rank rehearsal of bench.py on one GPU; (c) the new 8-rank native test
echo "== product kernels"; GENS=2 python tools/mmq2_bench.py 142 512 2>&1 | grep "^N="
echo "== MG4_MMQ2_NOSTAGE build (activation staging removed after the first super-block: wrong results, timing only)"
MINIGPT4_LIBRARY=$GRAFT_REPO_ROOT/minigpt4.cpp_amd/libminigpt4_ns_test.so GENS=2 python tools/mmq2_bench.py 142 512 2>&1 | grep "^N="
python -m pytest tests/test_gpu_serve.py -x -q -m gpu -k "real_ranks" 2>&1 | tail -3
/usr/bin/time -v -o gpurun_out/r06_rehearsal8.time env MG4_BENCH_REHEARSAL=1 python bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/r06_rehearsal8.json 2> gpurun_out/r06_rehearsal8.err
echo "rehearsal rc=$?"; grep -E "Elapsed|Maximum resident" gpurun_out/r06_rehearsal8.time; tail -c 1500 gpurun_out/r06_rehearsal8.json; tail -5 gpurun_out/r06_rehearsal8.err
