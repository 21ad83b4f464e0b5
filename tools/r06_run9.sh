python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_pytest_gpu_a.log; tail -5 gpurun_out/r06_pytest_gpu_a.log
python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err; tail -c 3000 gpurun_out/r06_bench_a.json
