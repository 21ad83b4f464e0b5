python -m pytest tests/test_gpu_serve.py -x -q -m gpu -k "stream_timeout" 2>&1 | grep -v "^$" | tail -45 > gpurun_out/r06_t3.log
tools/gemm_fetch_sweep.sh gpurun_out/gemmfetch 257 4224 1408 0 "0 37 3 4 5 11 31 23 20 32" > gpurun_out/gemmfetch_257.log 2>&1
tools/gemm_fetch_sweep.sh gpurun_out/gemmfetch 1028 4224 1408 0 "0 37 4 5 31 23 32 34 35 38" > gpurun_out/gemmfetch_1028.log 2>&1
tools/gemm_fetch_sweep.sh gpurun_out/gemmfetch6144 257 6144 1408 5 "0 37 5 31 4" > gpurun_out/gemmfetch_fc1_257.log 2>&1
cat gpurun_out/r06_t3.log | tail -30; cat gpurun_out/gemmfetch_257.log gpurun_out/gemmfetch_1028.log gpurun_out/gemmfetch_fc1_257.log
