# round 6: split-K workgroup -> XCD mapping A/B (micro-benchmark + FETCH_SIZE), encoder A/B, the stall test
python -m pytest tests/test_gpu_serve.py -x -q -m gpu -k "stream_timeout" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06_t4.log
tail -3 gpurun_out/r06_t4.log
SK="$((2 | (4 << 8)))"
for X in 0 1; do
  echo "== MG4_SPLITK_XCD=$X"
  export MG4_SPLITK_XCD=$X
  A=""
  for arm in 0 3 7 23 32 34; do A="$A 257 1408 6144 0 $((SK | (arm << 16))) 1028 1408 6144 0 $((SK | (arm << 16)))"; done
  python tools/timeline_gemm.py $A 2>&1 | grep "us per launch"
  ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/skfetch_$X -- python $GRAFT_REPO_ROOT/tools/timeline_gemm.py 257 1408 6144 0 $SK 1028 1408 6144 0 $((SK | (34 << 16))) > /dev/null 2>&1 )
  python tools/pmc_summary.py $(find gpurun_out/skfetch_$X -name "*counter_collection.csv" | head -1) gpurun_out/skfetch_$X.csv "MG4_SPLITK_XCD=$X: M 257 arm 0 (k_gemm_f16<128,64,64>) and M 1028 arm 34 (k_gemm_dma<256,128>) of the fc2 split-K GEMM" > /dev/null
  grep gemm gpurun_out/skfetch_$X.csv | cut -c1-200
  find gpurun_out/skfetch_$X -name "*.csv" -size +1M -delete; find gpurun_out/skfetch_$X -name "*.db" -delete
done
unset MG4_SPLITK_XCD
for X in 0 1; do echo "== encoder MINIGPT4_SPLITK_XCD=$X"; MINIGPT4_SPLITK_XCD=$X python bench_encode.py 8 4 2>&1 | grep -E "encode ms|batched"; done
