# round 3: the F16 language model's set launches re-routed through k_gemm_dma (256x128 tiles, K split to ~1 workgroup per CU): tests, the four launches of a 13B layer at 512 rows, 256x256 arms, config #5
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_goldens.py -x -q -k "f16 or gemm or batched_image" 2>&1 | tail -2
timeout 300 python tools/timeline_gemm.py 512 15360 5120 0 103 512 5120 5120 2 101 512 27648 5120 0 102 512 5120 13824 2 101 512 27648 5120 0 34 512 27648 5120 0 38 512 27648 5120 0 39 512 15360 5120 0 38 512 15360 5120 0 39 1028 1408 1408 2 0 1028 4224 1408 0 0 > gpurun_out/mb_gemm6.txt 2>&1
cat gpurun_out/mb_gemm6.txt
timeout 600 python bench_prefill.py --config 13b-f16 --tokens 512 2>/dev/null | tail -1 | cut -c1-200
MINIGPT4_GEMM_ARM=-3 timeout 600 python bench_prefill.py --config 13b-f16 --tokens 512 2>/dev/null | tail -1 | cut -c1-200
