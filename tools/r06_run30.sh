for rep in 1 2; do for X in 0 1; do echo "== MINIGPT4_COMPUTED_GELU=$X"; MINIGPT4_COMPUTED_GELU=$X python bench_encode.py 8 4 2>&1 | grep -E "encode ms|batched"; done; done
MINIGPT4_COMPUTED_GELU=1 python bench_encode.py 0 8 2>&1 | grep batched; MINIGPT4_COMPUTED_GELU=0 python bench_encode.py 0 8 2>&1 | grep batched
python -m pytest tests/test_gpu_parity.py tests/test_gpu_paritymode.py tests/test_gpu_quantized_vision.py tests/test_gpu_goldens.py tests/test_gpu_serve.py -x -q -m gpu -k "encode or image or vision or attn or gemm or vit or golden or fold or fresh or round6 or table" 2>&1 | tail -4
python -m pytest tests/test_gpu_headline.py -x -q -m gpu -k "vit_g" 2>&1 | tail -3; cat gpurun_out/parity_observed_vision_13b.json
