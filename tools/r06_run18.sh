for rep in 1 2; do for X in 0 1; do echo "== MINIGPT4_REDUCE_V4=$X"; MINIGPT4_REDUCE_V4=$X python bench_encode.py 8 4 2>&1 | grep -E "encode ms|batched"; done; done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_paritymode.py tests/test_gpu_quantized_vision.py tests/test_gpu_goldens.py tests/test_gpu_serve.py -x -q -m gpu -k "encode or image or vision or attn or gemm or vit or golden or fold or fresh or round6" 2>&1 | tail -4
python -m pytest tests/test_gpu_headline.py -x -q -m gpu -k "vit_g" 2>&1 | tail -3
