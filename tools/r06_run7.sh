python tools/attn_qt_bench.py 1 2 3 4 8
for Q in 0 2; do echo "== encoder MINIGPT4_ATTN_QT=$Q"; MINIGPT4_ATTN_QT=$Q python bench_encode.py 8 4 2>&1 | grep -E "encode ms|batched"; done
python bench_encode.py 0 2 2>&1 | grep -E "batched"
python bench_encode.py 0 8 2>&1 | grep -E "batched"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_paritymode.py tests/test_gpu_quantized_vision.py tests/test_gpu_goldens.py -x -q -m gpu -k "encode or image or vision or attn or gemm or vit or golden" 2>&1 | tail -5
python -m pytest tests/test_gpu_headline.py -x -q -m gpu -k "vit_g" 2>&1 | tail -3
