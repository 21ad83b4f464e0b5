python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "round6" 2>&1 | tail -3
for S in 4 2 3 6 8; do echo "== MINIGPT4_SPLITK_FC2=$S"; MINIGPT4_SPLITK_FC2=$S python bench_encode.py 8 4 2>&1 | grep -E "encode ms|batched"; done
