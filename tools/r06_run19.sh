for rep in 1 2; do for X in 0 1; do echo "== MINIGPT4_SKINNY_MT2=$X"; MINIGPT4_SKINNY_MT2=$X python bench_encode.py 0 4 2>&1 | grep -E "batched"; MINIGPT4_SKINNY_MT2=$X python bench_encode.py 0 8 2>&1 | grep -E "batched"; done; done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "table_gather" 2>&1 | tail -3
