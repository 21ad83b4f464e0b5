python tools/attn_qt_bench.py 1 2 4 8
for Q in 1 2 0; do echo "== encoder MINIGPT4_ATTN_QT=$Q"; MINIGPT4_ATTN_QT=$Q python bench_encode.py 8 4 2>&1 | grep -E "encode ms|batched"; done
