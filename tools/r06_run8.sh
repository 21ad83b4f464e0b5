for Q in 1 0; do for B in 2 4 8; do echo "== MINIGPT4_ATTN_QT=$Q B=$B"; MINIGPT4_ATTN_QT=$Q python bench_encode.py 0 $B 2>&1 | grep -E "batched"; done; done
echo "== B=1 (reduce / layernorm size classes)"; python bench_encode.py 8 0 2>&1 | grep -E "encode ms"
