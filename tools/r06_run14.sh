echo "== encoder, k_attn_vit single-tile computed form bounded to 3 waves per SIMD (168 VGPR, 12 B scratch)"; python bench_encode.py 8 4 2>&1 | grep -E "encode ms|batched"
python bench_encode.py 0 2 2>&1 | grep -E "batched"; python bench_encode.py 0 8 2>&1 | grep -E "batched"
python tools/attn_qt_bench.py 1 4
echo "== 7B Q4_0 decode: SiLU * mul + quantisation in the w2 launch's prologue (MINIGPT4_FUSE bit 3)"
python tools/ab_decode.py --config 7b --steps 128 base:MINIGPT4_FUSE=87 w2pro:MINIGPT4_FUSE=95 base2:MINIGPT4_FUSE=87 w2pro2:MINIGPT4_FUSE=95 2>&1 | tail -8
