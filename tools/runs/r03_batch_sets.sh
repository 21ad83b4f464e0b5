# round 3: batched decode of 5..32 conversations through set launches + deferred combines (MINIGPT4_BATCH_SETS=0: one launch + combine per matrix, as before)
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -2
for B in 5 8 16 32; do for v in 1 0; do echo "B=$B sets=$v $(MINIGPT4_BATCH_SETS=$v timeout 200 python tools/batch_decode.py $B 48 2>/dev/null | tail -1 | cut -c1-110)"; done; done
echo "B=4 rows_max=3 (MFMA path) $(MINIGPT4_BATCH_ROWS_MAX=3 timeout 200 python tools/batch_decode.py 4 48 2>/dev/null | tail -1 | cut -c1-110)"
echo "B=4 default $(timeout 200 python tools/batch_decode.py 4 48 2>/dev/null | tail -1 | cut -c1-110)"
