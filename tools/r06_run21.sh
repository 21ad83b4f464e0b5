MINIGPT4_RI_TAIL=1 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "configs3" 2>&1 | tail -4
for rep in 1 2; do for W in 0 1; do MINIGPT4_RI_TAIL=$W python tools/batch_decode.py 4 96 2>&1 | tail -1 | cut -c1-140; done; done
