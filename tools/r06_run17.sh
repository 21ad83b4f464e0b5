for rep in 1 2; do for W in 0 1; do for B in 4 3; do MINIGPT4_RI_WO=$W python tools/batch_decode.py $B 96 2>&1 | tail -1 | cut -c1-140; done; done; done
MINIGPT4_RI_WO=1 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "configs3 and 13b_l2" 2>&1 | tail -3
