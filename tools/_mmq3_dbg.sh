timeout 200 python -m pytest tests/test_gpu_mmq3.py -x -q 2>&1 | tail -5
for nw in 4 8; do echo "NW=$nw"; MMQ3_NW=$nw GENS=3,2 KS=0 timeout 100 python tools/mmq2_bench.py --child 64 142 512 2>&1 | tail -3 | cut -c1-120; done
