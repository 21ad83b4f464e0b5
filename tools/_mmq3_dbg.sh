timeout 200 python -m pytest tests/test_gpu_mmq3.py -x -q 2>&1 | tail -5
for e in 0; do echo "MMQ3_EXP=$e: $(MMQ3_EXP=$e GENS=3,2 KS=0 timeout 60 python tools/mmq2_bench.py --child 142 512 64 2>&1 | tail -3 | cut -c1-110)"; done
