for X in 1 2; do echo "== encoder MINIGPT4_SPLITK_PROJ=$X"; MINIGPT4_SPLITK_PROJ=$X python bench_encode.py 8 4 2>&1 | grep -E "encode ms|batched"; done
tools/encode_pmc.sh gpurun_out/encpmc2 > gpurun_out/encpmc2.log 2>&1; cat gpurun_out/encpmc2/encode_kernel_table_b1.txt; cat gpurun_out/encpmc2/encode_kernel_table_b4.txt | head -14
