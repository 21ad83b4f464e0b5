#!/bin/bash
# repacked Q4_K / Q5_K header (24-bit pair words): kernel times on the 13B shapes, then the parity tests that cover every consumer of the plane
set -u
MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so timeout 300 python tools/timeline.py q5_k 13824 5120 2 2  q5_k 5120 5120 3 2  q5_k 5120 13824 1 1  q5_k 5120 5120 1 1  q5_k 13824 5120 2 22  q5_k 13824 5120 2 14 2>&1 | grep -E "variant|LAST wave of the workgroup done"
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_mmq2.py tests/test_gpu_goldens.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
