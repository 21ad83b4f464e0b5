#!/bin/bash
# in-kernel timeline of the multi-row mat-vec on the w1|w3 and qkv shapes of the 13B model: 1 row with prologue (k_matvec_v2), 2 rows prepared / in-launch, 4 rows prepared / in-launch
set -u
MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so timeout 300 python tools/timeline.py q5_k 13824 5120 2 2  q5_k 13824 5120 2 4  q5_k 13824 5120 2 5  q5_k 13824 5120 2 3  q5_k 13824 5120 2 6  q5_k 5120 5120 3 2  q5_k 5120 5120 3 5  q5_k 5120 13824 1 1  q5_k 5120 13824 1 4 2>&1 | tail -70
