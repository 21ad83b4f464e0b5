#!/bin/bash
# all-wave entry / end stamps: is the launch-to-launch time of the multi-row mat-vec inside the kernel (late waves) or between kernels?
set -u
MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so timeout 300 python tools/timeline.py q5_k 13824 5120 2 2  q5_k 13824 5120 2 12  q5_k 13824 5120 2 14  q5_k 13824 5120 2 22 q5_k 5120 13824 1 12 2>&1 | grep -E "variant|results stored|LAST|entry"
