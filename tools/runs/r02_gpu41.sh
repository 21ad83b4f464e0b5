#!/bin/bash
# 2-row multi-row mat-vec: 512 / 640 / 768-thread workgroups (2 / 2.5 / 3 waves per SIMD) on the w1|w3 and qkv shapes, prepared rows and in-launch preparation
set -u
for T in 512 640 768; do echo "--- MINIGPT4_TN_THREADS=$T"; MINIGPT4_TN_THREADS=$T MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so timeout 300 python tools/timeline.py q5_k 13824 5120 2 12  q5_k 13824 5120 2 22 q5_k 5120 5120 3 22 q5_k 5120 5120 1 12 2>&1 | grep -E "variant|LAST wave of the workgroup done"; done
