#!/bin/bash
# issue-priority toggle between the two waves of a SIMD (build -DMG4_PRIO_TOGGLE -DMG4_TIMELINE) vs the timeline build without it: w1|w3, qkv, wo, w2 shapes; 1, 2 and 4 rows
set -u
CASES="q5_k 13824 5120 2 2  q5_k 5120 5120 3 2  q5_k 5120 5120 1 2  q5_k 5120 13824 1 1  q5_k 13824 5120 2 22  q5_k 13824 5120 2 14"
for L in tl prio; do echo "--- libminigpt4_$L.so"; MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_$L.so timeout 200 python tools/timeline.py $CASES 2>&1 | grep -E "variant|results stored|LAST wave of the workgroup done"; done
