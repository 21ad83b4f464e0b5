#!/bin/bash
# does the launch-to-launch time of the multi-row mat-vec follow the number of rows STORED (N) or the rows computed (TN)?  N = 1..4 prepared rows on the w1|w3 shape,
# timeline library (in-kernel span) and default library (launch-to-launch only)
set -u
MINIGPT4_LIBRARY=minigpt4.cpp_amd/libminigpt4_tl.so timeout 300 python tools/timeline.py q5_k 13824 5120 2 11  q5_k 13824 5120 2 12  q5_k 13824 5120 2 13  q5_k 13824 5120 2 14 q5_k 13824 5120 2 1 2>&1 | grep -E "variant|results stored|activation row ready"
echo "--- default library"
timeout 300 python tools/timeline.py q5_k 13824 5120 2 11  q5_k 13824 5120 2 12  q5_k 13824 5120 2 13  q5_k 13824 5120 2 14  q5_k 13824 5120 2 1  q5_k 13824 5120 2 2 q5_k 13824 5120 2 22 2>&1 | grep -E "variant"
