#!/bin/bash
# attention-output quantisation inside the wo launch at 3 and 4 rows too (it already is at 2): batch tests, B = 3 / 4 decode (before: 739 / 874-876), whole GPU suite
set -u
timeout 300 python -m pytest tests/test_gpu_batch.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for B in 3 4; do timeout 200 python tools/batch_decode.py $B 64 2>/dev/null | tail -1; done
timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
