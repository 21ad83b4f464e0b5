#!/bin/bash
# Per-kernel table of one image encode (13B-shaped vision file, B = 1): rocprofv3 --kernel-trace --stats of bench_encode.py, launches and microseconds per encode.
#   tools/encode_trace.sh <out-dir under gpurun_out> [library.so]
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/enc}; LIB=${2:-}
mkdir -p $OUT
[ -n "$LIB" ] && export MINIGPT4_LIBRARY=$LIB
N=8
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench_encode.py $N > $GRAFT_REPO_ROOT/$OUT/bench.log 2>&1 )
tail -1 $OUT/bench.log
python3 - "$OUT" $N <<'PY'
import csv, glob, sys
out, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + "/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = 0.0
lines = []
for r in rows:
    name = r["Name"].split("(")[0]
    if "k_repack" in name or "fillBuffer" in name: continue
    us = float(r["TotalDurationNs"]) / 1e3 / n
    tot += us
    lines.append(f"{int(r['Calls']) / n:7.1f} launches/encode {float(r['AverageNs']) / 1e3:8.2f} us avg {us:9.1f} us/encode  {name}")
open(out + "/encode_kernel_table.txt", "w").write("\n".join(lines) + f"\nsum of kernel durations per encode: {tot:.1f} us\n")
print("\n".join(lines[:14])); print(f"sum {tot:.1f} us")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
