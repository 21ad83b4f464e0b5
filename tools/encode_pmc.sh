#!/bin/bash
# Per-batch-size profile of the image encode (13B-shaped vision file): for B = 1 ONLY and B = 4 ONLY, each in its own process,
#   (a) rocprofv3 --kernel-trace --stats  -> launches and microseconds per encode, per kernel symbol;
#   (b) rocprofv3 --pmc FETCH_SIZE (own pass, --kernel-trace only: MI355X_MICROARCH.md HBM section) -> bytes fetched per launch = FETCH_SIZE[KB] x 1024 x 2 (gfx950).
# The round-5 record averaged B = 1 and B = 4 launches of one symbol into one figure (verdict, missing #4).
#   tools/encode_pmc.sh <out-dir under gpurun_out> [library.so]
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/encpmc}; LIB=${2:-}
mkdir -p $OUT
[ -n "$LIB" ] && export MINIGPT4_LIBRARY=$LIB
R=$GRAFT_REPO_ROOT
for B in 1 4; do
  if [ $B = 1 ]; then ARGS="8 0"; N=8; else ARGS="0 4"; N=5; fi
  ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_b$B -- python $R/bench_encode.py $ARGS > $R/$OUT/stats_b$B.log 2>&1 )
  ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_b$B -- python $R/bench_encode.py $ARGS > $R/$OUT/pmc_b$B.log 2>&1 )
  tail -1 $OUT/stats_b$B.log
  python3 - "$OUT" $B $N <<'PY'
import csv, glob, sys
from collections import defaultdict
out, B, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
st = glob.glob(f"{out}/stats_b{B}/**/*kernel_stats.csv", recursive=True)
pm = glob.glob(f"{out}/pmc_b{B}/**/*counter_collection.csv", recursive=True)
fetch = defaultdict(lambda: [0, 0.0])
if pm:
    for r in csv.DictReader(open(pm[0])):
        if r["Counter_Name"] == "FETCH_SIZE":
            a = fetch[r["Kernel_Name"].split("(")[0]]; a[0] += 1; a[1] += float(r["Counter_Value"])
lines, tot = [], 0.0
if st:
    for r in csv.DictReader(open(st[0])):
        name = r["Name"].split("(")[0]
        if "k_repack" in name or "fillBuffer" in name: continue
        us = float(r["TotalDurationNs"]) / 1e3 / n
        tot += us
        f = fetch.get(name)
        mb = f"{f[1] / f[0] * 1024 * 2 / 1e6:9.2f} MB fetched/launch" if f and f[0] else "        -"
        lines.append(f"{int(r['Calls']) / n:7.1f} launches/encode {float(r['AverageNs']) / 1e3:8.2f} us avg {us:9.1f} us/encode {mb}  {name}")
open(f"{out}/encode_kernel_table_b{B}.txt", "w").write(f"# B = {B} images per encode pass ({n} passes); fetched = FETCH_SIZE x 1024 x 2 from a separate --pmc pass of the same command\n" + "\n".join(lines) + f"\nsum of kernel durations per encode pass: {tot:.1f} us\n")
print("\n".join(lines[:12])); print(f"B={B} sum {tot:.1f} us")
PY
done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
