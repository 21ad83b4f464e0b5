#!/bin/bash
# Which tile shape of the image path's fp16 GEMM fetches what from beyond the L2?  For one (M, N, K) and a list of arms (launch_gemm_f16_arm in vision_kernels.hip):
#   pass 1 (no profiler): us per launch, back to back, weights cycled through HBM (tools/timeline_gemm.py);
#   pass 2 (rocprofv3 --pmc FETCH_SIZE, own pass): bytes per launch = FETCH_SIZE x 1024 x 2 (gfx950), per kernel symbol;
#   pass 3 (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum): L2 hit rate.
#   tools/gemm_fetch_sweep.sh <out-dir> M N K flags "arm arm ..."
set -u
export TMPDIR=/tmp
OUT=$1; M=$2; N=$3; K=$4; FL=$5; ARMS=$6
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
ARGS=""
for a in $ARMS; do ARGS="$ARGS $M $N $K $FL $a"; done
python $R/tools/timeline_gemm.py $ARGS > $OUT/us_M$M.log 2>&1
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/fetch_M$M -- python $R/tools/timeline_gemm.py $ARGS > $R/$OUT/fetch_M$M.log 2>&1 )
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/hit_M$M -- python $R/tools/timeline_gemm.py $ARGS > $R/$OUT/hit_M$M.log 2>&1 )
python3 - $OUT $M $N $K <<'PY'
import csv, glob, sys
from collections import defaultdict
out, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
def agg(pat):
    f = glob.glob(pat, recursive=True)
    a = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    if f:
        for r in csv.DictReader(open(f[0])):
            x = a[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]; x[0] += 1; x[1] += float(r["Counter_Value"])
    return a
fe, hi = agg(f"{out}/fetch_M{M}/**/*counter_collection.csv"), agg(f"{out}/hit_M{M}/**/*counter_collection.csv")
lines = [f"# M {M} N {N} K {K}: operands {(N * K + M * K) * 2 / 1e6:.2f} MB (W {N * K * 2 / 1e6:.2f} + A {M * K * 2 / 1e6:.2f}); fetched = FETCH_SIZE x 1024 x 2; the us-per-launch table (no profiler) is in us_M{M}.log"]
for k in sorted(fe, key=lambda k: -fe[k]["FETCH_SIZE"][1]):
    if "gemm" not in k: continue
    f = fe[k]["FETCH_SIZE"]; h = hi.get(k, {})
    hr = ""
    if h and h.get("TCC_HIT_sum") and h.get("TCC_MISS_sum"):
        hh, mm = h["TCC_HIT_sum"][1], h["TCC_MISS_sum"][1]
        hr = f"  L2 hit rate {hh / max(1.0, hh + mm):.3f}"
    lines.append(f"{f[0]:5d} launches {f[1] / f[0] * 2048 / 1e6:9.2f} MB fetched/launch{hr}  {k}")
open(f"{out}/fetch_table_M{M}.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
cat $OUT/us_M$M.log | grep "us per launch"
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
