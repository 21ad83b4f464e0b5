# round 3: kernel trace of the 142-row image-turn prefill (13B Q5_K_M), per pass
set -u
export TMPDIR=/tmp
OUT=gpurun_out/p142; mkdir -p $OUT
REPS=4
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench_prefill.py --config 13b --tokens 142 --reps $REPS > $GRAFT_REPO_ROOT/$OUT/bench.log 2>&1 )
tail -1 $OUT/bench.log | cut -c1-160
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
lines = []
for r in rows[:18]:
    lines.append(f"{int(r['Calls']):6d} calls {float(r['AverageNs']) / 1e3:8.2f} us avg {float(r['TotalDurationNs']) / 1e6:9.3f} ms total {r['Percentage']:>6s}%  {r['Name'].split('(')[0][:90]}")
open(out + "/kernel_table.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
