#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass (…_counter_collection.csv) per kernel name.
usage: tools/pmc_summary.py <counter_collection.csv> <out.csv> ["comment line" ...]
FETCH_SIZE (KB) is converted with the gfx950 correction of MI355X_MICROARCH.md (HBM section): hbm_read_bytes = FETCH_SIZE * 1024 * 2.
When SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE are both present, mfma_busy = MFMA_BUSY / ((GUI_ACTIVE / 8) * 1024): MFMA_BUSY counts cycles summed over the 1024
SIMDs, GRBM_GUI_ACTIVE is reported as the SUM over the 8 XCDs (round 1's summary forgot the / 8 and printed utilisations 8 x too small).
Other counters are reported as plain per-launch averages."""
import csv, sys
from collections import defaultdict

src, dst, comments = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(src)):
    a = acc[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
counters = sorted({c for k in acc.values() for c in k})
with open(dst, "w") as f:
    for c in comments:
        f.write("# " + c + "\n")
    mfma = "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters
    hdr = ["kernel", "calls"] + [f"avg_{c}" for c in counters] + (["avg_hbm_read_MB_corrected"] if "FETCH_SIZE" in counters else []) + (["mfma_busy_frac"] if mfma else [])
    f.write(",".join(hdr) + "\n")
    for k in sorted(acc, key=lambda k: -sum(v[1] for v in acc[k].values())):
        calls = max(v[0] for v in acc[k].values())
        row = ['"' + k.replace('"', "'") + '"', str(calls)] + [f"{acc[k][c][1] / max(1, acc[k][c][0]):.1f}" if c in acc[k] else "" for c in counters]
        if "FETCH_SIZE" in counters:
            row.append(f"{acc[k]['FETCH_SIZE'][1] / max(1, acc[k]['FETCH_SIZE'][0]) * 1024 * 2 / 1e6:.2f}" if "FETCH_SIZE" in acc[k] else "")
        if mfma:
            g = acc[k]["GRBM_GUI_ACTIVE"][1] / max(1, acc[k]["GRBM_GUI_ACTIVE"][0]); mb = acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1] / max(1, acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"][0])
            row.append(f"{mb / (g / 8.0 * 1024.0):.4f}" if g > 0 else "")
        f.write(",".join(row) + "\n")
print("wrote", dst)
