#!/usr/bin/env python3
"""Per-WAVE averages of SQ counters from a rocprofv3 --pmc pass (counter_collection.csv), per kernel symbol.
usage: tools/pmc_per_wave.py <counter_collection.csv> [substring of the kernel names to keep]
SQ_WAVES = waves launched; every other counter is divided by it.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_INST_CYCLES_* count quad-cycles (x 4 = cycles)."""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
keep = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if keep and keep not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = c.get("SQ_WAVES", 0)
    if not w: continue
    name = k.replace("void mg4::", "").split("(")[0]
    print(f"{name:40s} launches {len(calls[k]):4d} waves/launch {w / len(calls[k]):7.0f}  per wave: " + "  ".join(f"{n.replace('SQ_', '')}={v / w:.0f}" for n, v in sorted(c.items()) if n != "SQ_WAVES"))
