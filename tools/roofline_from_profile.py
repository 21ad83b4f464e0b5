#!/usr/bin/env python3
"""Recompute the bench line's roofline numbers from committed profiles (round-1 verdict, item 5: "make the bench line reproducible").

  tools/roofline_from_profile.py stats <kernel_stats.csv> <bench.json>
      per kernel symbol: average duration from the `rocprofv3 --kernel-trace --stats` summary  x  algorithmic bytes per call from the bench line's
      roofline.kernel_table (the weight planes / cached K, V rows a launch reads)  ->  GB/s and fraction of the 8 TB/s HBM peak,
      next to the bench's own hipEvent numbers; plus the whole-step check  sum(calls/token x avg)  vs  ms_per_step.
  tools/roofline_from_profile.py pmc <counter_collection.csv> <out.json> "<command that produced it>"
      HBM read bytes per launch and kernel symbol from a `rocprofv3 --pmc FETCH_SIZE` pass (FETCH_SIZE is in KB and counts 1/2 of streamed bytes on gfx950,
      MI355X_MICROARCH.md section HBM: bytes = FETCH_SIZE x 1024 x 2) -> profiles/pmc_traffic.json, the NAMED record bench.py's roofline.traffic is read from.
"""
import csv
import json
import re
import sys
from collections import defaultdict

HBM_PEAK_GBPS = 8000.0


def symbol(full_name: str) -> str:
    """'void mg4::k_matvec_v2<13, 3, 1, 1, 0>(mg4::MatSet, ...)' -> 'k_matvec_v2<13, 3, 1, 1, 0>' (the form Engine::profile_sites records)."""
    s = full_name.strip().strip('"')
    s = re.sub(r"^void\s+", "", s)
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):                      # cut the argument list: the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return s[:cut].replace("mg4::", "").strip()


def cmd_stats(stats_csv: str, bench_json: str) -> None:
    bench = json.loads([l for l in open(bench_json) if l.lstrip().startswith("{")][-1])
    table = {k["kernel"]: k for k in bench["roofline"]["kernel_table"]}
    stats = {}
    for r in csv.DictReader(open(stats_csv)):
        stats[symbol(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    print(f"{'kernel':58s} {'calls/tok':>9s} {'MB/call':>8s} {'rocprof us':>10s} {'GB/s':>8s} {'frac':>6s} | {'bench us':>8s} {'GB/s':>8s}")
    step_us = 0.0
    for name, k in sorted(table.items(), key=lambda kv: -kv[1]["us_per_token"]):
        parts = [p.strip() for p in name.split(" + ")]           # a site that launches two kernels back to back (argmax) is listed as "a + b"
        got = [stats[p] for p in parts if p in stats]
        if len(got) != len(parts):
            print(f"{name:58s} {k['calls_per_token']:9.2f} {k['bytes_per_call'] / 1e6:8.2f} {'(not in the CSV)':>10s}")
            continue
        us = sum(g[1] for g in got)
        gbps = k["bytes_per_call"] / (us * 1e-6) / 1e9 if us > 0 else 0.0
        step_us += k["calls_per_token"] * us
        print(f"{name:58s} {k['calls_per_token']:9.2f} {k['bytes_per_call'] / 1e6:8.2f} {us:10.2f} {gbps:8.0f} {gbps / HBM_PEAK_GBPS:6.3f} | {k['avg_us']:8.2f} {k['GBps']:8.0f}")
    dom = bench["roofline"]["kernel"]
    print(f"\ndominant kernel of the bench line: {dom}: frac {bench['roofline']['frac']:.4f} (bench, hipEvents)")
    print(f"sum over the table of calls/token x rocprof average = {step_us / 1e3:.3f} ms per token;  bench ms_per_step = {bench['ms_per_step']:.3f}")
    ws = bench["roofline"].get("whole_step")
    if ws:
        print(f"whole step: {ws['bytes'] / 1e9:.3f} GB / {ws['ms']:.3f} ms = {ws['GBps']:.0f} GB/s = {ws['frac']:.3f} of {HBM_PEAK_GBPS:.0f}")


def cmd_pmc(counter_csv: str, out_json: str, command: str) -> None:
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(counter_csv)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        a = acc[symbol(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    kernels = {k: {"launches": n, "bytes_per_launch": v / n * 1024.0 * 2.0} for k, (n, v) in acc.items() if n}
    json.dump({"source_csv": counter_csv, "command": command, "correction": "bytes = FETCH_SIZE[KB] x 1024 x 2 (gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes)",
               "kernels": kernels}, open(out_json, "w"), indent=1, sort_keys=True)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_per_launch"])[:12]:
        print(f"{k:60s} {v['launches']:6d} launches  {v['bytes_per_launch'] / 1e6:9.2f} MB/launch")
    print("wrote", out_json)


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "stats":
        cmd_stats(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 5 and sys.argv[1] == "pmc":
        cmd_pmc(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
