#!/usr/bin/env python3
"""Which loops of which kernels drain their vector-memory pipeline?  usage: tools/isa_waits.py file.s [name-filter]
For every kernel in a `hipcc -S --cuda-device-only` listing: every backward branch = a loop; per loop the vector-memory loads issued inside it and the
s_waitcnt vmcnt(N) values found inside it.  A loop that issues loads and also waits for vmcnt(0) has no load in flight across that wait: a software pipeline
written in the source did not survive instruction scheduling (round 3: k_gemm_f16's post-load zeroing selects and its reordered prologue)."""
import re
import sys


def kernels(path):
    cur, buf = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, buf = m.group(1), []
            continue
        if cur is not None:
            buf.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                yield cur, buf
                cur = None


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, body in kernels(path):
        if filt and filt not in name:
            continue
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        rows = []
        for a, b in loops:
            seg = body[a:b]
            loads = sum(1 for l in seg if re.search(r"\b(global_load|buffer_load|flat_load)", l))
            mf = sum(1 for l in seg if "v_mfma" in l)
            waits = [int(m.group(1)) for l in seg for m in [re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)] if m]
            if loads:
                rows.append((a, b - a, loads, mf, sorted(set(waits))))
        if rows:
            print(name[:110])
            for a, n, loads, mf, waits in rows:
                flag = "  <-- DRAINS" if 0 in waits else ""
                print(f"    loop @{a:5d} len {n:5d}  loads {loads:3d}  mfma {mf:3d}  vmcnt waits {waits}{flag}")


if __name__ == "__main__":
    main()
