#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel (demangled name, VGPR/AGPR/SGPR, scratch, LDS, occupancy).
usage: tools/kres.py file.hip [filter-substring]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-DMINIGPT4_SHARED", "-DMINIGPT4_BUILD", "-I/root/repo/include", "-I/root/repo/minigpt4.cpp_amd/csrc",
       "--offload-arch=gfx950", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name": cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k.split()[0]] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines() if rows else []
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void mg4::", "")
    if flt and flt not in n: continue
    print(f"{n:48s} vgpr {r.get('VGPRs'):>4} agpr {r.get('AGPRs'):>3} sgpr {r.get('TotalSGPRs'):>3} scratch {r.get('ScratchSize'):>4} lds {r.get('LDS'):>6} occ {r.get('Occupancy')}")
