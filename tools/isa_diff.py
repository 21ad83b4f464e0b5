#!/usr/bin/env python3
"""Do two revisions of a .hip file compile to the same device code?  usage: tools/isa_diff.py <git-rev> [file.hip ...]  [-- extra hipcc flags]
Compiles every kernel file at <git-rev> and in the working tree with `hipcc -S --cuda-device-only` (gfx950) and compares the kernels function by function with block
labels normalised.  Used after macro-gated experiments were added to show that the default build's kernels are still the ones that were validated on the GPU."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "minigpt4.cpp_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-DMINIGPT4_SHARED", "-DMINIGPT4_BUILD", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only"]


def funcs(path):
    out, cur, buf = {}, None, []
    for line in open(path):
        m = re.match(r"^(_ZN3mg4\w+):", line)
        if m:
            cur, buf = m.group(1), []
            continue
        if cur is not None:
            buf.append(line)
            if "s_endpgm" in line:
                out[cur] = [re.sub(r"\.LBB\d+_", ".LBB_", x) for x in buf if not x.strip().startswith((".", "; %", ".L"))]
                cur = None
    return out


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    rev = args[0]
    files = args[1:] or [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    old = tempfile.mkdtemp(prefix="isa_old_")
    for f in os.listdir(CSRC):
        if f.endswith((".hpp", ".hip")):
            r = subprocess.run(["git", "show", f"{rev}:minigpt4.cpp_amd/csrc/{f}"], cwd=ROOT, capture_output=True)
            if r.returncode == 0:
                open(os.path.join(old, f), "wb").write(r.stdout)
    bad = 0
    for f in files:
        a_s, b_s = os.path.join(old, f + ".old.s"), os.path.join(old, f + ".new.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(ROOT, "include"), "-I" + old, "-o", a_s, os.path.join(old, f)], stderr=subprocess.DEVNULL)
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", b_s, os.path.join(CSRC, f)], stderr=subprocess.DEVNULL)
        a, b = funcs(a_s), funcs(b_s)
        diff = [k for k in a if k in b and a[k] != b[k]]
        gone = [k for k in a if k not in b]
        print(f"{f}: {len(a)} kernels at {rev}, {len(b)} now, {len(diff)} changed, {len(gone)} removed, {len([k for k in b if k not in a])} new")
        for k in diff + gone:
            print("   ", k[:140])
        bad += len(diff) + len(gone)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
