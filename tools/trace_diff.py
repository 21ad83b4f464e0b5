#!/usr/bin/env python3
"""First differing intermediate between a GPU parity-mode trace (MINIGPT4_PARITY_TRACE=<file>) and an oracle trace (refcpu.orc_set_trace) of the same evaluation sequence.
usage: tools/trace_diff.py gpu.trace oracle.trace     Records: char name[32]; int64 n; float32[n] (Engine::forward_ref / oracle/refcpu.c trace())."""
import struct
import sys

import numpy as np


def records(path):
    with open(path, "rb") as f:
        while True:
            h = f.read(40)
            if len(h) < 40:
                return
            name = h[:32].split(b"\0")[0].decode()
            n = struct.unpack("<q", h[32:])[0]
            yield name, np.frombuffer(f.read(4 * n), np.float32)


def diff(gpu_path, orc_path, verbose=True):
    """Returns (index, name, n_bad, max_abs) of the first record that differs, or None."""
    for i, ((ga, gv), (oa, ov)) in enumerate(zip(records(gpu_path), records(orc_path))):
        assert ga == oa and gv.size == ov.size, (i, ga, oa, gv.size, ov.size)
        same = np.array_equal(gv.view(np.uint32), ov.view(np.uint32)) or np.array_equal(gv, ov)
        if verbose:
            print(f"{i:4d} {ga:12s} n={gv.size:7d} {'==' if same else 'DIFF'}")
        if not same:
            bad = np.nonzero(gv != ov)[0]
            if verbose:
                print("   first bad elements:", bad[:8], "gpu", gv[bad[:8]], "oracle", ov[bad[:8]])
            return i, ga, int(bad.size), float(np.abs(gv - ov).max())
    return None


if __name__ == "__main__":
    r = diff(sys.argv[1], sys.argv[2])
    print("identical" if r is None else f"first difference: record {r[0]} ({r[1]}): {r[2]} elements, max |delta| {r[3]:.3e}")
