#!/usr/bin/env python3
"""A/B runner for the image-encode path: tools/ab_encode.py NAME[:ENV=VAL,...] ...  (13B-shaped synthetic vision file, device ms per encode)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    name, _, envs = spec.partition(":")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v.replace(";", ",")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_encode.py"), "8"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = [l for l in p.stdout.splitlines() if l.startswith("encode ms")]
    print(f"{name:24s} {line[-1] if line else 'FAILED ' + p.stderr[-300:]}", flush=True)
