#!/usr/bin/env python3
"""c3 (1024^2, 64 rungs, exchange round every 10 sweeps): same box, interleaved A/B.
  python tools/c3_ab.py [steps] [lib.so ...]     default: the shipped library with ISINGMC_STRIP=1 and =0"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "400"
libs = sys.argv[2:]
arms = [({"ISINGMC_LIB_PATH": os.path.abspath(l)}, l) for l in libs] or [({"ISINGMC_STRIP": "1"}, "strip"), ({"ISINGMC_STRIP": "0"}, "streaming")]
for rep in range(2):
    for env_add, label in arms:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), "c3", "--steps", steps],
                             env=dict(os.environ, **env_add), capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        rec = json.loads(line[0]) if line else {"error": out.stderr[-400:]}
        print(label, json.dumps(rec), flush=True)
