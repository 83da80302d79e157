#!/usr/bin/env python3
"""c3 (1024^2, 64 rungs, exchange round every 10 sweeps) with the persistent strip kernel vs the per-colour launches,
same box, interleaved.  python tools/c3_ab.py [steps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "400"
for rep in range(2):
    for mode in ("1", "0"):
        env = dict(os.environ, ISINGMC_STRIP=mode)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), "c3", "--steps", steps], env=env,
                             capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        rec = json.loads(line[0]) if line else {"error": out.stderr[-400:]}
        print(f"ISINGMC_STRIP={mode}", json.dumps(rec), flush=True)
