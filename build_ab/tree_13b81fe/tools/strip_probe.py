#!/usr/bin/env python3
"""Persistent strip kernel: device time per timestep of ONE launch (no exchange rounds), by replica count and lattice.
  python tools/strip_probe.py [L] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import square  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
g = _capi.Graph(*square(L, L), nvars=L * L)
for R in (16, 32, 64, 128):
    row = []
    for mode in ("1", "0"):
        os.environ["ISINGMC_STRIP"] = mode
        st = _capi.States(g, _capi.make_seeds(1, R))
        st.set_betas(np.linspace(0.1, 1.0, R))
        st.do_time_steps(50)
        ms = min(st.do_time_steps_timed(steps, 0.4) for _ in range(3))
        row.append(ms / steps * 1e3)
    print(f"L={L} R={R}: strip {row[0]:.2f} us/step ({R * L * L / row[0] / 1e6:.2f}e12/s)   streaming {row[1]:.2f} us/step "
          f"({R * L * L / row[1] / 1e6:.2f}e12/s)", flush=True)
