#!/usr/bin/env python3
"""us/step of small problems (LDS-resident kernels): lattices that fit LDS and small general graphs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import square

for (W, H, R) in ((64, 64, 64), (256, 256, 64), (512, 512, 256), (16, 16, 4), (16, 16, 256), (32, 32, 64)):
    g = _capi.Graph(*square(W, H), nvars=W * H)
    st = _capi.States(g, _capi.make_seeds(1, R))
    st.do_time_steps(50, 0.4)
    ms = st.do_time_steps_timed(1000, 0.4)
    kind = "lattice" if g.kind == _capi.KIND_LATTICE2D else "general"
    print(f"{kind} {W}x{H} R={R}: {ms:.2f} us/step, {R * W * H * 1000 / (ms * 1e-3):.3e} attempts/s", flush=True)
