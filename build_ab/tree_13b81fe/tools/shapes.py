#!/usr/bin/env python3
"""Sweep rate of the lattice path by lattice shape (power-of-two rows take the division-free mapping and the looping kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import square
for (W, H, R) in ((4096, 4096, 64), (3072, 3072, 112), (6144, 2048, 84), (4096, 3000, 88), (1280, 1280, 640), (4032, 4032, 66)):
    g = _capi.Graph(*square(W, H), nvars=W * H)
    assert g.kind == _capi.KIND_LATTICE2D
    st = _capi.States(g, _capi.make_seeds(1, R)); st.do_time_steps(5, 0.4407)
    ms = st.do_time_steps_timed(40, 0.4407)
    print(f"{W}x{H} R={R}: {R * W * H * 40 / (ms * 1e-3):.3e} attempts/s", flush=True)
