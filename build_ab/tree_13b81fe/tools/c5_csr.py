#!/usr/bin/env python3
"""BASELINE c5 (256^3 cubic, 64 replicas) through the thread-per-site CSR kernel (ISINGMC_DISABLE_PACKED=1 forces it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import cubic
L, R = 256, 64
g = _capi.Graph(*cubic(L), nvars=L ** 3, force_general=True)
st = _capi.States(g, _capi.make_seeds(1, R))
st.do_time_steps(2, 0.2217)
ms = st.do_time_steps_timed(10, 0.2217)
print(f"c5 via the CSR kernel: {R * L**3 * 10 / (ms * 1e-3):.3e} attempts/s, {ms / 10:.2f} ms per sweep")
