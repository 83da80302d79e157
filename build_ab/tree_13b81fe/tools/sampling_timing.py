#!/usr/bin/env python3
"""run_monte_carlo_sampling at scale -- the whole call against its parts (1024^2 x 64 replicas, S = 100
samples, one every 10 timesteps): sweeps only, and the floor set by expanding bits to the reference's bool[R,S,N]."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import square  # noqa: E402

L, R, S, FREQ = 1024, 64, 100, 10
g = _capi.Graph(*square(L, L), nvars=L * L)
st = _capi.States(g, _capi.make_seeds(1, R))
st.do_time_steps(50, 0.44)
t0 = time.perf_counter(); st.do_time_steps(S * FREQ, 0.44); t_sweeps = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(S):
    st.do_time_steps(FREQ, 0.44)
t_blocks = time.perf_counter() - t0
for rep in range(2):
    t0 = time.perf_counter(); e, s = st.run_sampling(0.44, 0, FREQ, S); t_call = time.perf_counter() - t0
    print(f"run_sampling call {rep}: {t_call * 1e3:.1f} ms  ({s.nbytes / t_call / 1e9:.1f} GB/s of bools)")
out = np.empty((R, L * L), dtype=np.bool_)
t0 = time.perf_counter()
for _ in range(10):
    st.states(out=out)
t_get = (time.perf_counter() - t0) / 10
print(f"sweeps only ({S * FREQ} steps, one call): {t_sweeps * 1e3:.1f} ms; as {S} calls of {FREQ}: {t_blocks * 1e3:.1f} ms")
print(f"states() of one sample ({out.nbytes / 2**20:.0f} MiB of bools): {t_get * 1e3:.2f} ms -> {S} samples: {S * t_get * 1e3:.0f} ms "
      f"({out.nbytes / t_get / 1e9:.1f} GB/s)")
