#!/usr/bin/env python3
"""Cost of an exchange round on the strip path: 1024^2 x 64 rungs, 400 timesteps, by rounds per 400 steps."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd.tempering import ClassicalTempering  # noqa: E402
from tools.bench_configs import square  # noqa: E402

L, G, T = 1024, 64, 400
for ik in ("1", "0"):
    os.environ["ISINGMC_PT_IN_KERNEL"] = ik
    pt = ClassicalTempering(square(L, L), seed=1)
    for b in 0.1 + 0.9 * np.arange(G) / 511 * 8:
        pt.add_graph(float(b))
    pt.timesteps(20)
    for freq in (400, 200, 100, 50, 20, 10, 5):
        pt.timesteps(T, replica_swap_freq=freq)
        t0 = time.perf_counter(); pt.timesteps(T, replica_swap_freq=freq); dt = time.perf_counter() - t0
        print(f"in_kernel={ik} swap_every={freq:4d}: {dt / T * 1e6:6.2f} us/step  ({T // freq} rounds: {(dt / T * 1e6 - 9.6) * freq:6.1f} us per round above 9.6 us/step)", flush=True)
