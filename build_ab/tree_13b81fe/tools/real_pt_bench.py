#!/usr/bin/env python3
"""Parallel tempering of Gaussian glasses on the real-coupling path (and of +-J glasses on the bit-sliced packed path): exchange rounds on the engine's stream (measurement,
decisions and relabelling as kernels) against the host swap step (energies read back, isingmc_host_pt_swap_round, betas
uploaded).  usage: real_pt_bench.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd.tempering import ClassicalTempering  # noqa: E402
from tools.bench_configs import cubic, square  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(5)
for name, (ea, eb, _), G in (("32^3", cubic(32), 64), ("64^3", cubic(64), 64), ("512^2", square(512, 512), 128), ("2048^2", square(2048, 2048), 64),
                             ("32^3 +-J", cubic(32), 64), ("64^3 +-J", cubic(64), 64)):
    ej = rng.choice([-1.0, 1.0], size=len(ea)) if name.endswith("+-J") else rng.normal(size=len(ea))  # +-J: the bit-sliced packed path
    n = int(max(ea.max(), eb.max())) + 1
    for host in ("0", "1"):
        os.environ["ISINGMC_PT_HOST"] = host
        pt = ClassicalTempering((ea, eb, ej), seed=1)
        for b in np.linspace(0.2, 1.6, G):
            pt.add_graph(float(b))
        pt.timesteps(20)
        pt.timesteps(40, replica_swap_freq=10)
        t0 = time.perf_counter()
        pt.timesteps(steps, replica_swap_freq=10)
        dt = time.perf_counter() - t0
        print(f"{name:9s} {G:4d} rungs  {'host swap step' if host == '1' else 'on the stream '}  {dt / steps * 1e6:9.2f} us/step  "
              f"{G * n * steps / dt:.3e} attempts/s  swaps {pt.get_total_swaps()}  on_stream={pt._on_stream}", flush=True)
