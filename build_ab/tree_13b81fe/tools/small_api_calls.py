#!/usr/bin/env python3
"""Per-call cost of the persistent containers on small problems: ClassicIsing.run_monte_carlo (classicising.rs:88-110),
get_energies / get_states, and ClassicalTempering.timesteps with exchange rounds."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import py_monte_carlo  # noqa: E402


def edges(W, H):
    ids = np.arange(W * H).reshape(H, W)
    return [((int(a), int(b)), -1.0) for a, b in zip(ids.ravel(), np.roll(ids, -1, 1).ravel())] + \
           [((int(a), int(b)), -1.0) for a, b in zip(ids.ravel(), np.roll(ids, -1, 0).ravel())]


def timeit(f, n=200):
    f()
    t = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t) / n * 1e6


for W, H, R in ((16, 16, 4), (64, 64, 8)):
    ci = py_monte_carlo.ClassicIsing(edges(W, H), None, R, 7)
    print(f"ClassicIsing {W}x{H} x {R}: run_monte_carlo(0.4, 1) {timeit(lambda: ci.run_monte_carlo(0.4, 1)):.1f} us   (0.4, 100) {timeit(lambda: ci.run_monte_carlo(0.4, 100), 50):.1f} us   "
          f"get_energies {timeit(ci.get_energies):.1f} us   get_states {timeit(ci.get_states):.1f} us", flush=True)
pt = py_monte_carlo.ClassicalTempering(edges(64, 64), seed=3)
for b in np.linspace(0.38, 0.5, 16):
    pt.add_graph(float(b))
pt.timesteps(100, 5)
print(f"ClassicalTempering 64x64 x 16 rungs: timesteps(10, 5) {timeit(lambda: pt.timesteps(10, 5), 100):.1f} us   timesteps(100, 5) {timeit(lambda: pt.timesteps(100, 5), 30):.1f} us   "
      f"timesteps(10) {timeit(lambda: pt.timesteps(10), 100):.1f} us", flush=True)
