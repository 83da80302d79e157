#!/usr/bin/env python3
"""Wall time of BASELINE config c1 through the drop-in API (16x16, beta=0.3, 4 experiments, T=1000)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import py_monte_carlo as m
L = 16
ids = np.arange(L * L).reshape(L, L)
edges = [((int(a), int(b)), -1.0) for a, b in zip(ids.ravel(), np.roll(ids, -1, 1).ravel())] + \
        [((int(a), int(b)), -1.0) for a, b in zip(ids.ravel(), np.roll(ids, -1, 0).ravel())]
t = time.perf_counter(); lat = m.Lattice(edges, seed_gen=1234); t_new = time.perf_counter() - t
t = time.perf_counter(); e, s = lat.run_monte_carlo(0.3, 1000, 4); t_first = time.perf_counter() - t
ts = []
for _ in range(5):
    t = time.perf_counter(); e, s = lat.run_monte_carlo(0.3, 1000, 4); ts.append(time.perf_counter() - t)
t = time.perf_counter(); e2, s2 = lat.run_monte_carlo_sampling(0.3, 1000, 4, None, 100, 10); t_samp = time.perf_counter() - t
print(f"Lattice(): {t_new*1e3:.2f} ms; first run (graph build + HIP init): {t_first*1e3:.1f} ms; "
      f"run_monte_carlo(0.3, 1000, 4): {min(ts)*1e3:.2f} ms; sampling(1000 steps, 100 samples): {t_samp*1e3:.2f} ms; <E>/N = {e.mean()/256:.3f}")
