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
t = time.perf_counter(); e3, s3 = lat.run_monte_carlo_annealing_and_get_energies([(0, 0.1), (1000, 1.0)], 1000, 4); t_ann = time.perf_counter() - t
t = time.perf_counter(); e4, s4 = lat.run_monte_carlo_annealing([(0, 0.1), (1000, 1.0)], 1000, 4); t_ann0 = time.perf_counter() - t
print(f"annealing(1000 steps): {t_ann0*1e3:.2f} ms; annealing_and_get_energies(1000 steps): {t_ann*1e3:.2f} ms")
L2 = 64
ids = np.arange(L2 * L2).reshape(L2, L2)
edges2 = [((int(a), int(b)), -1.0) for a, b in zip(ids.ravel(), np.roll(ids, -1, 1).ravel())] + \
         [((int(a), int(b)), -1.0) for a, b in zip(ids.ravel(), np.roll(ids, -1, 0).ravel())]
lat2 = m.Lattice(edges2, seed_gen=1); lat2.run_monte_carlo(0.3, 10, 8)
t = time.perf_counter(); lat2.run_monte_carlo_annealing_and_get_energies([(0, 0.1), (1000, 1.0)], 1000, 8); t2 = time.perf_counter() - t
t = time.perf_counter(); lat2.run_monte_carlo_sampling(0.4, 1000, 8, None, 100, 10); t3 = time.perf_counter() - t
print(f"64x64 x 8: annealing_and_get_energies(1000): {t2*1e3:.2f} ms; sampling(1000 steps, 100 samples): {t3*1e3:.2f} ms")
