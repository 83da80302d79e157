#!/usr/bin/env python3
"""End-to-end timing of the drop-in API on the packed paths (host <-> device included): 2048^2 Gaussian glass x 128 experiments
(real-coupling path) and 128^3 +-J glass x 64 (bit-sliced packed path): graph build, run_monte_carlo with its bool[R,N] output."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import py_monte_carlo as m
from tools.bench_configs import cubic, square

rng = np.random.default_rng(1)
for name, (ea, eb, _), pm, R, T in (("2048^2 gaussian", square(2048, 2048), False, 128, 100), ("128^3 +-J", cubic(128), True, 64, 100)):
    ej = rng.choice([-1.0, 1.0], size=len(ea)) if pm else rng.normal(size=len(ea))
    N = int(max(ea.max(), eb.max())) + 1
    t = time.perf_counter(); lat = m.Lattice.from_arrays(ea, eb, ej, seed_gen=1); t_ingest = time.perf_counter() - t
    t = time.perf_counter(); info = lat.engine_info(); t_graph = time.perf_counter() - t
    t = time.perf_counter(); e, s = lat.run_monte_carlo(0.8, T, R); t_run = time.perf_counter() - t
    t = time.perf_counter(); e0, s0 = lat.run_monte_carlo(0.8, 0, R); t_run0 = time.perf_counter() - t
    print(json.dumps({"case": name, "R": R, "T": T, "ingest_from_arrays_s": t_ingest, "graph_build_s": t_graph, "run_monte_carlo_s": t_run,
                      "of_which_init_plus_output_s": t_run0, "attempts_per_s_api_inclusive": R * N * T / t_run, "output_GiB": s.nbytes / 2**30,
                      "energy_per_site": float(e.mean()) / N}), flush=True)
