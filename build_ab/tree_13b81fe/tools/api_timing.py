#!/usr/bin/env python3
"""End-to-end timing of the drop-in API at BASELINE c2's lattice (host <-> device included):
ingest, graph build, run_monte_carlo(beta_c, T, R) with its bool[R,N] output."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import py_monte_carlo as m

L, R, T = 4096, int(os.environ.get("R", 64)), int(os.environ.get("T", 100))
ids = np.arange(L * L, dtype=np.uint64).reshape(L, L)
ea = np.stack([ids, ids], -1).reshape(-1); eb = np.stack([np.roll(ids, -1, 1), np.roll(ids, -1, 0)], -1).reshape(-1)
ej = np.full(ea.shape, -1.0)
t = time.perf_counter(); lat = m.Lattice.from_arrays(ea, eb, ej, seed_gen=1); t_ingest = time.perf_counter() - t
t = time.perf_counter(); info = lat.engine_info(); t_graph = time.perf_counter() - t
t = time.perf_counter(); e, s = lat.run_monte_carlo(0.4407, T, R); t_run = time.perf_counter() - t
t = time.perf_counter(); e0, s0 = lat.run_monte_carlo(0.4407, 0, R); t_run0 = time.perf_counter() - t
print(json.dumps({"lattice": [L, L], "R": R, "T": T, "ingest_from_arrays_s": t_ingest, "graph_build_s": t_graph,
                  "run_monte_carlo_s": t_run, "of_which_init_plus_output_s": t_run0,
                  "attempts_per_s_api_inclusive": R * L * L * T / t_run, "output_GiB": s.nbytes / 2**30,
                  "energy_per_site": float(e.mean()) / L**2}))
