#!/usr/bin/env python3
"""Real-coupling packed path at VERDICT r02's size: 2048^2 Gaussian-J Edwards-Anderson glass x 128 replicas
(and a few neighbours: one biased site on the 4096^2 ferromagnet, the 3-d Gaussian glass 128^3, per-replica betas, the f64 CSR
path on the same inputs).  Usage: real_bench.py [steps] [exact case name]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import cubic, square  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
only = sys.argv[2] if len(sys.argv) > 2 else ""
rng = np.random.default_rng(2024)


def run(name, ea, eb, ej, n, reps, T, beta=0.8, biases=None, per_replica=False, env=None, per_step=False):
    if only and only != name:
        return
    for k, v in (env or {}).items():
        os.environ[k] = v
    g = _capi.Graph(ea, eb, ej, nvars=n, biases=biases)
    st = _capi.States(g, _capi.make_seeds(1, reps))
    if per_replica:
        st.set_betas(np.linspace(0.2, 1.6, reps))
    st.do_time_steps(3, None if per_replica else beta)
    ms = min(st.do_time_steps_timed(T, beta) for _ in range(2))
    rate = reps * n * T / (ms * 1e-3)
    rec = {"case": name, "kind": g.kind, "real_slots": g.info.real_slots, "replicas": reps, "sites": n, "steps": T,
           "attempts_per_s": rate, "us_per_step": ms / T * 1e3, "e_per_site": float(st.energies().mean() / n)}
    if per_step:  # energies after every timestep (lattice.rs:445-455): host clock around the blocking call
        import time
        st.do_time_steps(2, beta, per_step_energies=True)
        t0 = time.perf_counter()
        st.do_time_steps(T, beta, per_step_energies=True)
        rec["us_per_step_with_energies"] = (time.perf_counter() - t0) / T * 1e6
    print(json.dumps(rec), flush=True)
    for k in (env or {}):
        del os.environ[k]


L = 2048
ea, eb, _ = square(L, L)
ej = rng.normal(size=len(ea))
run("2048^2 gaussian x128", ea, eb, ej, L * L, 128, steps, per_step=True)
run("2048^2 gaussian x128 per-replica betas", ea, eb, ej, L * L, 128, steps, per_replica=True)
run("2048^2 gaussian x32", ea, eb, ej, L * L, 32, steps)
run("2048^2 gaussian x16 f64 CSR path", ea, eb, ej, L * L, 16, max(2, steps // 10), env={"ISINGMC_DISABLE_REAL": "1"})
h = rng.normal(size=L * L) * 0.5
run("2048^2 gaussian + gaussian fields x128", ea, eb, ej, L * L, 128, steps, biases=h)
L4 = 4096
ea4, eb4, ej4 = square(L4, L4)
h4 = np.zeros(L4 * L4)
h4[7] = -3.0
run("4096^2 ferromagnet, one biased site x64", ea4, eb4, ej4, L4 * L4, 64, max(2, steps // 2), beta=0.4407, biases=h4)
L3 = 128
ea3, eb3, _ = cubic(L3)
run("128^3 gaussian x128 (degree 6)", ea3, eb3, rng.normal(size=len(ea3)), L3 ** 3, 128, steps, beta=0.9)
