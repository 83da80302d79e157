#!/usr/bin/env python3
"""Exercises every device path once at medium size (for `rocprofv3 --kernel-trace --stats`: which kernels take how long)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import square, cubic

rng = np.random.default_rng(0)
# CSR path: random graph with real couplings and fields, 200k sites, 8 replicas
n = 200_000
ea = rng.integers(0, n, 3 * n).astype(np.uint64); eb = rng.integers(0, n, 3 * n).astype(np.uint64); ej = rng.normal(size=3 * n)
g = _capi.Graph(ea, eb, ej, nvars=n, biases=rng.normal(size=n))
st = _capi.States(g, _capi.make_seeds(1, 8))
st.do_time_steps(20, 0.5); st.energies(); st.states()
st.do_time_steps(10, 0.5, per_step_energies=True)
# packed path: 96^3 cubic, 64 replicas
g2 = _capi.Graph(*cubic(96), nvars=96 ** 3, force_general=True)
s2 = _capi.States(g2, _capi.make_seeds(2, 64))
s2.do_time_steps(20, 0.22); s2.energies(); s2.states()
s2.do_time_steps(10, 0.22, per_step_energies=True)
s2.run_sampling(0.22, 5, 2, 4)
# lattice path: 2048^2 +-J, 16 replicas, sampling + annealing energies
g3 = _capi.Graph(*square(2048, 2048, np.random.default_rng(3)), nvars=2048 * 2048)
s3 = _capi.States(g3, _capi.make_seeds(3, 16))
s3.do_time_steps(20, 0.5); s3.energies(); s3.states()
s3.do_time_steps(10, np.linspace(0.1, 1.0, 10), per_step_energies=True)
s3.run_sampling(0.5, 4, 2, 3)
print("census done")
