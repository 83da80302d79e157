#!/usr/bin/env python3
"""SURVEY 8f-4 at BASELINE c2's size: 4096^2 x 256 replicas with a uniform field (Lattice.set_global_bias(0.25)) and with
open boundaries, or anisotropic couplings -- multi-class checkerboard kernels vs the plain periodic lattice and vs the general (CSR) path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import square  # noqa: E402

L, R, BETA = 4096, 256, 0.4407
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ea, eb, ej = square(L, L)
N = L * L
open_keep = ~((ea % L == L - 1) & (eb % L == 0)) & ~((ea // L == L - 1) & (eb // L == 0))
cases = [("periodic, h = 0", dict(), R, steps),
         ("uniform field h = 0.25", dict(biases=np.full(N, 0.25)), R, steps),
         ("open boundaries (x and y)", dict(keep=open_keep), R, steps),
         ("anisotropic |Jy| = 0.5 |Jx|", dict(jy=0.5), R, steps),
         ("open boundaries + field h = 0.25", dict(keep=open_keep, biases=np.full(N, 0.25)), R, steps),
         ("random field h_i = +-0.5", dict(biases=0.5 * np.random.default_rng(5).choice([-1.0, 1.0], N)), R, steps),
         ("uniform field h = 0.25, general path", dict(biases=np.full(N, 0.25), force_general=True), 16, 3)]
for name, kw, reps, T in cases:
    keep = kw.pop("keep", None)
    a, b, j = (ea, eb, ej) if keep is None else (ea[keep], eb[keep], ej[keep])
    jy = kw.pop("jy", None)
    if jy is not None:
        j = j.copy()
        j[1::2] *= jy                    # square(): right and down bonds alternate
    g = _capi.Graph(a, b, j, nvars=N, **kw)
    st = _capi.States(g, _capi.make_seeds(1, reps))
    st.do_time_steps(3, BETA)
    ms = min(st.do_time_steps_timed(T, BETA) for _ in range(2))
    print(f"{name:40s} kind={g.kind} fast_path={g.info.fast_path}  {reps * N * T / (ms * 1e-3):.3e} attempts/s  {ms / T * 1e3:8.1f} us/step "
          f"e/site={st.energies().mean() / N:.4f}", flush=True)
