#!/usr/bin/env python3
"""the inputs that fell off the real-coupling packed path onto the f64 CSR kernels in round 3, measured on
both (attempts/s): (a) one pinning bias -- Lattice.set_individual_bias(7, 1e6), lattice.rs:104-126 -- on the 4096^2 ferromagnet
x 64; (b) graphs of degree 16 .. 31; (c) ONE experiment with real couplings (2048^2 Gaussian glass).
Usage: real_eligibility.py [steps]   -> one line per case (profiles/r04_real_eligibility.txt)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import square  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(2024)


def rate(ea, eb, ej, n, reps, T, beta, biases=None, csr=False):
    if csr:
        os.environ["ISINGMC_DISABLE_REAL"] = "1"
    try:
        g = _capi.Graph(ea, eb, ej, nvars=n, biases=biases, force_general=True)
        st = _capi.States(g, _capi.make_seeds(1, reps))
        st.do_time_steps(2, beta)
        ms = min(st.do_time_steps_timed(T, beta) for _ in range(2))
    finally:
        os.environ.pop("ISINGMC_DISABLE_REAL", None)
    return reps * n * T / (ms * 1e-3), g.info


def row(name, *args, csr_steps=None, **kw):
    T = args[5]
    real, info = rate(*args, **kw)
    a = list(args)
    a[5] = csr_steps or max(2, T // 5)
    csr, _ = rate(*a, csr=True, **kw)
    print(f"{name:58s} slots {info.real_slots:2d} heavy {info.real_heavy_sites:2d}  real path {real:.3e}  f64 CSR {csr:.3e}  ratio {real / csr:6.2f}", flush=True)


def random_graph(n, maxdeg, mean):
    stubs = np.repeat(np.arange(n, dtype=np.uint64), mean)
    rng.shuffle(stubs)
    a, b = stubs[0::2], stubs[1::2]
    keep = a != b
    a, b = a[keep], b[keep]
    hub = rng.integers(0, 4096, size=(4096 * (maxdeg - mean) // 2, 2)).astype(np.uint64)   # the first 4096 sites reach maxdeg
    hub = hub[hub[:, 0] != hub[:, 1]]
    a, b = np.concatenate([a, hub[:, 0]]), np.concatenate([b, hub[:, 1]])
    while True:
        deg = np.bincount(np.concatenate([a, b]).astype(np.int64), minlength=n)
        over = np.flatnonzero(deg > maxdeg)
        if len(over) == 0:
            return a, b, int(deg.max())
        bad = np.isin(a, over) | np.isin(b, over)
        drop = np.flatnonzero(bad)[: max(1, int((deg[over] - maxdeg).sum()))]
        a, b = np.delete(a, drop), np.delete(b, drop)


L4 = 4096
ea4, eb4, ej4 = square(L4, L4)
for hval in (3.0, 1e3, 1e6):
    h = np.zeros(L4 * L4)
    h[7] = hval
    row(f"4096^2 ferromagnet x64, set_individual_bias(7, {hval:g})", ea4, eb4, ej4, L4 * L4, 64, steps, 0.4407, biases=h)
for maxdeg, mean in ((20, 12), (31, 16)):
    n = 1 << 19
    a, b, dmax = random_graph(n, maxdeg, mean)
    row(f"random graph {n} sites, degree <= {dmax}, gaussian J x64", a, b, rng.normal(size=len(a)) * 0.3, n, 64, steps, 0.8)
L = 2048
ea, eb, _ = square(L, L)
ej = rng.normal(size=len(ea))
for reps in (1, 2, 4):
    row(f"2048^2 gaussian x{reps}", ea, eb, ej, L * L, reps, steps, 0.8, csr_steps=steps)
