#!/usr/bin/env python3
"""Energy FLUCTUATIONS against the exact specific heat (lives under tests/: Kaufman's partition function comes from
oracle/exact.py).  Var(E) = d^2 ln Z / d beta^2 exactly; a generator whose numbers were correlated between the spins of a quad,
between bit planes or between replicas could leave <E> intact and still distort the fluctuations.  Estimator: the variance ACROSS
the 256 independent replicas at one time (unbiased whatever the autocorrelation), averaged over the measured sweeps; its error
from 20 time blocks.

    python tests/highstat_var.py [out.txt]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import exact as X  # noqa: E402
from pyisingmontecarlo_amd import _capi  # noqa: E402


def exact_var(W, H, beta):
    """d^2 ln Z / d beta^2 by a Richardson-extrapolated second difference of Kaufman's ln Z (long double)."""
    def d2(h):
        return (X.kaufman_lnZ(W, H, beta + h) - 2.0 * X.kaufman_lnZ(W, H, beta) + X.kaufman_lnZ(W, H, beta - h)) / (h * h)
    h = 2e-3
    return (4.0 * d2(h / 2) - d2(h)) / 3.0


def case(name, W, H, beta, R, therm, steps):
    ea, eb, ej = X.square_lattice_edges(W, H, -1.0)
    g = _capi.Graph(ea, eb, ej)
    start = np.ones(W * H, dtype=np.uint8) if beta > 0.4407 else None
    st = _capi.States(g, _capi.make_seeds(int(beta * 1000) + W, R), initial_state=start)
    st.do_time_steps(therm, beta)
    t0 = time.time()
    var_t = []
    mean_t = []
    for _ in range(steps // 100):
        e = st.do_time_steps(100, beta, per_step_energies=True)        # [R, 100]
        var_t.append(e.var(axis=0, ddof=1))
        mean_t.append(e.mean(axis=0))
    var_t = np.concatenate(var_t)
    blocks = var_t[: len(var_t) // 20 * 20].reshape(20, -1).mean(axis=1)
    est, err = blocks.mean(), blocks.std(ddof=1) / np.sqrt(20)
    ref = exact_var(W, H, beta)
    z = (est - ref) / err
    e_mean = np.concatenate(mean_t).mean()
    print(f"{name:28s} {W}x{H} beta={beta:.2f} R={R}: Var(E) {est:14.2f} +- {err:10.2f} (rel {err / ref:.1e})  exact {ref:14.2f}  z = {z:+.2f}   "
          f"[<E> {e_mean:.1f} vs {X.kaufman_energy(W, H, beta):.1f}; specific heat per site {beta * beta * est / (W * H):.4f}; {time.time() - t0:.0f} s]", flush=True)
    return z


def main():
    zs = []
    for beta in (0.35, 0.42, 0.47, 0.55):
        zs.append(case("LDS-resident kernel", 64, 64, beta, 256, 20000, 400000))
    for beta in (0.35, 0.55):
        zs.append(case("persistent strips", 1024, 1024, beta, 64, 1500, 100000))
    for beta in (0.35, 0.40, 0.48, 0.55):
        zs.append(case("streaming kernels (c2 size)", 4096, 4096, beta, 256, 1500, 8000))
    zs = np.array(zs)
    print(f"{len(zs)} estimates: rms z = {np.sqrt((zs ** 2).mean()):.2f}, mean z = {zs.mean():+.2f}, max |z| = {np.abs(zs).max():.2f}")


if __name__ == "__main__":
    main()
