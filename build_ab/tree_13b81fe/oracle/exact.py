"""Exact results that pin the oracle (TEST INFRASTRUCTURE ONLY; numpy, no reference code involved).

* enumerate_graph: brute-force Boltzmann averages on graphs of <= ~22 spins for the Hamiltonian of
  README.md:45-46, E = sum J s_a s_b - sum h s.
* kaufman_energy: B. Kaufman, Phys. Rev. 76, 1232 (1949): exact partition function of the
  ferromagnetic Ising model on a finite L x M torus, hence exact <E>(beta) for the periodic
  lattices of BASELINE.json's configs.
"""
import numpy as np


def enumerate_graph(ea, eb, ej, nvars, beta, biases=None):
    """Returns dict(E=<E>, E2=<E^2>, absM=<|M|>, M2=<M^2>, Z=Z, probs=Boltzmann weights, energies=E(state)).

    State index bit i = spin i (1 = True = +1).
    """
    assert nvars <= 24
    idx = np.arange(1 << nvars, dtype=np.uint32)
    spins = [(((idx >> i) & 1).astype(np.int8) * 2 - 1) for i in range(nvars)]
    energies = np.zeros(1 << nvars, dtype=np.float64)
    for a, b, j in zip(ea, eb, ej):
        energies += float(j) * (spins[int(a)] * spins[int(b)])
    if biases is not None:
        for i, h in enumerate(biases):
            energies -= float(h) * spins[i]
    mag = np.zeros(1 << nvars, dtype=np.int32)
    for s in spins:
        mag += s
    w = np.exp(-beta * (energies - energies.min()))
    Z = w.sum()
    p = w / Z
    return dict(E=float((p * energies).sum()), E2=float((p * energies ** 2).sum()),
                absM=float((p * np.abs(mag)).sum()), M2=float((p * mag.astype(np.float64) ** 2).sum()),
                Z=float(Z), probs=p, energies=energies, mag=mag)


def _log2cosh(x):
    x = np.abs(x)
    return x + np.log1p(np.exp(-2 * x))


def _log2sinh_abs(x):
    x = np.abs(x)
    with np.errstate(divide="ignore"):
        return x + np.log(-np.expm1(-2 * x))


def kaufman_lnZ(L, M, K):
    """ln Z of the ferromagnetic (J=1, E = -sum s s) Ising model on an L x M torus at K = beta J."""
    ld = np.longdouble
    K = ld(K)
    N = L * M
    l = np.arange(2 * L, dtype=np.longdouble)
    c = np.cosh(2 * K) / np.tanh(2 * K) - np.cos(l * np.pi / L)
    gamma = np.arccosh(np.maximum(c, ld(1)))
    gamma[0] = 2 * K + np.log(np.tanh(K))  # signed: negative above T_c
    odd, even = gamma[1::2], gamma[0::2]
    half = ld(M) / 2
    terms = [
        (_log2cosh(half * odd).sum(), 1.0),
        (_log2sinh_abs(half * odd).sum(), 1.0),
        (_log2cosh(half * even).sum(), 1.0),
        (_log2sinh_abs(half * even).sum(), -1.0 if even[0] < 0 else 1.0),
    ]
    mx = max(t[0] for t in terms)
    s = sum(sg * np.exp(t - mx) for t, sg in terms)
    return float(-np.log(ld(2)) + ld(N) / 2 * np.log(2 * np.sinh(2 * K)) + mx + np.log(s))


def kaufman_energy(L, M, beta, J=1.0):
    """Exact <E> (total, not per site) for E = -J sum_<ij> s_i s_j on the periodic L x M lattice.

    Richardson-extrapolated central difference of ln Z (long double); relative error ~1e-10.
    """
    K = beta * J

    def d(h):
        return (kaufman_lnZ(L, M, K + h) - kaufman_lnZ(L, M, K - h)) / (2 * h)

    h = 1e-4
    dlnZ_dK = (4 * d(h / 2) - d(h)) / 3
    return -J * dlnZ_dK


def square_lattice_edges(W, H, J=-1.0, rng=None):
    """Periodic W x H lattice, ids y*W+x, edges (i, right(i)), (i, down(i))  (SURVEY.md 8d).

    rng given: J_e = +-|J| i.i.d. with p = 1/2 (config c4's +-J spin glass).
    Returns (ea, eb, ej) arrays in the order right(0), down(0), right(1), down(1), ...
    """
    ids = np.arange(W * H, dtype=np.uint64).reshape(H, W)
    right = np.roll(ids, -1, axis=1)
    down = np.roll(ids, -1, axis=0)
    ea = np.stack([ids, ids], axis=-1).reshape(-1)
    eb = np.stack([right, down], axis=-1).reshape(-1)
    if rng is None:
        ej = np.full(ea.shape, float(J), dtype=np.float64)
    else:
        ej = abs(float(J)) * rng.choice(np.array([-1.0, 1.0]), size=ea.shape)
    return np.ascontiguousarray(ea), np.ascontiguousarray(eb), np.ascontiguousarray(ej)


def cubic_lattice_edges(L, J=-1.0):
    """Periodic L^3 lattice, ids (z*L+y)*L+x, edges to right/down/back."""
    ids = np.arange(L ** 3, dtype=np.uint64).reshape(L, L, L)
    nb = [np.roll(ids, -1, axis=2), np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)]
    ea = np.stack([ids] * 3, axis=-1).reshape(-1)
    eb = np.stack(nb, axis=-1).reshape(-1)
    ej = np.full(ea.shape, float(J), dtype=np.float64)
    return np.ascontiguousarray(ea), np.ascontiguousarray(eb), ej
