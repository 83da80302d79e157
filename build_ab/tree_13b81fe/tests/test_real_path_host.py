"""Real-coupling packed path, the parts that need no GPU: the host halves of the spec through the C ABI
against the oracle (quantisation, acceptance scales, log table), the accuracy of the integer acceptance test, and oracle
engine E against exact enumeration (K2).  Reference surface served by this path: any f64 coupling (lattice.rs:46-50) and any
site bias (lattice.rs:104-131, 186-189)."""
import hashlib
import json
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "real_path.json")


def _random_graph(rng, n, m, maxdeg):
    pairs, deg = set(), np.zeros(n, dtype=int)
    while len(pairs) < m:
        a, b = (int(v) for v in rng.integers(0, n, 2))
        if a != b and deg[a] < maxdeg and deg[b] < maxdeg and (min(a, b), max(a, b)) not in pairs:
            pairs.add((min(a, b), max(a, b)))
            deg[a] += 1
            deg[b] += 1
    pairs = sorted(pairs)
    rng.shuffle(pairs)
    ea = np.array([p[0] for p in pairs], dtype=np.uint64)
    eb = np.array([p[1] for p in pairs], dtype=np.uint64)
    return ea, eb


def test_log_table_matches_oracle_and_golden(capi, oracle):
    lt = capi.rj_log_table()
    assert np.array_equal(lt, oracle.rj_log_table())
    assert lt[0] == 0 and lt[2048] in (1 << 24, (1 << 24) + 1) and np.all(np.diff(lt.astype(np.int64)) > 0)
    golden = json.load(open(GOLDEN))
    assert hashlib.sha256(lt.tobytes()).hexdigest() == golden["log_table_sha256"]


def test_lambda_is_minus_log2_to_2e_minus_7(oracle):
    rng = np.random.default_rng(1)
    us = np.concatenate([rng.integers(1, 2 ** 32, 20000, dtype=np.uint64), np.arange(1, 2000, dtype=np.uint64),
                         2 ** 32 - np.arange(1, 2000, dtype=np.uint64)])
    err = np.array([oracle.rj_lambda(int(u)) / 2 ** 24 - (32 - math.log2(float(u))) for u in us])
    assert np.abs(err).max() < 2e-7 and abs(err.mean()) < 1e-8
    assert oracle.rj_lambda(0) == 159 << 24 and oracle.rj_lambda(1) == 32 << 24 and oracle.rj_lambda(2 ** 32 - 1) == 0
    golden = json.load(open(GOLDEN))
    for u, lam in golden["lambda"]:
        assert oracle.rj_lambda(u) == lam


def test_quantisation_and_beta_scales_match_oracle(capi, oracle):
    rng = np.random.default_rng(3)
    for trial in range(30):
        n = int(rng.integers(5, 60))
        ea, eb = _random_graph(rng, n, min(2 * n, n * (n - 1) // 2 - 1), 15)
        scale = 10.0 ** rng.integers(-6, 7)
        ej = rng.normal(size=len(ea)) * scale
        h = None if trial % 3 == 0 else rng.normal(size=n) * scale * 0.5
        if trial % 5 == 0:  # a self-loop and a duplicated bond
            ea = np.concatenate([ea, [ea[0], 2]]).astype(np.uint64)
            eb = np.concatenate([eb, [eb[0], 2]]).astype(np.uint64)
            ej = np.concatenate([ej, [0.3 * scale, 1.5 * scale]])
        if trial % 4 == 1 and h is not None:  # heavy sites: a pinning bias, and a site whose large terms could cancel
            h[3] = 1e4 * scale
            if trial % 8 == 1:
                h[1] = -2e3 * scale
        if trial % 7 == 2:
            ej[0] = 5e5 * scale  # one enormous bond: both its ends are heavy and dominated by it
        k, jq, hq, d, ok = capi.rj_quantise(ea, eb, ej, n, h)
        k2, jq2, hq2, d2 = oracle.rj_quantise(ea, eb, ej, n, h)
        assert (k, ok) == (k2, oracle.rj_eligible(ea, eb, ej, n, h))
        assert np.array_equal(jq, jq2) and np.array_equal(hq, hq2) and np.array_equal(d, d2)
        assert np.abs(jq).max() < 2 ** 30 + 8 and np.abs(hq).max() < 2 ** 30 + 8
        if trial % 4 == 1 and h is not None or trial % 7 == 2:
            assert d.max() > 0
        # every site sees its bonds and its bias to half a quantum of its own scale 2^(k + d_i)
        ia, ib = ea.astype(np.int64), eb.astype(np.int64)
        bond = np.where(ea == eb, 0.0, ej)
        np.testing.assert_array_less(np.abs(jq[:, 0] * 2.0 ** (k + d[ia].astype(int)) - bond), 2.0 ** (k + d[ia].astype(float) - 1) * (1 + 1e-12) + 1e-300)
        np.testing.assert_array_less(np.abs(jq[:, 1] * 2.0 ** (k + d[ib].astype(int)) - bond), 2.0 ** (k + d[ib].astype(float) - 1) * (1 + 1e-12) + 1e-300)
        # the two energy levels reproduce every term to Fmax 2^-54
        kE, jhi, jlo, hhi, hlo = capi.rj_energy_levels(ea, eb, ej, n, h)
        lev = oracle.rj_energy_levels(ea, eb, ej, n, h)
        assert kE == lev[0] and all(np.array_equal(x, y) for x, y in zip((jhi, jlo, hhi, hlo), lev[1:]))
        assert np.abs(jlo).max() <= 2 ** 23 and np.abs(jhi).max() < 2 ** 29 + 8
        np.testing.assert_array_less(np.abs(jhi * 2.0 ** kE + jlo * 2.0 ** (kE - 24) - bond), 2.0 ** (kE - 25) * (1 + 1e-9) + 1e-300)
        if h is not None:
            np.testing.assert_array_less(np.abs(hhi * 2.0 ** kE + hlo * 2.0 ** (kE - 24) - h), 2.0 ** (kE - 25) * (1 + 1e-9) + 1e-300)
        for beta in (0.0, -1.0, 1e-12, 0.01, 0.4407, 1.0, 7.5, 1e9 / scale):
            assert capi.rj_beta(beta / scale, k) == oracle.rj_beta(beta / scale, k)


def test_eligibility_bounds(capi, oracle):
    ea, eb = np.array([0, 1, 2], dtype=np.uint64), np.array([1, 2, 3], dtype=np.uint64)
    j = np.array([1.0, -0.5, 0.25])
    assert capi.rj_quantise(ea, eb, j, 4)[4]
    # one enormous bias: the site quantises at its own coarse scale (it is dominated by that bias: no decision can hinge on the
    # bits that scale drops), everything else keeps the graph's quantum (round 3: such graphs fell to the f64 CSR path)
    k, jq, hq, d, ok = capi.rj_quantise(ea, eb, j, 4, np.array([1e9, 0, 0, 0]))
    assert ok and d[0] > 0 and not d[1:].any() and k == capi.rj_quantise(ea, eb, j, 4)[0] + 5   # the quantum follows 64 x the median term
    assert jq[0, 1] * 2.0 ** k == 1.0 and abs(hq[0] * 2.0 ** (k + int(d[0])) - 1e9) <= 2.0 ** (k + int(d[0]) - 1)
    # a heavy site whose two large couplings can cancel each other is not dominated by one term: f64 CSR path
    ca, cb = np.arange(9, dtype=np.uint64), np.arange(1, 10, dtype=np.uint64)       # a chain of ordinary bonds ...
    big = np.ones(9)
    big[3], big[4] = 1e6, -1e6 + 1                                                    # ... and site 4 between two enormous ones
    assert not capi.rj_quantise(ca, cb, big, 10)[4] and not oracle.rj_eligible(ca, cb, big, 10)
    big[4] = 1.0                                                                      # one enormous bond alone dominates both its ends
    assert capi.rj_quantise(ca, cb, big, 10)[4] and oracle.rj_eligible(ca, cb, big, 10)
    # degree 32 is one too many (eight index nibbles hold 31 bonds + the own spin)
    hub_a = np.zeros(32, dtype=np.uint64)
    hub_b = np.arange(1, 33, dtype=np.uint64)
    assert not capi.rj_quantise(hub_a, hub_b, np.ones(32) * 0.7, 33)[4]
    assert capi.rj_quantise(hub_a[:31], hub_b[:31], np.ones(31) * 0.7, 32)[4]


def test_heavy_site_decisions_match_f64(oracle):
    """A site pinned by a bias 10^6 x the couplings: at every beta from 1e-9 to 1e3 the integer test with the site's shift takes
    the decision exp(-beta dE) dictates, up to the 2^-23 resolution of beta dE and the 2^-32 grain of the uniform."""
    ea, eb = np.array([0, 1, 2], dtype=np.uint64), np.array([1, 2, 3], dtype=np.uint64)
    j, h = np.array([1.0, -0.5, 0.25]), np.array([1e6, 0.0, 0.0, 0.0])
    k, jq, hq, d, _ = oracle.rj_quantise(ea, eb, j, 4, h) + (None,)
    X = int(hq[0]) - int(jq[0, 0])          # site 0 up, its neighbour up: dE = 2 (h - J) in units of 2^(k + d_0)
    dE = 2.0 * X * 2.0 ** (k + int(d[0]))
    rng = np.random.default_rng(4)
    for beta in (1e-9, 1e-7, 3e-6, 1e-4, 0.01, 1.0, 1e3):
        sh, mant = oracle.rj_beta(beta, k)
        p = math.exp(-beta * dE)
        us = np.concatenate([rng.integers(0, 2 ** 32, 2000, dtype=np.uint64), [0, 1, 2 ** 32 - 1]])
        for u in us:
            acc = oracle.rj_accept(X, int(u), sh, mant, int(d[0]))
            uf = (int(u) + 0.5) / 2 ** 32
            if uf < p * (1 - 1e-5) - 2.0 ** -31:
                assert acc, (beta, u)
            if uf > p * (1 + 1e-5) + 2.0 ** -31 and u != 0:
                assert not acc, (beta, u)
        assert oracle.rj_accept(-X, 12345, sh, mant, int(d[0]))   # towards the field: always


def test_acceptance_probability_is_exp_to_1e_minus_7(oracle):
    """P(accept | X) = (number of accepting u) / 2^32 by bisection (the test is monotone in u) against exp(-beta dE)."""
    k = -28
    for beta in (0.05, 0.4407, 1.0, 3.0):
        sh, mant = oracle.rj_beta(beta, k)
        for dE in (0.01, 0.5, 2.0, 6.0):
            X = int(round(dE / 2 / 2.0 ** k))
            lo, hi = 0, 2 ** 32 - 1
            assert oracle.rj_accept(X, 0, sh, mant)
            while lo < hi:
                mid = (lo + hi + 1) // 2
                if oracle.rj_accept(X, mid, sh, mant):
                    lo = mid
                else:
                    hi = mid - 1
            p, exact = (lo + 1) / 2 ** 32, math.exp(-beta * dE)
            assert abs(p - exact) <= 2.5e-7 * exact + 2.0 ** -31, (beta, dE, p, exact)
    # downhill and flat moves are always accepted, beta <= 0 accepts everything
    sh, mant = oracle.rj_beta(0.7, k)
    for u in (0, 1, 12345, 2 ** 32 - 1):
        assert oracle.rj_accept(0, u, sh, mant) and oracle.rj_accept(-5, u, sh, mant) and oracle.rj_accept(-2 ** 30, u, sh, mant)
        assert oracle.rj_accept(2 ** 30, u, *oracle.rj_beta(0.0, k))


def test_engine_e_against_exact_enumeration_k2(oracle, exact):
    """K2 for the new engine: a 14-spin random graph with Gaussian couplings and biases, and a 4 x 4 torus with one biased
    site (set_individual_bias on an otherwise uniform lattice): <E>, <|M|> within 4 sigma of the exact Boltzmann averages."""
    rng = np.random.default_rng(5)
    n = 14
    ea, eb = _random_graph(rng, n, 24, 6)
    ej, h = rng.normal(size=len(ea)), rng.normal(size=n) * 0.5
    cases = [(ea, eb, ej, n, h, 0.6)]
    ea2, eb2, ej2 = exact.square_lattice_edges(4, 4, -1.0)
    h2 = np.zeros(16)
    h2[7] = -3.0
    cases.append((ea2, eb2, ej2, 16, h2, 0.35))
    ka, kb = np.triu_indices(14, 1)                                   # complete graph on 14 spins: degree 13, the four-nibble shape
    cases.append((ka.astype(np.uint64), kb.astype(np.uint64), rng.normal(size=len(ka)) / 3.5, 14, None, 0.9))
    for ea, eb, ej, n, h, beta in cases:
        assert oracle.rj_eligible(ea, eb, ej, n, h)
        ex = exact.enumerate_graph(ea, eb, ej, n, beta, h)
        seeds = oracle.make_seeds(11, 64)
        T, burn = 3000, 200
        states = None
        e_acc, m_acc = np.zeros(64), np.zeros(64)
        e, st, eps = oracle.rj_run(ea, eb, ej, n, seeds, T, betas=[beta] * T, biases=h, per_step=True)
        e_acc = eps[:, burn:].mean(axis=1)
        # energies are those of the ORIGINAL couplings (two integer levels): within (terms) x 2^(kE-25) + rounding of the f64 energy
        kE = oracle.rj_energy_levels(ea, eb, ej, n, h)[0]
        for r in range(4):
            ref = oracle.energy(ea, eb, ej, n, st[r], h)
            assert abs(e[r] - ref) <= (len(ea) + n) * 2.0 ** (kE - 25) + 8 * np.finfo(float).eps * (np.abs(ej).sum() + (0 if h is None else np.abs(h).sum()))
        # magnetisation: re-run in blocks to sample |M| (states only come out at the end of a call)
        m_samples = []
        t0, states = T, st  # (rj_run continues IN PLACE in the array it is handed)
        for blk in range(150):
            _, states = oracle.rj_run(ea, eb, ej, n, seeds, 5, betas=[beta] * 5, biases=h, states=states, t0=t0)
            t0 += 5
            m_samples.append(np.abs(2 * states[:64].sum(axis=1).astype(np.int64) - n))
        m_acc = np.mean(m_samples, axis=0)
        for got, want in ((e_acc, ex["E"]), (m_acc, ex["absM"])):
            z = (got.mean() - want) / (got.std(ddof=1) / np.sqrt(len(got)))
            assert abs(z) < 4.0, (z, got.mean(), want)


def test_engine_e_regression_pin(oracle, exact):
    golden = json.load(open(GOLDEN))["glass_12x10"]
    ea, eb, _ = exact.square_lattice_edges(12, 10, 1.0)
    grng = np.random.default_rng(7)
    ej, h = grng.normal(size=len(ea)), grng.normal(size=120) * 0.3
    e, st = oracle.rj_run(ea, eb, ej, 120, oracle.make_seeds(5, 40), 6, betas=[0.8] * 6, biases=h)
    assert hashlib.sha256(st[:40].tobytes()).hexdigest() == golden["sha256"] and float(e[0]) == golden["energy0"]
