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
    for trial in range(20):
        n = int(rng.integers(5, 60))
        ea, eb = _random_graph(rng, n, min(2 * n, n * (n - 1) // 2 - 1), 15)
        scale = 10.0 ** rng.integers(-6, 7)
        ej = rng.normal(size=len(ea)) * scale
        h = None if trial % 3 == 0 else rng.normal(size=n) * scale * 0.5
        if trial % 5 == 0:  # a self-loop and a duplicated bond
            ea = np.concatenate([ea, [ea[0], 2]]).astype(np.uint64)
            eb = np.concatenate([eb, [eb[0], 2]]).astype(np.uint64)
            ej = np.concatenate([ej, [0.3 * scale, 1.5 * scale]])
        k, jq, hq, ok = capi.rj_quantise(ea, eb, ej, n, h)
        k2, jq2, hq2 = oracle.rj_quantise(ea, eb, ej, n, h)
        assert (k, ok) == (k2, oracle.rj_eligible(ea, eb, ej, n, h))
        assert np.array_equal(jq, jq2) and np.array_equal(hq, hq2)
        assert np.abs(jq).max() < 2 ** 30 + 8
        np.testing.assert_allclose(jq * 2.0 ** k, np.where(ea == eb, 0.0, ej), atol=2.0 ** (k - 1))
        for beta in (0.0, -1.0, 1e-12, 0.01, 0.4407, 1.0, 7.5, 1e9 / scale):
            assert capi.rj_beta(beta / scale, k) == oracle.rj_beta(beta / scale, k)


def test_eligibility_bounds(capi):
    ea, eb = np.array([0, 1, 2], dtype=np.uint64), np.array([1, 2, 3], dtype=np.uint64)
    assert capi.rj_quantise(ea, eb, np.array([1.0, -0.5, 0.25]), 4)[3]
    # one enormous bias: the common quantum would wipe out the other couplings -> not eligible (f64 CSR path)
    assert not capi.rj_quantise(ea, eb, np.array([1.0, -0.5, 0.25]), 4, np.array([1e9, 0, 0, 0]))[3]
    # degree 16 is one too many (four index nibbles hold 15 bonds + the own spin)
    hub_a = np.zeros(16, dtype=np.uint64)
    hub_b = np.arange(1, 17, dtype=np.uint64)
    assert not capi.rj_quantise(hub_a, hub_b, np.ones(16) * 0.7, 17)[3]
    assert capi.rj_quantise(hub_a[:15], hub_b[:15], np.ones(15) * 0.7, 16)[3]


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
        # energies are those of the couplings rounded to 2^k: within (terms) x 2^(k-1) of the f64 energy
        k = oracle.rj_quantise(ea, eb, ej, n, h)[0]
        for r in range(4):
            assert abs(e[r] - oracle.energy(ea, eb, ej, n, st[r], h)) <= (len(ea) + n) * 2.0 ** (k - 1)
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
