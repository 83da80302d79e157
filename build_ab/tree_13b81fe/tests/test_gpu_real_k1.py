"""Real-coupling path, round 4:

 * K1 as a TESTED tolerance: the energies the HIP path returns against the UNROUNDED f64 Hamiltonian recomputed on the host in
   extended precision from the original `ej` / biases (lattice.rs:208, 284, 454 return the f64 energy of the configuration) --
   2048^2 Gaussian x 128 (the bench case; final energies and the per-step column), 128^3 Gaussian, a random degree-15 graph with
   Gaussian fields.  Asserted: the analytic bound terms x Fmax 2^-54 (+ a few ulp of the sum of |terms| for the two f64
   roundings of the result) AND the relative figure 1e-13 of sum |terms| -- four decades inside BASELINE.md's 1e-9.
 * heavy sites (one pinning bias, lattice.rs:104-127 set_individual_bias(var, 1e6)): bit-exact against engine E, K2 against
   exact enumeration, and the same configurations as the f64 CSR path would pin.
 * degrees 16 .. 31 (23 / 31 ELL slots) against engine E.
 * ISINGMC_FLAG_STABLE_PATH: experiment k does not depend on the number of experiments (lattice.rs:83-91, 198)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _f64_energy(ea, eb, ej, spins, h=None):
    """sum J s s - sum h s accumulated in extended precision (x87 long double: 64-bit mantissa) -> the correctly rounded f64
    energy up to ~1e-19 x terms."""
    s = spins.astype(np.int8) * 2 - 1
    prod = (s[ea.astype(np.int64)] * s[eb.astype(np.int64)]).astype(np.float64) * ej
    prod = np.where(ea == eb, ej, prod)                                   # self-loops: s s = 1
    e = np.sum(prod, dtype=np.longdouble)
    if h is not None:
        e -= np.sum(h * s, dtype=np.longdouble)
    return e


@pytest.mark.parametrize("case", ["gauss2d_2048x128", "gauss3d_128", "deg15_fields"])
def test_k1_energies_against_the_unrounded_f64_hamiltonian(capi, exact, case, monkeypatch):
    rng = np.random.default_rng(11)
    h = None
    if case == "gauss2d_2048x128":                                         # the bench case of tools/real_bench.py
        L, R, T = 2048, 128, 3
        n = L * L
        ea, eb, _ = exact.square_lattice_edges(L, L, 1.0)
        ej = rng.normal(size=len(ea))
        slots, check = 4, (0, 31, 64, 127)
    elif case == "gauss3d_128":
        L, R, T = 128, 32, 2
        n = L ** 3
        ids = np.arange(n, dtype=np.uint64).reshape(L, L, L)
        ea = np.concatenate([ids.ravel()] * 3)
        eb = np.concatenate([np.roll(ids, -1, axis=a).ravel() for a in (2, 1, 0)])
        ej = rng.normal(size=len(ea))
        slots, check = 7, (0, 13, 31)
    else:
        n, R, T = 200_000, 64, 2
        # a random graph of degree <= 15 (a configuration-model pairing with rejection of loops and overflow)
        stubs = np.repeat(np.arange(n, dtype=np.uint64), 12)
        rng.shuffle(stubs)
        a, b = stubs[0::2], stubs[1::2]
        keep = a != b
        ea, eb = a[keep], b[keep]
        lo, hi = np.minimum(ea, eb), np.maximum(ea, eb)
        _, first = np.unique(lo * np.uint64(n) + hi, return_index=True)   # drop duplicated bonds: keeps the degree <= 12
        ea, eb = ea[np.sort(first)], eb[np.sort(first)]
        extra = rng.integers(0, 1000, size=(1500, 2)).astype(np.uint64)   # raise a few degrees beyond 12
        extra = extra[extra[:, 0] != extra[:, 1]][:1200]
        ea, eb = np.concatenate([ea, extra[:, 0]]), np.concatenate([eb, extra[:, 1]])
        deg = np.bincount(np.concatenate([ea, eb]).astype(np.int64), minlength=n)
        while deg.max() > 15:                                               # trim the rare overflow
            worst = int(np.argmax(deg))
            drop = np.flatnonzero((ea == worst) | (eb == worst))[-1]
            ea, eb = np.delete(ea, drop), np.delete(eb, drop)
            deg = np.bincount(np.concatenate([ea, eb]).astype(np.int64), minlength=n)
        assert 12 <= deg.max() <= 15
        ej = rng.normal(size=len(ea)) * 0.4
        h = rng.normal(size=n) * 0.3
        slots, check = 15, (0, 33, 63)
    g = capi.Graph(ea, eb, ej, nvars=n, biases=h, force_general=True)
    assert g.info.real_slots == slots and g.info.real_heavy_sites == 0
    st = capi.States(g, capi.make_seeds(3, R))
    eps = st.do_time_steps(T, 0.9, per_step_energies=True)
    e = st.energies()
    assert np.array_equal(eps[:, -1], e)
    spins = st.states()
    scale = float(np.abs(ej).sum() + (0.0 if h is None else np.abs(h).sum()))
    terms = len(ej) + (0 if h is None else n)
    kE = g.info.real_energy_log2
    bound = terms * 2.0 ** (kE - 25) + 4 * np.finfo(np.float64).eps * scale
    worst = 0.0
    for r in check:
        ref = _f64_energy(ea, eb, ej, spins[r], h)
        err = abs(float(np.longdouble(e[r]) - ref))
        assert err <= bound, (case, r, err, bound)
        assert err <= 1e-13 * scale, (case, r, err / scale)
        worst = max(worst, err / scale)
    # the energy after an EARLIER timestep is the f64 energy of that configuration too: a second container stopped there
    st2 = capi.States(g, capi.make_seeds(3, R))
    st2.do_time_steps(T - 1, 0.9)
    assert np.array_equal(st2.energies(), eps[:, T - 2])
    r = check[1]
    assert abs(float(np.longdouble(eps[r, T - 2]) - _f64_energy(ea, eb, ej, st2.states()[r], h))) <= bound
    print(f"K1 {case}: worst |E_gpu - E_f64| / sum|terms| = {worst:.2e}, bound {bound / scale:.2e}")


def test_one_pinning_bias_stays_on_the_real_path_and_matches_engine_e(capi, oracle, exact):
    """set_individual_bias(7, 1e6) on a ferromagnet and on a Gaussian glass with fields: the biased site is HEAVY (its own
    quantum, the bound shifted by d), everything bit-exact against engine E -- constant beta, a schedule from hot (the pinned
    site's beta dE ~ 1) to cold, per-replica betas, partial groups."""
    W, H = 48, 20
    n = W * H
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    rng = np.random.default_rng(5)
    for ej_case, h in ((ej, np.zeros(n)), (rng.normal(size=len(ea)), rng.normal(size=n) * 0.3)):
        h = h.copy()
        h[7] = 1e6
        h[333] = -4e3
        assert oracle.rj_eligible(ea, eb, ej_case, n, h)
        g = capi.Graph(ea, eb, ej_case, nvars=n, biases=h, stable_path=True)
        assert g.kind == capi.KIND_GENERAL and g.info.real_slots == 4 and g.info.real_heavy_sites == 2
        for R, T, kw in ((40, 6, dict(betas=[0.7] * 6)), (3, 5, dict(betas=list(np.geomspace(2e-7, 3.0, 5)))),
                         (37, 4, dict(beta_replica=np.geomspace(1e-6, 2.0, 37)))):
            seeds = capi.make_seeds(9, R)
            st = capi.States(g, seeds)
            if "beta_replica" in kw:
                st.set_betas(kw["beta_replica"])
                eps = st.do_time_steps(T, per_step_energies=True)
            else:
                eps = st.do_time_steps(T, kw["betas"], per_step_energies=True)
            e_ref, s_ref, eps_ref = oracle.rj_run(ea, eb, ej_case, n, seeds, T, biases=h, per_step=True, **kw)
            np.testing.assert_array_equal(st.states().astype(np.uint8), s_ref[:R])
            np.testing.assert_array_equal(eps, eps_ref)
            np.testing.assert_array_equal(st.energies(), e_ref)
            if "beta_replica" not in kw and kw["betas"][-1] >= 0.5:      # cold enough: the pinned spins point along their fields
                assert st.states()[:, 7].all() and not st.states()[:, 333].any()


def test_pinned_spin_k2_against_exact_enumeration(capi, exact):
    """16 spins (4 x 4 torus, Gaussian couplings) with spin 5 pinned by h = 1e6 and a moderate field on spin 9: <E>, <|M|> from
    the HIP path within 4 sigma of the exact Boltzmann averages (the pinned spin contributes its -h s to every energy)."""
    rng = np.random.default_rng(17)
    ea, eb, _ = exact.square_lattice_edges(4, 4, 1.0)
    ej = rng.normal(size=len(ea))
    h = np.zeros(16)
    h[5], h[9] = 1e6, 0.8
    beta = 0.7
    ex = exact.enumerate_graph(ea, eb, ej, 16, beta, h)
    g = capi.Graph(ea, eb, ej, nvars=16, biases=h, stable_path=True)
    assert g.info.real_slots == 4 and g.info.real_heavy_sites == 1
    R = 256
    st = capi.States(g, capi.make_seeds(23, R))
    st.do_time_steps(200, beta)
    e_acc, m_acc, n_s = np.zeros(R), np.zeros(R), 0
    for _ in range(300):
        st.do_time_steps(4, beta)
        e_acc += st.energies()
        m_acc += np.abs(st.magnetisations())
        n_s += 1
    assert st.states()[:, 5].all()
    for got, want in ((e_acc / n_s, ex["E"]), (m_acc / n_s, ex["absM"])):
        z = (got.mean() - want) / (got.std(ddof=1) / np.sqrt(R))
        # <E> ~ -1e6: the statistical error is that of the fluctuating part, which f64 resolves to 1e-10
        assert abs(z) < 4.0, (z, got.mean(), want)


def test_degrees_16_to_31(capi, oracle):
    """23 and 31 ELL slots (six / eight index nibbles, three / four transpositions, 64-thread workgroups): random graphs of
    maximum degree 20 and 31 with Gaussian couplings and fields against engine E."""
    rng = np.random.default_rng(41)
    n = 300
    for maxdeg, m, slots in ((20, 2600, 23), (31, 4000, 31)):
        pairs, deg = set(), np.zeros(n, dtype=int)
        while len(pairs) < m:
            a, b = (int(v) for v in rng.integers(0, n - 4, 2))
            if a != b and deg[a] < maxdeg and deg[b] < maxdeg and (min(a, b), max(a, b)) not in pairs:
                pairs.add((min(a, b), max(a, b)))
                deg[a] += 1
                deg[b] += 1
        assert deg.max() > (15 if slots == 23 else 23)
        pairs = sorted(pairs)
        rng.shuffle(pairs)
        ea = np.array([p[0] for p in pairs], dtype=np.uint64)
        eb = np.array([p[1] for p in pairs], dtype=np.uint64)
        ej, h = rng.normal(size=len(ea)) * 0.3, rng.normal(size=n) * 0.4
        g = capi.Graph(ea, eb, ej, nvars=n, biases=h, stable_path=True)
        assert g.info.real_slots == slots
        for R, T, kw in ((37, 4, dict(betas=[0.6] * 4)), (64, 3, dict(betas=[0.2, 0.9, 1.5])), (5, 3, dict(beta_replica=np.linspace(0.1, 1.2, 5)))):
            seeds = capi.make_seeds(77, R)
            st = capi.States(g, seeds)
            if "beta_replica" in kw:
                st.set_betas(kw["beta_replica"])
                eps = st.do_time_steps(T, per_step_energies=True)
            else:
                eps = st.do_time_steps(T, kw["betas"], per_step_energies=True)
            e_ref, s_ref, eps_ref = oracle.rj_run(ea, eb, ej, n, seeds, T, biases=h, per_step=True, **kw)
            np.testing.assert_array_equal(st.states().astype(np.uint8), s_ref[:R])
            np.testing.assert_array_equal(eps, eps_ref)
            np.testing.assert_array_equal(st.energies(), e_ref)
            np.testing.assert_array_equal(st.magnetisations(), 2 * s_ref[:R].sum(axis=1).astype(np.int64) - n)


def test_stable_path_makes_experiments_prefix_stable(capi, exact):
    """lattice.rs:83-91, 198: experiment k depends on seed k alone, so row 0 of run_monte_carlo(beta, T, 1) equals row 0 of
    run_monte_carlo(beta, T, 512).  Here the kernel family follows the experiment count on SMALL graphs (the LDS-resident f64 CSR
    kernel for few experiments, replica-packed for many: two different chains); with ISINGMC_FLAG_STABLE_PATH it follows the graph
    alone.  Big graphs (beyond the resident kernel) take the packed family from ONE experiment on since round 4."""
    W, H = 64, 64                                                           # 4 096 sites: the LDS-resident CSR kernel takes few experiments
    ea, eb, _ = exact.square_lattice_edges(W, H, 1.0)
    ej = np.random.default_rng(3).normal(size=len(ea))
    T, beta = 5, 0.8
    seeds = capi.make_seeds(1, 512)
    res = {}
    for stable in (True, False):
        g = capi.Graph(ea, eb, ej, nvars=W * H, stable_path=stable)
        assert g.info.stable_path == int(stable)
        for R in (1, 2, 512):
            st = capi.States(g, seeds[:R])
            st.do_time_steps(T, beta)
            res[stable, R] = (st.states(), st.energies())
    for R in (1, 2):
        assert np.array_equal(res[True, R][0], res[True, 512][0][:R]) and np.array_equal(res[True, R][1], res[True, 512][1][:R])
    assert np.array_equal(res[True, 512][0], res[False, 512][0])          # the flag changes nothing where the packed family is chosen anyway
    assert np.array_equal(res[False, 1][0], res[False, 2][0][:1])         # both on the CSR kernels
    # without the flag ONE experiment runs on the f64 CSR kernels: another chain for the same Hamiltonian (documented)
    assert not np.array_equal(res[False, 1][0][0], res[False, 512][0][0])
    # a big graph: stable without the flag (real-coupling family from one experiment on)
    Wb, Hb = 160, 128
    eab, ebb, _ = exact.square_lattice_edges(Wb, Hb, 1.0)
    gb = capi.Graph(eab, ebb, np.random.default_rng(4).normal(size=len(eab)), nvars=Wb * Hb)
    one, many = capi.States(gb, seeds[:1]), capi.States(gb, seeds[:40])
    one.do_time_steps(T, beta)
    many.do_time_steps(T, beta)
    assert np.array_equal(one.states()[0], many.states()[0]) and one.energies()[0] == many.energies()[0]
    # uniform-|J| graphs (bit-sliced packed family): the same guarantee with the flag
    ea3, eb3, ej3 = exact.square_lattice_edges(96, 64, -1.0)
    g3 = capi.Graph(ea3, eb3, ej3, force_general=True, stable_path=True)
    a = capi.States(g3, seeds[:1]); a.do_time_steps(T, 0.4)
    b = capi.States(g3, seeds[:40]); b.do_time_steps(T, 0.4)
    assert np.array_equal(a.states()[0], b.states()[0])
