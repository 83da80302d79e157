"""BASELINE configs c4 and c5 AT FULL SIZE under -m gpu: size-independent properties -- K1 (host
recomputation of returned energies), limits (beta = 0 flips everything, a cold ordered start does not move), equality of the
two packed kernels, energies-after-every-step against the plain annealing call (lattice.rs:395-470 vs 309-385), shard
invariance -- plus the CSR path below 16 experiments against the oracle at a size the oracle finishes in seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _energy(ea, eb, ej, spins_bool):
    s = spins_bool.astype(np.int8) * 2 - 1
    return float(np.sum(ej * (s[ea.astype(np.int64)] * s[eb.astype(np.int64)])))


def test_c4_full_size_annealing_with_energies(exact):
    """c4: 2048^2 +-J glass (default_rng(2024)) x 128 replicas (one GPU's share), geometric schedule 0.1 -> 3.0 given as T stops."""
    import py_monte_carlo
    L, R, T = 2048, 128, 30
    ea, eb, ej = exact.square_lattice_edges(L, L, 1.0, np.random.default_rng(2024))
    lat = py_monte_carlo.Lattice.from_arrays(ea, eb, ej, seed_gen=1)
    info = lat.engine_info()
    assert info["kind"] == "lattice2d" and not info["uniform_sign"]
    stops = [(t, float(0.1 * 30.0 ** (t / (T - 1)))) for t in range(T)]
    e_all, s_all = lat.run_monte_carlo_annealing_and_get_energies(stops, T, R)
    assert e_all.shape == (R, T) and s_all.shape == (R, L * L)
    e_fin, s_fin = lat.run_monte_carlo_annealing(stops, T, R)
    assert np.array_equal(e_all[:, -1], e_fin) and np.array_equal(s_all, s_fin)
    for r in (0, R - 1):                                                  # K1 (integer couplings: exact)
        assert e_fin[r] == _energy(ea, eb, ej, s_fin[r])
    # annealing lowers the energy, and far below the random start's 0 +- sqrt(2N)
    assert np.all(e_all[:, -1] < e_all[:, 0]) and e_fin.max() < -1.3 * L * L
    # shard invariance: replica 77 alone == replica 77 of the batch
    e77, s77 = lat.run_monte_carlo_annealing(stops, T, R, replica_range=(77, 78))
    assert e77[0] == e_fin[77] and np.array_equal(s77[0], s_fin[77])


@pytest.mark.parametrize("L,R,T", [(4096, 48, 3), (1024, 256, 4), (2048, 64, 3), (4096, 32, 9)])  # T >= 8: the plain run uses two stream lanes
def test_c2_energies_after_every_step_leave_the_trajectory_alone(capi, oracle, exact, L, R, T):
    """The uniform-J lattice at sizes whose launches exceed what the chip holds at once (c2's 4096^2; 1024^2 x 256; 2048^2 x 64):
    the run that also returns the energy after every timestep (colour-1 half-sweep fused with the measurement) must end in the
    SAME configurations as the plain run, its last energy must be the energy of that configuration, and a few replicas are
    compared with the oracle.  Round 3 found the fused kernel's vector store racing with the counting code behind it -- only
    under load, so only from ~1500 workgroups per launch on, which no earlier test reached with uniform couplings."""
    ea, eb, ej = exact.square_lattice_edges(L, L, -1.0)
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(5, R)
    ones = np.ones(L * L, dtype=np.uint8)
    beta = 0.55
    plain = capi.States(g, seeds, initial_state=ones)
    plain.do_time_steps(T, beta)
    fused = capi.States(g, seeds, initial_state=ones)
    per_step = fused.do_time_steps(T, beta, per_step_energies=True)
    e_plain, e_fused = plain.energies(), fused.energies()
    assert np.array_equal(e_plain, e_fused)
    assert np.array_equal(per_step[:, -1], e_fused)
    assert np.array_equal(plain.magnetisations(), fused.magnetisations())
    s_plain, s_fused = plain.states(), fused.states()
    assert np.array_equal(s_plain, s_fused)
    lat = oracle.Lat(L, L)
    for r in (1, R - 1):
        ost = lat.pack(ones)
        want = []
        for t in range(T):
            lat.sweep(ost, seeds[r], t, beta)
            want.append(lat.energy_mag(ost)[0])
        assert np.array_equal(lat.unpack(ost), s_fused[r].astype(np.uint8))
        assert per_step[r].tolist() == want


@pytest.mark.parametrize("beta", [0.35, 0.55])
def test_c2_energy_against_kaufman_at_full_size(capi, exact, beta):
    """K3 at the benchmark's own size: <E> of the 4096^2 ferromagnet x 64 replicas against Kaufman's exact finite-torus energy
    on both sides of beta_c, through both measurement paths (energy after every sweep; energies() of plain runs).  The
    standard error is ~2e-6 of the energy: the store bug of round 3 stood out here by 15-100 sigma while every small test
    passed.  Seeded, hence deterministic: the 4 sigma bound cannot flake."""
    L, R = 4096, 64
    ea, eb, ej = exact.square_lattice_edges(L, L, -1.0)
    g = capi.Graph(ea, eb, ej)
    start = np.ones(L * L, dtype=np.uint8) if beta > 0.4407 else None
    st = capi.States(g, capi.make_seeds(int(beta * 100), R), initial_state=start)
    st.do_time_steps(400, beta)
    fused = st.do_time_steps(300, beta, per_step_energies=True).mean(axis=1)
    plain = np.zeros(R)
    for _ in range(40):
        st.do_time_steps(8, beta)
        plain += st.energies()
    plain /= 40
    ref = exact.kaufman_energy(L, L, beta)
    for name, x in (("fused", fused), ("plain", plain)):
        err = x.std(ddof=1) / np.sqrt(R)
        assert err < 1e-5 * abs(ref), (name, err)
        assert abs(x.mean() - ref) < 4.0 * err, (name, beta, x.mean(), ref, err)


def test_c2_full_size_through_the_python_api(exact):
    """c2's 4096^2 ferromagnet through the reference's Python surface: annealing with the energy after every timestep
    (lattice.rs:395-470) against plain annealing (lattice.rs:309-385) and against run_monte_carlo at the last stop's beta -- the
    same configurations from all three, K1 on the returned arrays."""
    import py_monte_carlo
    L, R, T = 4096, 24, 6
    ea, eb, ej = exact.square_lattice_edges(L, L, -1.0)
    lat = py_monte_carlo.Lattice.from_arrays(ea, eb, ej, seed_gen=11)
    assert lat.engine_info()["kind"] == "lattice2d"
    stops = [(0, 0.2), (T, 0.6)]
    e_all, s_all = lat.run_monte_carlo_annealing_and_get_energies(stops, T, R)
    e_fin, s_fin = lat.run_monte_carlo_annealing(stops, T, R)
    assert e_all.shape == (R, T) and s_all.shape == (R, L * L) and s_all.dtype == np.bool_
    assert np.array_equal(e_all[:, -1], e_fin) and np.array_equal(s_all, s_fin)
    for r in (0, R // 2, R - 1):
        assert e_fin[r] == _energy(ea, eb, ej, s_fin[r])
    e_const, s_const = lat.run_monte_carlo_annealing([(0, 0.5), (T, 0.5)], T, R)
    e_plain, s_plain = lat.run_monte_carlo(0.5, T, R)
    assert np.array_equal(e_const, e_plain) and np.array_equal(s_const, s_plain)


def test_c5_full_size_general_path(capi, oracle, exact, monkeypatch):
    """c5: 256^3 cubic lattice, 64 replicas, forced through the general edge-list path (replica-packed kernels)."""
    L, R = 256, 64
    N = L ** 3
    ea, eb, ej = exact.cubic_lattice_edges(L, -1.0)
    g = capi.Graph(ea, eb, ej, nvars=N, force_general=True)
    assert g.kind == capi.KIND_GENERAL and g.info.n_colours == 2 and g.info.packed_degree == 6
    seeds = capi.make_seeds(1, R)
    # K4a: beta = 0 accepts every attempt: one timestep flips every spin
    up = np.ones(N, dtype=np.uint8)
    st = capi.States(g, seeds, initial_state=up)
    st.do_time_steps(1, 0.0)
    assert np.all(st.magnetisations() == -N) and np.all(st.energies() == -3.0 * N)
    # K4b: ordered and cold: nothing moves
    st = capi.States(g, seeds, initial_state=up)
    st.do_time_steps(2, 10.0)
    assert np.all(st.magnetisations() == N) and np.all(st.energies() == -3.0 * N)
    # the one-degree kernel and the general packed kernel take the same decisions; K1 on one replica
    T, beta = 3, 0.2217
    st = capi.States(g, seeds)
    st.do_time_steps(T, beta)
    e_uni, m_uni = st.energies(), st.magnetisations()
    spins = st.states()
    assert e_uni[5] == _energy(ea, eb, ej, spins[5]) and m_uni[5] == 2 * int(spins[5].sum()) - N
    assert np.all(np.abs(e_uni / N + 0.9) < 0.15)                      # a hot-started 3-d lattice after 3 sweeps at beta_c: -0.30 per bond
    del st
    monkeypatch.setenv("ISINGMC_DISABLE_PACKED_UNIFORM", "1")
    st2 = capi.States(g, seeds)
    st2.do_time_steps(T, beta)
    assert np.array_equal(st2.energies(), e_uni) and np.array_equal(st2.magnetisations(), m_uni)
    assert np.array_equal(st2.states()[[0, 31, 32, 63]], spins[[0, 31, 32, 63]])


def test_c5_csr_path_against_the_oracle(capi, oracle, exact, monkeypatch):
    """The f64 CSR kernels (one replica per word set; what small graphs, and graphs the packed paths cannot take, run on) at
    32^3 against oracle engine C.  (Since round 3 a uniform-|J| graph of this size is packed even for two experiments:
    the CSR path is selected explicitly.)"""
    monkeypatch.setenv("ISINGMC_DISABLE_PACKED", "1")
    L = 32
    ea, eb, ej = exact.cubic_lattice_edges(L, -1.0)
    g = capi.Graph(ea, eb, ej, nvars=L ** 3, force_general=True)
    seeds = capi.make_seeds(4, 2)
    st = capi.States(g, seeds)
    betas = [0.2217] * 4
    st.do_time_steps(4, 0.2217)
    spins, e = st.states(), st.energies()
    for r in range(2):
        e_ref, s_ref = oracle.gen_run(ea, eb, ej, L ** 3, int(seeds[r]), betas)
        assert np.array_equal(spins[r].astype(np.uint8), s_ref) and e[r] == e_ref
