"""GPU parity of the replica-packed REAL-COUPLING path (csrc/real_kernels.hpp) against oracle engine E,
bit for bit: Gaussian couplings on a 2-d lattice, a random real-J graph with biases (degree up to 7), one biased site on
an otherwise uniform lattice (Lattice.set_individual_bias, lattice.rs:104-127), per-replica betas, per-step energies
(lattice.rs:445-455), sampling, shards that cut a replica group, and the Boltzmann averages of a 16-spin graph (K2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_graph(rng, n, m, maxdeg, skip=0):
    pairs, deg = set(), np.zeros(n, dtype=int)
    while len(pairs) < m:
        a, b = (int(v) for v in rng.integers(0, n - skip, 2))
        if a != b and deg[a] < maxdeg and deg[b] < maxdeg and (min(a, b), max(a, b)) not in pairs:
            pairs.add((min(a, b), max(a, b)))
            deg[a] += 1
            deg[b] += 1
    pairs = sorted(pairs)
    rng.shuffle(pairs)
    return np.array([p[0] for p in pairs], dtype=np.uint64), np.array([p[1] for p in pairs], dtype=np.uint64)


def _case(capi, oracle, ea, eb, ej, nvars, R, T, beta=None, beta_replica=None, biases=None, initial=None, slots=None):
    seeds = capi.make_seeds(77, R)
    g = capi.Graph(ea, eb, ej, nvars=nvars, biases=biases)
    assert g.kind == capi.KIND_GENERAL and g.info.real_slots == (slots or g.info.real_slots) and g.info.real_slots in (4, 7, 11, 15)
    assert g.info.real_quantum_log2 == oracle.rj_quantise(ea, eb, ej, nvars, biases)[0]
    st = capi.States(g, seeds, initial_state=initial)
    ref_states = None if initial is None else np.tile(np.asarray(initial, dtype=np.uint8), (32 * ((R + 31) // 32), 1))
    if beta_replica is not None:
        st.set_betas(beta_replica)
        eps = st.do_time_steps(T, per_step_energies=True)
        e_ref, s_ref, eps_ref = oracle.rj_run(ea, eb, ej, nvars, seeds, T, beta_replica=beta_replica, biases=biases,
                                              states=ref_states, per_step=True)
    else:
        betas = [beta] * T if np.ndim(beta) == 0 else beta
        eps = st.do_time_steps(T, beta, per_step_energies=True)
        e_ref, s_ref, eps_ref = oracle.rj_run(ea, eb, ej, nvars, seeds, T, betas=betas, biases=biases, states=ref_states,
                                              per_step=True)
    np.testing.assert_array_equal(st.states().astype(np.uint8), s_ref[:R])
    np.testing.assert_array_equal(eps, eps_ref)            # exact integer sums scaled by a power of two: bit-equal
    np.testing.assert_array_equal(st.energies(), e_ref)
    np.testing.assert_array_equal(st.magnetisations(), 2 * s_ref[:R].sum(axis=1).astype(np.int64) - nvars)
    return st, s_ref, seeds


def test_gaussian_couplings_on_a_square_lattice(capi, oracle, exact, monkeypatch):
    """2-d Edwards-Anderson glass with Gaussian J: degree 4 -> the one-table kernel (slots = 4)."""
    monkeypatch.setenv("ISINGMC_FORCE_REAL", "1")
    W, H = 48, 20
    ea, eb, _ = exact.square_lattice_edges(W, H, 1.0)
    ej = np.random.default_rng(2024).normal(size=len(ea))
    _case(capi, oracle, ea, eb, ej, W * H, R=40, T=6, beta=0.9, slots=4)                      # partial last group
    _case(capi, oracle, ea, eb, ej, W * H, R=64, T=5, beta=np.geomspace(0.1, 3.0, 5), slots=4)  # annealing schedule
    _case(capi, oracle, ea, eb, ej, W * H, R=3, T=4, beta=0.0, slots=4)
    _case(capi, oracle, ea, eb, ej, W * H, R=33, T=4, beta=40.0, slots=4)                     # deep quench
    init = (np.arange(W * H) % 3 == 0).astype(np.uint8)
    _case(capi, oracle, ea, eb, ej, W * H, R=20, T=4, beta=0.7, initial=init, slots=4)
    _case(capi, oracle, ea, eb, ej, W * H, R=37, T=5, beta_replica=np.linspace(0.05, 2.5, 37), slots=4)


def test_random_graph_with_real_couplings_and_biases(capi, oracle, monkeypatch):
    """degree up to 7 -> the two-table kernel (slots = 7); isolated sites, a self-loop, a duplicated bond, biases."""
    monkeypatch.setenv("ISINGMC_FORCE_REAL", "1")
    rng = np.random.default_rng(8)
    n = 700
    ea, eb = _random_graph(rng, n, 1900, 7, skip=9)
    ea = np.concatenate([ea, [5, ea[3]]]).astype(np.uint64)
    eb = np.concatenate([eb, [5, eb[3]]]).astype(np.uint64)
    ej = rng.normal(size=len(ea)) * 0.37
    h = rng.normal(size=n) * 0.2
    _case(capi, oracle, ea, eb, ej, n, R=45, T=6, beta=1.1, biases=h, slots=7)
    _case(capi, oracle, ea, eb, ej, n, R=32, T=5, beta=np.linspace(0.2, 2.0, 5), biases=h, slots=7)
    _case(capi, oracle, ea, eb, ej, n, R=70, T=4, beta_replica=np.linspace(3.0, 0.1, 70), biases=h, slots=7)
    st, s_ref, seeds = _case(capi, oracle, ea, eb, ej, n, R=20, T=3, beta=0.5, slots=7)       # no biases
    # continue the same container: the timestep counter carries on, set_state replaces one replica
    new = (np.arange(n) % 2).astype(np.uint8)
    st.set_state(4, new)
    s_ref[4] = new
    st.do_time_steps(3, 0.8)
    _, s2 = oracle.rj_run(ea, eb, ej, n, seeds, 3, betas=[0.8] * 3, states=s_ref, t0=3)
    np.testing.assert_array_equal(st.states().astype(np.uint8), s2[:20])


def test_degrees_up_to_15(capi, oracle, exact, monkeypatch):
    """11 and 15 slots: three / four index nibbles, two transpositions, 128-thread workgroups -- a square lattice with second
    neighbours (J1-J2, degree 8), a random graph of degree <= 11 and one of degree <= 15, Gaussian couplings and biases."""
    monkeypatch.setenv("ISINGMC_FORCE_REAL", "1")
    rng = np.random.default_rng(31)
    W, H = 20, 14
    ids = np.arange(W * H, dtype=np.uint64).reshape(H, W)
    nb = [np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0), np.roll(np.roll(ids, -1, axis=0), -1, axis=1),
          np.roll(np.roll(ids, -1, axis=0), 1, axis=1)]
    ea = np.concatenate([ids.ravel()] * 4)
    eb = np.concatenate([n.ravel() for n in nb])
    ej = np.concatenate([np.full(2 * W * H, 1.0), np.full(2 * W * H, 0.45)]) * rng.choice([-1.0, 1.0], 4 * W * H)
    _case(capi, oracle, ea, eb, ej, W * H, R=40, T=5, beta=0.7, slots=11)
    _case(capi, oracle, ea, eb, ej, W * H, R=33, T=4, beta_replica=np.linspace(0.1, 1.2, 33), biases=rng.normal(size=W * H) * 0.3, slots=11)
    n = 400
    for maxdeg, m, slots in ((11, 1800, 11), (15, 2600, 15)):
        ga, gb = _random_graph(rng, n, m, maxdeg, skip=5)
        gj, gh = rng.normal(size=len(ga)), rng.normal(size=n) * 0.4
        deg = np.bincount(np.concatenate([ga, gb]).astype(np.int64), minlength=n).max()
        assert deg > (7 if slots == 11 else 11)
        _case(capi, oracle, ga, gb, gj, n, R=37, T=5, beta=0.6, biases=gh, slots=slots)
        _case(capi, oracle, ga, gb, gj, n, R=64, T=4, beta=np.linspace(0.2, 1.5, 4), slots=slots)


def test_one_biased_site_on_a_uniform_lattice_through_the_python_api(oracle, exact):
    """Lattice.set_individual_bias (lattice.rs:104-127) on a 128 x 128 ferromagnet: from 2 experiments on the real-coupling
    path serves it (integer couplings quantise exactly: the energies equal the f64 energy of the configuration)."""
    import py_monte_carlo
    W = H = 128                                                     # 16 384 sites: above the LDS-resident bound
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    lat = py_monte_carlo.Lattice.from_arrays(ea, eb, ej, seed_gen=9)
    lat.set_individual_bias(7, -3.0)
    info = lat.engine_info()
    assert info["kind"] == "general" and info["real_slots"] == 4
    R, T, beta = 40, 5, 0.42
    e, s = lat.run_monte_carlo(beta, T, R)
    h = np.zeros(W * H)
    h[7] = -3.0
    seeds = np.array(lat.make_seeds(R), dtype=np.uint64)
    e_ref, s_ref = oracle.rj_run(ea, eb, ej, W * H, seeds, T, betas=[beta] * T, biases=h)
    assert np.array_equal(s, s_ref[:R].astype(bool)) and np.array_equal(e, e_ref)
    for r in (0, 17, 39):                                                          # K1 against the f64 Hamiltonian
        assert e[r] == oracle.energy(ea, eb, ej, W * H, s[r].astype(np.uint8), h)
    # shards that cut a replica group, per-step energies, sampling
    for cuts in ([(0, 5), (5, 37), (37, 40)], [(0, 16), (16, 40)]):
        parts = [lat.run_monte_carlo(beta, T, R, replica_range=r) for r in cuts]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), e) and np.array_equal(np.concatenate([p[1] for p in parts]), s)
    stops = [(0, 0.2), (T, 0.9)]
    ea_full, sa_full = lat.run_monte_carlo_annealing_and_get_energies(stops, T, R)
    e_ann, s_ann = lat.run_monte_carlo_annealing(stops, T, R)
    assert np.array_equal(ea_full[:, -1], e_ann) and np.array_equal(sa_full, s_ann)
    ea_part, _ = lat.run_monte_carlo_annealing_and_get_energies(stops, T, R, replica_range=(5, 37))
    assert np.array_equal(ea_part, ea_full[5:37])
    es, ss = lat.run_monte_carlo_sampling(beta, 6, R, None, 2, 2)
    assert es.shape == (R, 3) and ss.shape == (R, 3, W * H)
    _, s_t4 = oracle.rj_run(ea, eb, ej, W * H, seeds, 4, betas=[beta] * 4, biases=h)
    assert np.array_equal(ss[:, 0], s_t4[:R].astype(bool))
    for r in (1, 22):
        assert es[r, 0] == oracle.energy(ea, eb, ej, W * H, ss[r, 0].astype(np.uint8), h)
    # few experiments: a partly used replica group (only the owned replicas' random numbers are drawn)
    e3, s3 = lat.run_monte_carlo(beta, T, 3)
    assert np.array_equal(e3, e_ref[:3]) and np.array_equal(s3, s_ref[:3].astype(bool))       # the same seeds: the same chains
    # a single experiment runs on the f64 CSR path: a different (equally valid) chain
    e1, s1 = lat.run_monte_carlo(beta, T, 1)
    e1_ref = [oracle.gen_run(ea, eb, ej, W * H, int(sd), [beta] * T, biases=h)[0] for sd in lat.make_seeds(1)]
    np.testing.assert_allclose(e1, e1_ref, rtol=1e-12)


def test_per_replica_betas_on_any_shard(capi, oracle, exact, monkeypatch):
    """The real-coupling path decides every replica on its own (no ties numbered over a group): per-replica betas work on a
    shard that starts and ends inside a group, and the union of the shards is the unsharded run."""
    monkeypatch.setenv("ISINGMC_FORCE_REAL", "1")
    ea, eb, _ = exact.square_lattice_edges(20, 12, 1.0)
    rng = np.random.default_rng(5)
    ej = rng.normal(size=len(ea))
    R, T = 70, 4
    betas = np.linspace(0.1, 1.5, R)
    seeds = capi.make_seeds(13, R)
    g = capi.Graph(ea, eb, ej, nvars=240)
    _, s_ref = oracle.rj_run(ea, eb, ej, 240, seeds, T, beta_replica=betas)
    got = []
    for lo, hi in [(0, 7), (7, 45), (45, 70)]:
        st = capi.States(g, seeds, replica_range=(lo, hi))
        st.set_betas(betas[lo:hi])
        st.do_time_steps(T)
        got.append(st.states().astype(np.uint8))
    np.testing.assert_array_equal(np.concatenate(got), s_ref[:R])


def test_k2_boltzmann_averages_on_the_gpu(capi, oracle, exact, monkeypatch):
    """K2 for the new kernels: 16 spins, Gaussian couplings and biases, exact enumeration; 3 sigma."""
    monkeypatch.setenv("ISINGMC_FORCE_REAL", "1")
    rng = np.random.default_rng(17)
    n = 16
    ea, eb = _random_graph(rng, n, 30, 6)
    ej, h = rng.normal(size=len(ea)), rng.normal(size=n) * 0.4
    beta = 0.55
    ex = exact.enumerate_graph(ea, eb, ej, n, beta, h)
    R = 256
    g = capi.Graph(ea, eb, ej, nvars=n, biases=h)
    st = capi.States(g, capi.make_seeds(3, R))
    st.do_time_steps(200, beta)
    eps = st.do_time_steps(4000, beta, per_step_energies=True)
    m = eps.mean(axis=1)
    z = (m.mean() - ex["E"]) / (m.std(ddof=1) / np.sqrt(R))
    assert abs(z) < 3.0, (z, m.mean(), ex["E"])
    mags = []
    for _ in range(200):
        st.do_time_steps(5, beta)
        mags.append(np.abs(st.magnetisations()))
    mm = np.mean(mags, axis=0)
    zm = (mm.mean() - ex["absM"]) / (mm.std(ddof=1) / np.sqrt(R))
    assert abs(zm) < 3.0, (zm, mm.mean(), ex["absM"])


def test_tempering_on_the_real_coupling_path(capi, oracle, exact, monkeypatch):
    """Parallel tempering of a Gaussian glass -- the use the per-replica acceptance scales exist for: the classical ladder
    (host swap step, tempering.rs:172-212's loop) on the HIP path against the same host logic driven by oracle engine E."""
    from helpers import OracleRjEngine
    from pyisingmontecarlo_amd.tempering import ClassicalTempering
    monkeypatch.setenv("ISINGMC_FORCE_REAL", "1")
    W, H, G = 24, 16, 20
    ea, eb, _ = exact.square_lattice_edges(W, H, 1.0)
    ej = np.random.default_rng(11).normal(size=len(ea))
    runs = []
    for factory in (None, lambda: OracleRjEngine(ea, eb, ej, W * H)):
        pt = ClassicalTempering((ea, eb, ej), seed=5, engine_factory=factory)
        for b in np.linspace(0.3, 1.4, G):
            pt.add_graph(float(b))
        pt.timesteps(4)
        pt.timesteps(12, replica_swap_freq=3)
        states, energies = pt.timesteps_sample(12, replica_swap_freq=2, sampling_freq=4)
        runs.append((states, energies, pt.get_permutation(), pt.get_total_swaps()))
    assert runs[0][3] == runs[1][3] > 0
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)


def test_classic_ising_on_the_packed_paths_and_append(oracle, exact, monkeypatch):
    """ClassicIsing (classicising.rs:27-110) with 16+ experiments on a graph that is not a recognised lattice: the constructor
    creates them at once, so the replica-packed kernels serve them (bit-sliced path: uniform |J|; real-coupling path:
    anything else), and add_graph grows the container -- into the open group, or opening a new one at a multiple of 32."""
    import py_monte_carlo
    W, H = 136, 136                                               # 18 496 sites, not 64-wide: general path; x 31 experiments: past the packed paths' break-even
    N = W * H
    ea, eb, ej_u = exact.square_lattice_edges(W, H, -1.0)
    ej_g = np.random.default_rng(3).normal(size=len(ea))
    edges = lambda ej: [((int(a), int(b)), float(j)) for a, b, j in zip(ea, eb, ej)]
    for ej, run in ((ej_u, oracle.pk_run), (ej_g, oracle.rj_run)):
        ci = py_monte_carlo.ClassicIsing(edges(ej), None, 31, 42)
        seeds = list(oracle.make_seeds(42, 31))
        ci.run_monte_carlo(0.6, 2)
        _, ref = run(ea, eb, ej, N, np.array(seeds, dtype=np.uint64), 2, betas=[0.6] * 2)
        assert np.array_equal(ci.get_states(), ref[:31].astype(bool))
        # replica 31 joins the open group.  Bit-sliced path: it takes over the chain bit 31 has been running since t = 0 (the
        # unused replicas of a group are simulated: its tie numbering needs them).  Real-coupling path: bits a container does
        # not own are not simulated, and the new replica starts from its random start now (t = 2), like on any other path.
        ci.add_graph()
        seeds.append(int(oracle.make_seeds(42, 32)[-1]))
        if run is oracle.rj_run:
            _, start0 = run(ea, eb, ej, N, np.array(seeds, dtype=np.uint64), 0, betas=[])
            ref[31] = start0[31]
        assert ci.get_num_graphs() == 32 and np.array_equal(ci.get_states()[31], ref[31].astype(bool))
        # replica 32 opens a new group, keyed by its seed, started now (t = 2); replica 33 comes with an explicit state
        ci.add_graph()
        seeds.append(int(oracle.make_seeds(42, 33)[-1]))
        up = [True] * N
        ci.add_graph(up)
        seeds.append(int(oracle.make_seeds(42, 34)[-1]))
        ci.run_monte_carlo(0.6, 3)
        e_all, s_all = ci.get_energies(), ci.get_states()
        e0, ref0 = run(ea, eb, ej, N, np.array(seeds[:32], dtype=np.uint64), 3, betas=[0.6] * 3, states=ref, t0=2)
        assert np.array_equal(s_all[:32], ref0[:32].astype(bool)) and np.array_equal(e_all[:32], e0)
        # the second group on its own: random start, then replica 33 (bit 1) set to all up, three timesteps from t = 2
        _, start = run(ea, eb, ej, N, np.array(seeds[32:], dtype=np.uint64), 0, betas=[])
        start[1] = 1
        e1, ref1 = run(ea, eb, ej, N, np.array(seeds[32:], dtype=np.uint64), 3, betas=[0.6] * 3, states=start, t0=2)
        assert np.array_equal(s_all[32:], ref1[:2].astype(bool)) and np.array_equal(e_all[32:], e1)


def test_on_stream_tempering_shards_of_a_gaussian_glass(capi, oracle, exact, monkeypatch):
    """The multi-GPU exchange protocol of the real-coupling path with both 'ranks' on this GPU: 40 rungs of a Gaussian glass
    cut 24 + 16 (the cut falls inside a replica group), the all-gather between pt_measure and pt_swap emulated by device
    copies on each engine's stream.  Equal to the unsharded on-stream ladder and to the host swap step on oracle engine E."""
    import torch
    monkeypatch.setenv("ISINGMC_FORCE_REAL", "1")
    W, H, G, per = 24, 16, 40, 24
    ea, eb, _ = exact.square_lattice_edges(W, H, 1.0)
    ej = np.random.default_rng(12).normal(size=len(ea))
    g = capi.Graph(ea, eb, ej, nvars=W * H)
    seeds = capi.make_seeds(3, G)
    betas = np.linspace(0.3, 1.5, G)
    full = capi.States(g, seeds)
    full.pt_attach(betas, 0, G, 1, 99)
    shards = [capi.States(g, seeds, replica_range=(0, per)), capi.States(g, seeds, replica_range=(per, G))]
    for k, sh in enumerate(shards):
        sh.pt_attach(betas, per * k, per, 2, 99)
    bufs = [sh.pt_buffers() for sh in shards]
    streams = [sh.pt_stream() for sh in shards]
    # the host twin: oracle engine E + isingmc_host_pt_swap_round
    perm_ref = np.arange(G, dtype=np.uint32)
    st_ref, swaps_ref, t = None, 0, 0
    for rnd in range(8):
        full.pt_time_steps(2); full.pt_measure(); full.pt_swap()
        for sh in shards:
            sh.pt_time_steps(2)
            sh.pt_measure()
        for sh in shards:
            sh.synchronize()
        for k in range(2):                                  # "all-gather": rank-major concatenation of the locals (per slots each)
            with torch.cuda.stream(streams[k]):
                bufs[k][1][:per].copy_(bufs[0][0])
                bufs[k][1][per:].copy_(bufs[1][0])
        for sh in shards:
            sh.pt_swap()
        beta_of_slot = np.empty(G)
        beta_of_slot[perm_ref] = betas
        e_ref, st_ref = oracle.rj_run(ea, eb, ej, W * H, seeds, 2, beta_replica=beta_of_slot, states=st_ref, t0=t)
        t += 2
        swaps_ref += capi.pt_swap_round(99, rnd, betas, e_ref, perm_ref)
    perm, rounds, swaps = full.pt_state()
    assert rounds == 8 and swaps == swaps_ref > 0 and np.array_equal(perm, perm_ref)
    assert np.array_equal(full.states().astype(np.uint8), st_ref[:G])
    for sh in shards:
        p, r, s = sh.pt_state()
        assert np.array_equal(p, perm) and r == rounds and s == swaps
    assert np.array_equal(np.concatenate([sh.states() for sh in shards]), full.states())


def test_on_stream_tempering_on_the_bit_sliced_packed_path(capi, oracle, exact, monkeypatch):
    """The 3-d +-J Edwards-Anderson glass, the standard tempering workload: uniform |J| -> the bit-sliced packed path (engine D).
    Exchange rounds on the stream (per-slot thresholds relabelled by pt_swap_kernel, the groups' bit-sliced tables rebuilt by
    pk_tables_from_slots_kernel) against the host swap step on the oracle engine; then two group-aligned shards."""
    import torch
    from helpers import OracleRjEngine
    from pyisingmontecarlo_amd.tempering import ClassicalTempering
    monkeypatch.setenv("ISINGMC_FORCE_PACKED", "1")
    L, G = 8, 40
    ea, eb, _ = exact.cubic_lattice_edges(L, 1.0)
    ej = np.random.default_rng(21).choice([-1.0, 1.0], size=len(ea))
    runs = []
    for factory in (None, lambda: OracleRjEngine(ea, eb, ej, L ** 3, bit_sliced=True)):
        pt = ClassicalTempering((ea, eb, ej), seed=5, engine_factory=factory)
        for b in np.linspace(0.2, 1.1, G):
            pt.add_graph(float(b))
        pt.timesteps(3)
        pt.timesteps(14, replica_swap_freq=2)
        states, energies = pt.timesteps_sample(12, replica_swap_freq=3, sampling_freq=4)
        runs.append((states, energies, pt.get_permutation(), pt.get_total_swaps()))
        if factory is None:
            assert pt._on_stream
    assert runs[0][3] == runs[1][3] > 0
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)
    # sharded: 64 rungs cut 32 + 32 (whole groups), the all-gather emulated by device copies on each engine's stream
    G, per = 64, 32
    g = capi.Graph(ea, eb, ej, nvars=L ** 3)
    seeds = capi.make_seeds(3, G)
    betas = np.linspace(0.2, 1.2, G)
    full = capi.States(g, seeds)
    full.pt_attach(betas, 0, G, 1, 99)
    shards = [capi.States(g, seeds, replica_range=(0, per)), capi.States(g, seeds, replica_range=(per, G))]
    for k, sh in enumerate(shards):
        sh.pt_attach(betas, per * k, per, 2, 99)
    bufs = [sh.pt_buffers() for sh in shards]
    streams = [sh.pt_stream() for sh in shards]
    for rnd in range(8):
        full.pt_time_steps(2); full.pt_measure(); full.pt_swap()
        for sh in shards:
            sh.pt_time_steps(2)
            sh.pt_measure()
        for sh in shards:
            sh.synchronize()
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                bufs[k][1][:per].copy_(bufs[0][0])
                bufs[k][1][per:].copy_(bufs[1][0])
        for sh in shards:
            sh.pt_swap()
    perm, rounds, swaps = full.pt_state()
    assert rounds == 8 and swaps > 0
    for sh in shards:
        p, r, s = sh.pt_state()
        assert np.array_equal(p, perm) and r == rounds and s == swaps
    assert np.array_equal(np.concatenate([sh.states() for sh in shards]), full.states())
    with pytest.raises(ValueError, match="multiples of 32"):
        capi.States(g, seeds, replica_range=(0, 40)).pt_attach(betas, 0, 40, 2, 99)
