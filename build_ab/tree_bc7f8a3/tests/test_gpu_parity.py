"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Integer / bit work (spin configurations, lattice energies) must match BIT FOR BIT; f64 energies of the
general path within 1e-9 relative (the reduction order differs, nothing else).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEEDS = np.array([0x0123456789ABCDEF, 42, 2**64 - 1], dtype=np.uint64)


def _lattice_case(capi, oracle, exact, W, H, J, beta, T, rng=None, per_step=False):
    ea, eb, ej = exact.square_lattice_edges(W, H, J, rng)
    g = capi.Graph(ea, eb, ej)
    assert g.kind == capi.KIND_LATTICE2D, "recogniser must take the checkerboard path"
    st = capi.States(g, SEEDS)
    if rng is None:
        lat = oracle.Lat(W, H, abs(J), int(J > 0))
    else:
        jr = (ej[0::2] > 0).astype(np.uint8)
        jd = (ej[1::2] > 0).astype(np.uint8)
        lat = oracle.Lat(W, H, abs(J), 0, jr, jd)
    ref = [lat.init(s) for s in SEEDS]
    np.testing.assert_array_equal(st.packed(), np.stack(ref), err_msg="initial state")
    eps = st.do_time_steps(T, beta, per_step_energies=per_step)
    ref_eps = np.zeros((len(SEEDS), T))
    for r, s in enumerate(SEEDS):
        for t in range(T):
            lat.sweep(ref[r], s, t, beta)
            ref_eps[r, t] = lat.energy_mag(ref[r])[0]
    np.testing.assert_array_equal(st.packed(), np.stack(ref), err_msg="state after sweeps")
    if per_step:
        np.testing.assert_array_equal(eps, ref_eps)
    em = [lat.energy_mag(x) for x in ref]
    np.testing.assert_array_equal(st.energies(), [e for e, _ in em])
    np.testing.assert_array_equal(st.magnetisations(), [m for _, m in em])
    spins = st.states()
    for r in range(len(SEEDS)):
        np.testing.assert_array_equal(spins[r].astype(np.uint8), lat.unpack(ref[r]))
        # K1: energy recomputed from the returned configuration (README.md:45-46)
        e_k1 = oracle.energy(ea, eb, ej, W * H, spins[r])
        if float(J).is_integer():
            assert st.energies()[r] == e_k1
        else:  # the oracle sums 2N non-integer terms sequentially; the engine returns |J| x integer
            np.testing.assert_allclose(st.energies()[r], e_k1, rtol=1e-9)
    assert st.timestep == T
    return st


@pytest.mark.parametrize("W,H", [(64, 64), (128, 32), (256, 64), (512, 16), (64, 4), (16384, 16), (8192, 6)])
@pytest.mark.parametrize("beta", [0.4407, 0.0, 1.5])
def test_lattice_uniform_ferro_bit_exact(capi, oracle, exact, W, H, beta):
    _lattice_case(capi, oracle, exact, W, H, -1.0, beta, T=6)


def test_lattice_uniform_antiferro_bit_exact(capi, oracle, exact):
    _lattice_case(capi, oracle, exact, 256, 32, 1.0, 0.6, T=6)


def test_lattice_scaled_coupling_bit_exact(capi, oracle, exact):
    _lattice_case(capi, oracle, exact, 128, 64, -0.37, 1.1, T=6)


@pytest.mark.parametrize("W,H", [(64, 64), (256, 32)])
def test_lattice_pm_j_bit_exact(capi, oracle, exact, W, H):
    _lattice_case(capi, oracle, exact, W, H, 1.0, 0.8, T=6, rng=np.random.default_rng(2024), per_step=True)


def test_lattice_negative_beta_always_accepts(capi, oracle, exact):
    _lattice_case(capi, oracle, exact, 64, 64, -1.0, -0.3, T=3)


def test_lattice_per_step_energies(capi, oracle, exact):
    _lattice_case(capi, oracle, exact, 256, 16, -1.0, 0.5, T=9, per_step=True)


def test_lattice_initial_state_and_set_state(capi, oracle, exact):
    W, H = 128, 16
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    rng = np.random.default_rng(7)
    ini = rng.integers(0, 2, W * H).astype(np.uint8)
    g = capi.Graph(ea, eb, ej)
    st = capi.States(g, SEEDS, initial_state=ini)
    lat = oracle.Lat(W, H)
    for r in range(3):
        np.testing.assert_array_equal(st.packed()[r], lat.pack(ini))
    other = rng.integers(0, 2, W * H).astype(np.uint8)
    st.set_state(1, other)
    np.testing.assert_array_equal(st.states()[1].astype(np.uint8), other)
    np.testing.assert_array_equal(st.states()[2].astype(np.uint8), ini)
    st.do_time_steps(4, 0.7)
    ref = lat.pack(other)
    for t in range(4):
        lat.sweep(ref, SEEDS[1], t, 0.7)
    np.testing.assert_array_equal(st.packed()[1], ref)


def test_lattice_per_replica_betas(capi, oracle, exact):
    W, H = 256, 16
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    g = capi.Graph(ea, eb, ej)
    st = capi.States(g, SEEDS)
    betas = [0.2, 0.4407, 0.9]
    st.set_betas(betas)
    st.do_time_steps(5)
    lat = oracle.Lat(W, H)
    for r, s in enumerate(SEEDS):
        ref = lat.init(s)
        for t in range(5):
            lat.sweep(ref, s, t, betas[r])
        np.testing.assert_array_equal(st.packed()[r], ref)


def test_lattice_append_and_continue(capi, oracle, exact):
    """Timesteps continue across calls; an appended replica joins at the current timestep."""
    W, H = 64, 64
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    g = capi.Graph(ea, eb, ej)
    st = capi.States(g, SEEDS[:2])
    st.do_time_steps(3, 0.4)
    st.append(int(SEEDS[2]))
    st.do_time_steps(2, 0.4)
    lat = oracle.Lat(W, H)
    for r, s in enumerate(SEEDS):
        ref = lat.init(s)
        for t in (range(5) if r < 2 else range(3, 5)):
            lat.sweep(ref, s, t, 0.4)
        np.testing.assert_array_equal(st.packed()[r], ref)


def _general_case(capi, oracle, ea, eb, ej, nvars, beta, T, biases=None, force_general=False, initial=None):
    g = capi.Graph(ea, eb, ej, nvars=nvars, biases=biases, force_general=force_general)
    assert g.kind == capi.KIND_GENERAL
    st = capi.States(g, SEEDS, initial_state=initial)
    eps = st.do_time_steps(T, beta, per_step_energies=True)
    spins = st.states()
    energies = st.energies()
    for r, s in enumerate(SEEDS):
        betas = [beta] * T if np.ndim(beta) == 0 else beta
        e_ref, s_ref, eps_ref = oracle.gen_run(ea, eb, ej, nvars, s, betas, biases=biases, initial=initial,
                                               per_step=True)
        np.testing.assert_array_equal(spins[r].astype(np.uint8), s_ref, err_msg=f"replica {r}")
        np.testing.assert_allclose(eps[r], eps_ref, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(energies[r], e_ref, rtol=1e-9, atol=1e-9)
    return st


def test_general_small_lattice_16x16(capi, oracle, exact):
    """BASELINE config c1's lattice (16x16, beta=0.3): not a multiple of 64 wide -> general path."""
    ea, eb, ej = exact.square_lattice_edges(16, 16, -1.0)
    _general_case(capi, oracle, ea, eb, ej, 256, 0.3, T=25)


def test_general_random_real_couplings_with_bias(capi, oracle):
    rng = np.random.default_rng(11)
    n, m = 300, 900
    ea = rng.integers(0, n, m).astype(np.uint64)
    eb = rng.integers(0, n, m).astype(np.uint64)  # includes a few self loops and duplicate edges
    ej = rng.normal(size=m)
    biases = rng.normal(size=n) * 0.5
    _general_case(capi, oracle, ea, eb, ej, n, 0.7, T=20, biases=biases)


def test_general_isolated_sites_and_initial_state(capi, oracle):
    """nvars = max index + 1 (lattice.rs:51-55): unused indices are still spins."""
    ea = np.array([0, 5, 9], dtype=np.uint64)
    eb = np.array([5, 9, 40], dtype=np.uint64)
    ej = np.array([1.0, -2.0, 0.5])
    ini = (np.arange(41) % 3 == 0).astype(np.uint8)
    _general_case(capi, oracle, ea, eb, ej, 41, 0.9, T=12, initial=ini)


def test_general_cubic_forced(capi, oracle, exact):
    """BASELINE config c5's shape (3-d cubic through the general path), small."""
    ea, eb, ej = exact.cubic_lattice_edges(8, -1.0)
    _general_case(capi, oracle, ea, eb, ej, 512, 0.2217, T=10, force_general=True)


def test_general_forced_on_recognisable_lattice(capi, oracle, exact):
    ea, eb, ej = exact.square_lattice_edges(64, 8, -1.0)
    _general_case(capi, oracle, ea, eb, ej, 512, 0.5, T=8, force_general=True)


def test_general_annealing_schedule(capi, oracle, exact):
    ea, eb, ej = exact.square_lattice_edges(12, 10, 1.0, np.random.default_rng(5))
    betas = capi.expand_schedule([(0, 0.1), (10, 2.0)], 15)
    _general_case(capi, oracle, ea, eb, ej, 120, betas, T=15)


def test_readme_three_spin_chain_energies(capi, oracle):
    """README.md:50-53 edge list; hand-checked energies."""
    ea, eb, ej = np.array([0, 1], dtype=np.uint64), np.array([1, 2], dtype=np.uint64), np.array([1.0, -1.0])
    g = capi.Graph(ea, eb, ej)
    for state, e in [((1, 1, 1), 0.0), ((1, 0, 1), 0.0), ((1, 0, 0), -2.0), ((1, 1, 0), 2.0)]:
        st = capi.States(g, SEEDS[:1], initial_state=np.array(state, dtype=np.uint8))
        assert st.energies()[0] == e
        assert oracle.energy(ea, eb, ej, 3, np.array(state, dtype=np.uint8)) == e


def test_many_replicas_chunked_grid(capi, oracle, exact):
    """More replicas than one grid.y launch covers (32768): results must not depend on the chunking."""
    W, H, R = 64, 4, 33000
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(5, R)
    st = capi.States(g, seeds)
    st.do_time_steps(3, 0.5)
    packed = st.packed()
    lat = oracle.Lat(W, H)
    for r in (0, 1, 32767, 32768, 32999):
        ref = lat.init(seeds[r])
        for t in range(3):
            lat.sweep(ref, seeds[r], t, 0.5)
        np.testing.assert_array_equal(packed[r], ref)
    e = st.energies()
    assert e.shape == (R,) and e[32999] == lat.energy_mag(ref)[0]
    # general path with more replicas than one launch of replica groups
    ea, eb, ej = exact.square_lattice_edges(6, 4, -1.0)
    g2 = capi.Graph(ea, eb, ej)
    assert g2.kind == capi.KIND_GENERAL
    st2 = capi.States(g2, seeds[:13])                                       # not a multiple of the replica block (8)
    st2.do_time_steps(5, 0.6)
    spins = st2.states()
    for r in (0, 7, 8, 12):
        _, s_ref = oracle.gen_run(ea, eb, ej, 24, seeds[r], [0.6] * 5)
        np.testing.assert_array_equal(spins[r].astype(np.uint8), s_ref)


def test_empty_and_degenerate_inputs(capi, exact):
    ea, eb, ej = exact.square_lattice_edges(64, 4, -1.0)
    g = capi.Graph(ea, eb, ej)
    st = capi.States(g, np.zeros(0, dtype=np.uint64))                       # zero experiments
    assert st.count == 0
    st.do_time_steps(5, 0.3)
    assert st.energies().shape == (0,) and st.states().shape == (0, 256)
    st.append(9)
    st.do_time_steps(0, 0.3)                                                # zero timesteps
    assert st.timestep == 5 and st.count == 1
    with pytest.raises(ValueError):
        capi.Graph(np.zeros(0), np.zeros(0), np.zeros(0), nvars=3)          # lattice.rs:70-72
    with pytest.raises(ValueError):
        capi.Graph([0], [1], [float("inf")])
    with pytest.raises(ValueError):
        st.do_time_steps(2, float("nan"))
    with pytest.raises(ValueError):
        st.set_state(3, np.zeros(256, dtype=np.uint8))
    with pytest.raises(RuntimeError):
        capi.Graph(ea, eb, ej, device=99)
    single = capi.Graph([0], [0], [2.5], nvars=1)                           # one self-loop: constant energy, free spin
    s1 = capi.States(single, [1])
    s1.do_time_steps(3, 1.0)
    assert s1.energies()[0] == 2.5


@pytest.mark.parametrize("W,H,pm", [(64, 64, False), (128, 32, True), (256, 64, False), (512, 128, True), (64, 4, False)])
def test_resident_kernel_equals_per_colour_launches(capi, exact, monkeypatch, W, H, pm):
    """Small lattices run T timesteps inside one LDS-resident launch; the per-colour launch path
    (ISINGMC_DISABLE_RESIDENT=1) must give the same bits, energies per step included."""
    ea, eb, ej = exact.square_lattice_edges(W, H, 1.0 if pm else -1.0, np.random.default_rng(3) if pm else None)
    betas = np.linspace(0.2, 0.9, 7)
    out = []
    for disable in ("0", "1"):
        monkeypatch.setenv("ISINGMC_DISABLE_RESIDENT", disable)
        g = capi.Graph(ea, eb, ej)
        st = capi.States(g, SEEDS)
        eps = st.do_time_steps(7, betas, per_step_energies=True)
        st.do_time_steps(5, 0.4407)
        st.set_betas([0.3, 0.5, 0.7])
        eps2 = st.do_time_steps(4, per_step_energies=True)
        out.append((st.packed(), eps, eps2, st.energies(), st.magnetisations(), st.timestep))
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)


def test_general_resident_kernel_equals_per_class_launches(capi, oracle, monkeypatch):
    rng = np.random.default_rng(21)
    n, m = 500, 1600
    ea = rng.integers(0, n, m).astype(np.uint64)
    eb = rng.integers(0, n, m).astype(np.uint64)
    ej = rng.normal(size=m)
    biases = rng.normal(size=n) * 0.3
    betas = np.linspace(0.1, 1.2, 9)
    out = []
    for disable in ("0", "1"):
        monkeypatch.setenv("ISINGMC_DISABLE_RESIDENT", disable)
        g = capi.Graph(ea, eb, ej, nvars=n, biases=biases)
        st = capi.States(g, SEEDS)
        eps = st.do_time_steps(9, betas, per_step_energies=True)
        st.set_betas([0.3, 0.6, 0.9])
        st.do_time_steps(6)
        out.append((st.states(), eps, st.energies()))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-12, atol=1e-9)


def _packed_case(capi, oracle, ea, eb, ej, nvars, R, T, beta=None, beta_replica=None, initial=None):
    seeds = capi.make_seeds(77, R)
    g = capi.Graph(ea, eb, ej, nvars=nvars, force_general=True)
    assert g.kind == capi.KIND_GENERAL
    st = capi.States(g, seeds, initial_state=initial)
    ref_states = None
    if initial is not None:
        ref_states = np.tile(np.asarray(initial, dtype=np.uint8), (32 * ((R + 31) // 32), 1))
    if beta_replica is not None:
        st.set_betas(beta_replica)
        eps = st.do_time_steps(T, per_step_energies=True)
        e_ref, s_ref, eps_ref = oracle.pk_run(ea, eb, ej, nvars, seeds, T, beta_replica=beta_replica, states=ref_states,
                                              per_step=True)
    else:
        betas = [beta] * T if np.ndim(beta) == 0 else beta
        eps = st.do_time_steps(T, beta, per_step_energies=True)
        e_ref, s_ref, eps_ref = oracle.pk_run(ea, eb, ej, nvars, seeds, T, betas=betas, states=ref_states, per_step=True)
    np.testing.assert_array_equal(st.states().astype(np.uint8), s_ref[:R])
    np.testing.assert_allclose(eps, eps_ref, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(st.energies(), e_ref, rtol=1e-12, atol=1e-9)
    mags = st.magnetisations()
    np.testing.assert_array_equal(mags, 2 * s_ref[:R].sum(axis=1).astype(np.int64) - nvars)
    return st, s_ref


def test_packed_general_path_bit_exact(capi, oracle, exact, monkeypatch):
    """Replica-packed general path (uniform |J|, degree <= 6) against oracle engine D."""
    monkeypatch.setenv("ISINGMC_FORCE_PACKED", "1")
    ea, eb, ej = exact.cubic_lattice_edges(8, -1.0)                          # BASELINE c5's shape, small
    _packed_case(capi, oracle, ea, eb, ej, 512, R=40, T=8, beta=0.2217)     # partial last group
    _packed_case(capi, oracle, ea, eb, ej, 512, R=64, T=5, beta=np.linspace(0.1, 0.6, 5))
    _packed_case(capi, oracle, ea, eb, ej, 512, R=3, T=6, beta=0.0)
    ea2, eb2, ej2 = exact.square_lattice_edges(10, 6, 0.7, np.random.default_rng(4))   # +-J, not 64-wide
    _packed_case(capi, oracle, ea2, eb2, ej2, 60, R=33, T=10, beta=0.9)
    rng = np.random.default_rng(8)                                           # irregular degrees, isolated sites
    n = 150
    pairs = set()
    deg = np.zeros(n, dtype=int)
    while len(pairs) < 260:
        a, b = rng.integers(0, n - 10, 2)
        if a != b and deg[a] < 6 and deg[b] < 6 and (min(a, b), max(a, b)) not in pairs:
            pairs.add((min(a, b), max(a, b))); deg[a] += 1; deg[b] += 1
    ea3 = np.array([p[0] for p in pairs], dtype=np.uint64)
    eb3 = np.array([p[1] for p in pairs], dtype=np.uint64)
    ej3 = rng.choice([-1.5, 1.5], len(pairs))
    _packed_case(capi, oracle, ea3, eb3, ej3, n, R=50, T=12, beta=0.4)
    _packed_case(capi, oracle, ea3, eb3, ej3, n, R=50, T=7, beta_replica=np.linspace(0.05, 1.5, 50))
    ini = (rng.integers(0, 2, n)).astype(np.uint8)
    st, s_ref = _packed_case(capi, oracle, ea3, eb3, ej3, n, R=35, T=4, beta=0.6, initial=ini)
    other = 1 - ini
    st.set_state(33, other)                                                  # one replica of the second group
    got = st.states().astype(np.uint8)
    assert np.array_equal(got[33], other) and np.array_equal(got[32], s_ref[32]) and np.array_equal(got[34], s_ref[34])
    # ClassicIsing.add_graph on a packed container (round 3): replica 35 joins the open group and takes over the chain bit 3 of
    # group 1 has been running since the group was created
    st.append(5)
    assert st.count == 36 and np.array_equal(st.states().astype(np.uint8)[35], s_ref[35])


def _circulant(n, offsets, J):
    """Site i bonded to i + d (mod n) for d in offsets; d == n/2 gives one bond per pair: every site has the same degree."""
    ea, eb = [], []
    for d in offsets:
        idx = np.arange(n if 2 * d != n else n // 2, dtype=np.uint64)
        ea.append(idx)
        eb.append((idx + d) % n)
    ea, eb = np.concatenate(ea), np.concatenate(eb)
    return ea, eb, np.full(len(ea), J)


@pytest.mark.parametrize("deg,n,offsets", [(3, 3000, (1, 1500)), (4, 2900, (1, 2)), (5, 3400, (1, 2, 1700)), (6, 3700, (1, 2, 3)),
                                           (6, 2048, (1, 5, 11))])
@pytest.mark.parametrize("J", [-1.0, 0.6, "glass"])
def test_packed_uniform_degree_kernels_bit_exact(capi, oracle, monkeypatch, deg, n, offsets, J):
    """One degree: pk_sweep_uni_kernel<D, UB, PMJ> (packed_uni_kernels.hpp) on the full 256-blocks of every colour class,
    the general packed kernel on the padded tails -- against oracle engine D, with one coupling sign or random signs
    (PMJ), one beta for all replicas (UB), a beta schedule, and per-replica betas."""
    monkeypatch.setenv("ISINGMC_FORCE_PACKED", "1")
    ea, eb, ej = _circulant(n, offsets, 1.3 if J == "glass" else J)
    if J == "glass":
        ej = ej * np.random.default_rng(n).choice([-1.0, 1.0], len(ej))
    g = capi.Graph(ea, eb, ej, nvars=n)
    assert g.kind == capi.KIND_GENERAL and g.info.packed_degree == deg
    _packed_case(capi, oracle, ea, eb, ej, n, R=40, T=6, beta=0.35)
    _packed_case(capi, oracle, ea, eb, ej, n, R=33, T=5, beta=np.array([0.0, 0.1, 0.25, 0.6, 2.0]))
    _packed_case(capi, oracle, ea, eb, ej, n, R=64, T=5, beta_replica=np.linspace(-0.2, 1.4, 64))
    out = []                                                          # the same binary with the general kernel: equal
    for disable in ("0", "1"):
        monkeypatch.setenv("ISINGMC_DISABLE_PACKED_UNIFORM", disable)
        st = capi.States(g, capi.make_seeds(3, 96))
        st.do_time_steps(7, 0.45)
        out.append(st.states().copy())
    monkeypatch.delenv("ISINGMC_DISABLE_PACKED_UNIFORM")
    np.testing.assert_array_equal(out[0], out[1])
    # a site of another degree: the general kernel (packed_degree == 0), same oracle
    assert capi.Graph(ea[1:], eb[1:], ej[1:], nvars=n).info.packed_degree == 0
    _packed_case(capi, oracle, ea[1:], eb[1:], ej[1:], n, R=20, T=3, beta=0.35)


def test_packed_path_equilibrium_vs_kaufman(capi, exact, monkeypatch):
    """K3 for the packed path: 32x32 torus (general path: not 64-wide), 64 replicas."""
    monkeypatch.setenv("ISINGMC_FORCE_PACKED", "1")
    L, beta, R = 32, 0.35, 64
    ea, eb, ej = exact.square_lattice_edges(L, L, -1.0)
    g = capi.Graph(ea, eb, ej)
    st = capi.States(g, capi.make_seeds(5, R))
    st.do_time_steps(400, beta)
    per_replica = st.do_time_steps(1500, beta, per_step_energies=True).mean(axis=1)
    mean, err = per_replica.mean(), per_replica.std(ddof=1) / np.sqrt(R)
    ref = exact.kaufman_energy(L, L, beta)
    assert abs(mean - ref) < 4.5 * err, (mean, ref, err)


def test_packed_path_is_selected_for_large_uniform_graphs(capi, oracle, exact, monkeypatch):
    """No env override: >= 16 experiments on a uniform-|J| graph too big for the LDS-resident kernel take
    the replica-packed path (600x600 torus: not 64-wide, 360 000 sites), and match oracle engine D."""
    W = H = 600
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    seeds = capi.make_seeds(9, 32)
    g = capi.Graph(ea, eb, ej)
    assert g.kind == capi.KIND_GENERAL
    st = capi.States(g, seeds)
    st.do_time_steps(3, 0.44)
    e_ref, s_ref = oracle.pk_run(ea, eb, ej, W * H, seeds, 3, betas=[0.44] * 3)
    np.testing.assert_array_equal(st.states().astype(np.uint8), s_ref[:32])
    np.testing.assert_array_equal(st.energies(), e_ref)
    # Round 3: on a graph of >= 8 000 sites the packed kernels are ahead from ONE experiment on (a mostly empty 32-replica word
    # still beats the per-replica CSR launches 1.4x, profiles/r03_few_replicas.txt): two experiments = oracle engine D too
    st2 = capi.States(g, seeds[:2])
    st2.do_time_steps(2, 0.44)
    _, s_d = oracle.pk_run(ea, eb, ej, W * H, seeds[:2], 2, betas=[0.44] * 2)
    np.testing.assert_array_equal(st2.states().astype(np.uint8), s_d[:2])
    # the thread-per-site path (oracle engine C) stays selectable
    monkeypatch.setenv("ISINGMC_DISABLE_PACKED", "1")
    st3 = capi.States(g, seeds[:2])
    st3.do_time_steps(2, 0.44)
    for r in range(2):
        _, s_c = oracle.gen_run(ea, eb, ej, W * H, seeds[r], [0.44] * 2)
        np.testing.assert_array_equal(st3.states()[r].astype(np.uint8), s_c)


@pytest.mark.parametrize("glass", [False, True])
def test_large_launch_takes_the_looping_kernel_bit_exact(capi, oracle, exact, glass):
    """512 x 512 x 2048 replicas = 2 x 2048 workgroups of quad pairs per launch: the host picks
    lat_sweep_loop_kernel (two quads per thread; uniform J and +-J instantiations).  Replicas are independent,
    so the first, two middle and the last one against the oracle pin the whole launch."""
    W, H, R, T, beta = 512, 512, 2048, 3, 0.4407
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0, np.random.default_rng(5) if glass else None)
    g = capi.Graph(ea, eb, ej)
    assert g.kind == capi.KIND_LATTICE2D and bool(g.info.uniform_sign) != glass
    seeds = capi.make_seeds(99, R)
    st = capi.States(g, seeds)
    st.do_time_steps(T, beta)
    packed = st.packed()
    energies = st.energies()
    if glass:
        lat = oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8))
    else:
        lat = oracle.Lat(W, H, 1.0, 0)
    for r in (0, 777, 1024, R - 1):
        ref = lat.init(seeds[r])
        for t in range(T):
            lat.sweep(ref, seeds[r], t, beta)
        np.testing.assert_array_equal(packed[r], ref, err_msg=f"replica {r}")
        assert energies[r] == lat.energy_mag(ref)[0]


# ---- round 2: the three lattice-kernel branches that had no oracle comparison ----------

@pytest.mark.parametrize("strip", ["0", "1"])
def test_per_step_energies_many_workgroups_per_replica(capi, oracle, exact, monkeypatch, strip):
    """lat_sweep_measure_kernel with 32 workgroups per replica (4096 x 512: 8192 quads per colour): the 16 counter
    slots per replica are each hit twice and summed on the host -- energies after every timestep
    (lattice.rs:445-455) and the final configuration against the oracle, +-J couplings."""
    monkeypatch.setenv("ISINGMC_STRIP", strip)     # "0": the streaming kernels; "1": the persistent strip kernel (32 strips)
    W, H, T, beta = 4096, 512, 3, 0.7
    ea, eb, ej = exact.square_lattice_edges(W, H, 1.0, np.random.default_rng(12))
    g = capi.Graph(ea, eb, ej)
    assert g.kind == capi.KIND_LATTICE2D and not g.info.uniform_sign
    seeds = SEEDS[:2]
    st = capi.States(g, seeds)
    eps = st.do_time_steps(T, beta, per_step_energies=True)
    lat = oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8))
    packed = st.packed()
    for r, s in enumerate(seeds):
        ref = lat.init(s)
        for t in range(T):
            lat.sweep(ref, s, t, beta)
            assert eps[r, t] == lat.energy_mag(ref)[0], (r, t)
        np.testing.assert_array_equal(packed[r], ref)


def test_per_step_energies_2048_square_vs_separate_measurement(capi, oracle, exact, monkeypatch):
    """BASELINE c4's geometry (2048^2: 64 workgroups per replica, every counter slot hit 4x): the fused per-step
    energies equal lat_measure_kernel's after single steps, and K1 (host recomputation from the returned spins)."""
    monkeypatch.setenv("ISINGMC_STRIP", "0")       # this test is about lat_sweep_measure_kernel's counter slots
    L, T = 2048, 3
    ea, eb, ej = exact.square_lattice_edges(L, L, -1.0)
    g = capi.Graph(ea, eb, ej)
    seeds = SEEDS[:2]
    betas = np.array([0.3, 0.4407, 0.6])
    a = capi.States(g, seeds)
    eps = a.do_time_steps(T, betas, per_step_energies=True)
    b = capi.States(g, seeds)
    for t in range(T):
        b.do_time_steps(1, float(betas[t]))
        np.testing.assert_array_equal(eps[:, t], b.energies(), err_msg=f"step {t}")
    np.testing.assert_array_equal(a.packed(), b.packed())
    spins = a.states()
    for r in range(2):
        assert eps[r, T - 1] == oracle.energy(ea, eb, ej, L * L, spins[r])


def test_streaming_kernel_rows_of_16384_bit_exact(capi, oracle, exact):
    """W = 16384: 64 quads per row, a wavefront never leaves its row (the cl >= 6 branch of load_quad_uni);
    H = 64 gives 4096 quads per colour, too many for the LDS-resident kernel."""
    st = _lattice_case(capi, oracle, exact, 16384, 64, -1.0, 0.4407, T=3, per_step=False)
    assert st.graph.info.width == 16384
    _lattice_case(capi, oracle, exact, 16384, 64, 1.0, 0.8, T=2, rng=np.random.default_rng(3), per_step=True)


def test_looping_kernel_at_headline_width_bit_exact(capi, oracle, exact):
    """lat_sweep_loop_kernel<uniform J> at the headline geometry W = 4096 (cols_log2 = 4): 4096 x 128 x 2048
    replicas = 4 workgroups of quad pairs per replica (2048 quads per colour: not LDS-resident), enough
    workgroups for the host to choose the looping kernel with or without replica lanes."""
    W, H, R, T, beta = 4096, 128, 2048, 3, 0.4407
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    g = capi.Graph(ea, eb, ej)
    assert g.kind == capi.KIND_LATTICE2D and g.info.uniform_sign
    seeds = capi.make_seeds(4096, R)
    st = capi.States(g, seeds)
    st.do_time_steps(T, beta)
    packed = st.packed()
    energies = st.energies()
    lat = oracle.Lat(W, H, 1.0, 0)
    for r in (0, 1023, 1024, R - 1):
        ref = lat.init(seeds[r])
        for t in range(T):
            lat.sweep(ref, seeds[r], t, beta)
        np.testing.assert_array_equal(packed[r], ref, err_msg=f"replica {r}")
        assert energies[r] == lat.energy_mag(ref)[0]


@pytest.mark.parametrize("kind", ["everything staged", "topology staged", "nothing staged"])
def test_resident_csr_kernel_with_the_graph_in_lds(capi, oracle, exact, kind):
    """gen_resident_kernel keeps the graph in LDS when it fits (everything: 24 x 20; the topology alone: 64 x 64 with f64
    couplings; nothing: 19^3) -- the same arithmetic in the same order, so all three equal oracle engine C bit for bit."""
    rng = np.random.default_rng(19)
    if kind == "everything staged":
        ea, eb, _ = exact.square_lattice_edges(24, 20, 1.0)
    elif kind == "topology staged":
        ea, eb, _ = exact.square_lattice_edges(64, 64, 1.0)
    else:
        ea, eb, _ = exact.cubic_lattice_edges(19, 1.0)
    n = int(max(ea.max(), eb.max())) + 1
    ej = rng.normal(size=len(ea))
    biases = rng.normal(size=n) * 0.3
    g = capi.Graph(ea, eb, ej, nvars=n, biases=biases, force_general=True)
    seeds = capi.make_seeds(23, 3)
    st = capi.States(g, seeds)
    betas = np.array([0.3, 0.7, 1.4])
    eps = st.do_time_steps(3, betas, per_step_energies=True)
    spins = st.states().astype(np.uint8)
    for r in range(3):
        e_ref, s_ref, eps_ref = oracle.gen_run(ea, eb, ej, n, seeds[r], betas, biases=biases, per_step=True)
        assert np.array_equal(spins[r], s_ref), (kind, r)
        np.testing.assert_allclose(eps[r], eps_ref, rtol=1e-12)
