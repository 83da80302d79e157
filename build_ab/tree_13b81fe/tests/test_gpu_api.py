"""GPU tests of the drop-in Python API (py_monte_carlo) and of the full-size properties.

Small cases are checked against the oracle bit for bit; BASELINE.json's full sizes through
size-independent properties (energy recomputation from the returned states, exact limits,
shard invariance) and against exact physics within Monte-Carlo error.
"""
import math
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _edges(ea, eb, ej):
    return [((int(a), int(b)), float(j)) for a, b, j in zip(ea, eb, ej)]


@pytest.fixture(scope="module")
def mod():
    import py_monte_carlo
    return py_monte_carlo


def test_run_monte_carlo_c1_shape_and_oracle(mod, oracle, exact):
    """BASELINE config c1: 16x16, beta=0.3, num_experiments=4 (general path on the GPU)."""
    ea, eb, ej = exact.square_lattice_edges(16, 16, -1.0)
    lat = mod.Lattice(_edges(ea, eb, ej), seed_gen=1234)
    assert lat.engine_info()["kind"] == "general"
    energies, states = lat.run_monte_carlo(0.3, 1000, 4)
    assert energies.shape == (4,) and energies.dtype == np.float64
    assert states.shape == (4, 256) and states.dtype == np.bool_ and states.flags.c_contiguous
    for r, s in enumerate(lat.make_seeds(4)):
        e_ref, s_ref = oracle.gen_run(ea, eb, ej, 256, s, [0.3] * 1000)
        assert np.array_equal(states[r], s_ref.astype(bool)) and energies[r] == e_ref
    e2, s2 = lat.run_monte_carlo(0.3, 1000, 4)       # seed_gen set: reruns reuse the seeds (lattice.rs:76-80)
    assert np.array_equal(e2, energies) and np.array_equal(s2, states)


def test_readme_example_runs(mod, oracle):
    lat = mod.Lattice([((0, 1), 1.0), ((1, 2), -1.0)])
    e, s = lat.run_monte_carlo(1.0, 100, 8)
    ea, eb, ej = oracle.split_edges([((0, 1), 1.0), ((1, 2), -1.0)])
    for r in range(8):
        assert e[r] == oracle.energy(ea, eb, ej, 3, s[r])
    assert set(np.unique(e)) <= {-2.0, 0.0, 2.0}


def test_sampling_semantics(mod, oracle, exact):
    """lattice.rs:244-250, 271-287: S = T // freq samples, each after `freq` more steps, after thermalisation."""
    W, H = 64, 8
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    lat = mod.Lattice.from_arrays(ea, eb, ej, seed_gen=5)
    assert lat.engine_info()["kind"] == "lattice2d"
    energies, states = lat.run_monte_carlo_sampling(0.5, 10, 3, None, 4, 3)   # therm=4, freq=3 -> S=3
    assert energies.shape == (3, 3) and states.shape == (3, 3, W * H) and states.dtype == np.bool_
    olat = oracle.Lat(W, H)
    for r, seed in enumerate(lat.make_seeds(3)):
        st = olat.init(seed)
        t = 0
        for _ in range(4):
            olat.sweep(st, seed, t, 0.5); t += 1
        for k in range(3):
            for _ in range(3):
                olat.sweep(st, seed, t, 0.5); t += 1
            assert np.array_equal(states[r, k], olat.unpack(st).astype(bool))
            assert energies[r, k] == olat.energy_mag(st)[0]
    e1, s1 = lat.run_monte_carlo_sampling(0.5, 7, 2)                        # defaults: therm 0, freq 1
    assert e1.shape == (2, 7) and s1.shape == (2, 7, W * H)
    with pytest.raises(ValueError):
        lat.run_monte_carlo_sampling(0.5, 7, 2, sampling_freq=0)


def test_annealing_and_energies(mod, oracle, exact):
    W, H = 128, 8
    ea, eb, ej = exact.square_lattice_edges(W, H, 1.0, np.random.default_rng(2024))
    lat = mod.Lattice.from_arrays(ea, eb, ej, seed_gen=9)
    T = 12
    stops = [(0, 0.1), (6, 1.0), (12, 3.0)]
    e_all, s_fin = lat.run_monte_carlo_annealing_and_get_energies(stops, T, 2)
    e_fin, s_fin2 = lat.run_monte_carlo_annealing(stops, T, 2)
    assert e_all.shape == (2, T) and s_fin.shape == (2, W * H)
    assert np.array_equal(s_fin, s_fin2) and np.array_equal(e_all[:, -1], e_fin)
    from pyisingmontecarlo_amd import _capi
    betas = _capi.expand_schedule(stops, T)
    olat = oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8))
    for r, seed in enumerate(lat.make_seeds(2)):
        st = olat.init(seed)
        for t in range(T):
            olat.sweep(st, seed, t, betas[t])
            assert e_all[r, t] == olat.energy_mag(st)[0]
        assert np.array_equal(s_fin[r], olat.unpack(st).astype(bool))
    e_def, _ = lat.run_monte_carlo_annealing([], T, 2)                      # default schedule beta = 1 (lattice.rs:321-324)
    e_one, _ = lat.run_monte_carlo(1.0, T, 2)
    assert np.array_equal(e_def, e_one)


def test_annealing_compat_mode_is_constant_beta(mod, exact, monkeypatch):
    ea, eb, ej = exact.square_lattice_edges(64, 8, -1.0)
    lat = mod.Lattice.from_arrays(ea, eb, ej, seed_gen=3)
    monkeypatch.setenv("ISINGMC_COMPAT_ANNEAL_BUG", "1")
    e_bug, s_bug = lat.run_monte_carlo_annealing([(0, 0.1), (10, 0.7)], 10, 2)
    monkeypatch.delenv("ISINGMC_COMPAT_ANNEAL_BUG")
    e_const, s_const = lat.run_monte_carlo(0.7, 10, 2)
    assert np.array_equal(s_bug, s_const) and np.array_equal(e_bug, e_const)


def test_initial_state_and_bias_paths(mod, oracle, exact):
    ea, eb, ej = exact.square_lattice_edges(64, 8, -1.0)
    lat = mod.Lattice.from_arrays(ea, eb, ej, seed_gen=2)
    ini = (np.arange(512) % 2 == 0)
    lat.set_initial_state(ini.tolist())
    e, s = lat.run_monte_carlo(0.4, 0, 2)                                   # zero timesteps: the initial state comes back
    assert np.array_equal(s[0], ini) and np.array_equal(s[1], ini)
    assert e[0] == oracle.energy(ea, eb, ej, 512, ini.astype(np.uint8))
    lat.set_initial_state([])
    lat.set_global_bias(0.25)                                               # a field -> general path
    assert lat.engine_info()["kind"] == "general"
    e, s = lat.run_monte_carlo(0.4, 20, 2)
    for r, seed in enumerate(lat.make_seeds(2)):
        e_ref, s_ref = oracle.gen_run(ea, eb, ej, 512, seed, [0.4] * 20, biases=np.full(512, 0.25))
        assert np.array_equal(s[r], s_ref.astype(bool)) and abs(e[r] - e_ref) < 1e-9
    lat.set_individual_bias(7, -3.0)
    e, s = lat.run_monte_carlo(0.4, 5, 1)
    b = np.full(512, 0.25); b[7] = -3.0
    e_ref, s_ref = oracle.gen_run(ea, eb, ej, 512, lat.make_seeds(1)[0], [0.4] * 5, biases=b)
    assert np.array_equal(s[0], s_ref.astype(bool))


def test_replica_range_is_shard_invariant(mod, exact):
    """K8: keyed by global experiment index, any sharding of the experiments gives the same arrays."""
    ea, eb, ej = exact.square_lattice_edges(256, 16, -1.0)
    lat = mod.Lattice.from_arrays(ea, eb, ej, seed_gen=1)
    e, s = lat.run_monte_carlo(0.4407, 9, 8)
    parts = [lat.run_monte_carlo(0.4407, 9, 8, replica_range=r) for r in [(0, 3), (3, 4), (4, 8)]]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), e)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), s)
    with pytest.raises(ValueError):
        lat.run_monte_carlo(0.4407, 1, 8, replica_range=(5, 9))


def test_replica_range_is_shard_invariant_on_packed_general_graphs(mod, capi, oracle, exact):
    """the replica-packed general path groups 32 replicas (shared Philox words, ties numbered over the
    group).  Group, key and bit follow the GLOBAL experiment index, so 64 experiments cut 4 x 16, 8 x 8 or
    unevenly give the arrays of the unsharded call -- which match oracle engine D."""
    W = H = 120                                                    # 14 400 sites, not 64-wide: general, above the resident bound
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    lat = mod.Lattice.from_arrays(ea, eb, ej, seed_gen=3)
    assert lat.engine_info()["kind"] == "general"
    R, T, beta = 64, 4, 0.42
    e, s = lat.run_monte_carlo(beta, T, R)
    e_ref, s_ref = oracle.pk_run(ea, eb, ej, W * H, np.array(lat.make_seeds(R), dtype=np.uint64), T, betas=[beta] * T)
    assert np.array_equal(s, s_ref[:R].astype(bool)) and np.array_equal(e, e_ref)
    for cuts in ([(16 * k, 16 * k + 16) for k in range(4)], [(8 * k, 8 * k + 8) for k in range(8)],
                 [(0, 5), (5, 37), (37, 64)]):
        parts = [lat.run_monte_carlo(beta, T, R, replica_range=r) for r in cuts]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), e), cuts
        assert np.array_equal(np.concatenate([p[1] for p in parts]), s), cuts
    # per-step energies and sampling through a shard that owns bits 5..36 of two groups
    betas = [(0, 0.2), (T, 0.6)]
    ea_full, _ = lat.run_monte_carlo_annealing_and_get_energies(betas, T, R)
    ea_part, _ = lat.run_monte_carlo_annealing_and_get_energies(betas, T, R, replica_range=(5, 37))
    assert np.array_equal(ea_part, ea_full[5:37])
    es_full, ss_full = lat.run_monte_carlo_sampling(beta, 4, R, None, 1, 2)
    es_part, ss_part = lat.run_monte_carlo_sampling(beta, 4, R, None, 1, 2, replica_range=(5, 37))
    assert np.array_equal(es_part, es_full[5:37]) and np.array_equal(ss_part, ss_full[5:37])
    # per-replica betas need whole groups
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(3, R)
    st = capi.States(g, seeds, replica_range=(5, 37))
    with pytest.raises(ValueError, match="multiples of 32"):
        st.set_betas(np.linspace(0.2, 0.3, 32))
    # a shard that ENDS inside a group (and not at the last experiment) is refused too -- the foreign bits of the
    # group would otherwise run at betas this shard does not know
    with pytest.raises(ValueError, match="multiples of 32"):
        capi.States(g, seeds, replica_range=(0, 40)).set_betas(np.linspace(0.2, 0.3, 40))
    tail = capi.States(g, capi.make_seeds(3, 40), replica_range=(32, 40))      # ends at the last experiment: fine
    tail.set_betas(np.linspace(0.2, 0.3, 8))
    whole = capi.States(g, seeds)
    whole.set_betas(np.linspace(0.1, 1.0, R))
    whole.do_time_steps(3)
    half = capi.States(g, seeds, replica_range=(32, 64))
    half.set_betas(np.linspace(0.1, 1.0, R)[32:])
    half.do_time_steps(3)
    assert np.array_equal(half.states(), whole.states()[32:]) and np.array_equal(half.energies(), whole.energies()[32:])
    # fewer than 16 experiments in the shard, 64 in total: still the packed trajectories, not the per-replica CSR ones
    few = capi.States(g, seeds, replica_range=(60, 64))
    few.do_time_steps(T, beta)
    assert np.array_equal(few.states(), s[60:64])


@pytest.mark.parametrize("kind", ["lattice", "packed_general", "real_coupling"])
def test_in_process_device_fan_out_equals_single_device(mod, exact, monkeypatch, kind):
    """The rayon fan-out of lattice.rs:192-197 over the device list (ISINGMC_DEVICES / set_devices: one host
    thread + one isingmc_states per entry).  The list 0,0 runs two blocks side by side on the one GPU here;
    all four run_* methods must return the arrays of the single-device call."""
    if kind == "lattice":
        ea, eb, ej = exact.square_lattice_edges(256, 64, -1.0, np.random.default_rng(1))
    else:
        ea, eb, ej = exact.square_lattice_edges(120, 120, -1.0)
        if kind == "real_coupling":                                # Gaussian couplings: the replica-packed real-coupling path
            ej = np.random.default_rng(2).normal(size=len(ea))
    one = mod.Lattice.from_arrays(ea, eb, ej, seed_gen=11)
    assert one.get_devices() == [0]
    if kind == "real_coupling":
        assert one.engine_info()["real_slots"] == 4
    monkeypatch.setenv("ISINGMC_DEVICES", "0,0,0")
    many = mod.Lattice.from_arrays(ea, eb, ej, seed_gen=11)
    assert many.get_devices() == [0, 0, 0]
    R, T = 70, 6                                                   # blocks 32 + 32 + 6
    stops = [(0, 0.2), (T, 0.7)]
    for call in (lambda l: l.run_monte_carlo(0.44, T, R),
                 lambda l: l.run_monte_carlo_annealing(stops, T, R),
                 lambda l: l.run_monte_carlo_annealing_and_get_energies(stops, T, R),
                 lambda l: l.run_monte_carlo_sampling(0.44, T, R, None, 2, 3),
                 lambda l: l.run_monte_carlo(0.44, T, R, replica_range=(9, 50)),
                 lambda l: l.run_monte_carlo(0.44, T, 2),          # fewer experiments than devices
                 lambda l: l.run_monte_carlo(0.44, T, 0)):
        a, b = call(one), call(many)
        assert a[0].shape == b[0].shape and a[1].shape == b[1].shape
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    many.set_devices([0, 99])                                      # a block that fails reports, nothing hangs
    with pytest.raises(RuntimeError):
        many.run_monte_carlo(0.44, 2, 8)
    with pytest.raises(ValueError):
        many.set_devices([])


def test_classic_ising_persistent(mod, oracle, exact):
    W, H = 64, 8
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    ci = mod.ClassicIsing(_edges(ea, eb, ej), None, 2, 42)
    assert ci.get_num_graphs() == 2
    ci.add_graph([True] * (W * H))
    assert ci.run_monte_carlo(0.5, 4) is None                               # classicising.rs:88-110 returns nothing
    ci.run_monte_carlo(0.5, 3, nspinupdates=2 * W * H)                      # 2 sweeps per timestep
    seeds = oracle.make_seeds(42, 3)                                        # the container rng's draws (classicising.rs:67)
    olat = oracle.Lat(W, H)
    states, energies = ci.get_states(), ci.get_energies()
    for r in range(3):
        st = olat.init(seeds[r]) if r < 2 else olat.pack(np.ones(W * H, dtype=np.uint8))
        for t in range(10):
            olat.sweep(st, seeds[r], t, 0.5)
        assert np.array_equal(states[r], olat.unpack(st).astype(bool)) and energies[r] == olat.energy_mag(st)[0]
    e, s = ci.run_monte_carlo_sampling(0.5, 6, None, None, None, None, 2, 2)
    assert e.shape == (3, 3) and s.shape == (3, 3, W * H)
    with pytest.raises(ValueError):
        mod.ClassicIsing([])
    with pytest.raises(ValueError, match="positive"):
        ci.run_monte_carlo(0.5, 1, nspinupdates=0)
    field = mod.ClassicIsing(_edges(ea, eb, ej), 0.5, 1, 1)                 # longitudinal field -> general path
    field.run_monte_carlo(0.3, 5)
    e_ref, s_ref = oracle.gen_run(ea, eb, ej, W * H, oracle.make_seeds(1, 1)[0], [0.3] * 5, biases=np.full(W * H, 0.5))
    assert np.array_equal(field.get_states()[0], s_ref.astype(bool))


def test_classic_ising_any_nspinupdates(mod, oracle, exact):
    """classicising.rs:88-110 takes any count of single-spin attempts per timestep.  Attempts are executed sweep by sweep:
    timesteps x nspinupdates attempts accumulate (across calls), every nvars of them run as one sweep, the rest stays pending."""
    W, H = 64, 8
    N = W * H
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    ci = mod.ClassicIsing(_edges(ea, eb, ej), None, 2, 7)
    seeds = oracle.make_seeds(7, 2)
    olat = oracle.Lat(W, H)
    ref = [olat.init(s) for s in seeds]
    done = [0]

    def advance(sweeps):
        for r in range(2):
            for t in range(done[0], done[0] + sweeps):
                olat.sweep(ref[r], seeds[r], t, 0.5)
        done[0] += sweeps

    def check():
        states, energies = ci.get_states(), ci.get_energies()
        for r in range(2):
            assert np.array_equal(states[r], olat.unpack(ref[r]).astype(bool)) and energies[r] == olat.energy_mag(ref[r])[0]

    with pytest.warns(UserWarning, match="sweep by sweep"):
        ci.run_monte_carlo(0.5, 3, nspinupdates=100)        # 300 attempts < 512: nothing runs yet
    check()
    ci.run_monte_carlo(0.5, 3, nspinupdates=100)            # 600 attempts: one sweep, 88 pending
    advance(1)
    check()
    ci.run_monte_carlo(0.5, 5, nspinupdates=N + N // 2)     # 88 + 3840 attempts: 7 sweeps, 344 pending
    advance(7)
    check()
    # sampling with attempts that are not whole sweeps: blocks of varying length (pending 344; 2 x 300 per block)
    e, s = ci.run_monte_carlo_sampling(0.5, 6, 300, None, None, None, 1, 2)
    assert e.shape == (2, 3) and s.shape == (2, 3, N)
    pending = 344
    blocks = []
    for attempts in (300, 600, 600, 600):                   # thermalisation (1 timestep), then 3 blocks of 2 timesteps
        pending += attempts
        blocks.append(pending // N)
        pending %= N
    advance(blocks[0])
    for k in range(3):
        advance(blocks[k + 1])
        for r in range(2):
            assert np.array_equal(s[r, k], olat.unpack(ref[r]).astype(bool)) and e[r, k] == olat.energy_mag(ref[r])[0]
    check()


def _blocked(x, n=16):
    m = np.array([b.mean() for b in np.array_split(np.asarray(x, dtype=np.float64), n)])
    return m.mean(), m.std(ddof=1) / math.sqrt(n)


def test_equilibrium_energy_vs_kaufman(capi, exact):
    """K3 on the GPU: <E> against the exact finite-torus value, error from independent replicas.
    256x256 away from T_c; at beta_c a 64x64 torus (critical slowing down: tau ~ L^2.17 sweeps)."""
    for L, beta, therm, steps, R in ((256, 0.3, 300, 1500, 32), (256, 0.6, 500, 1500, 32),
                                     (64, 0.4407, 20000, 40000, 64)):
        ea, eb, ej = exact.square_lattice_edges(L, L, -1.0)
        g = capi.Graph(ea, eb, ej)
        st = capi.States(g, capi.make_seeds(17, R), initial_state=np.ones(L * L, dtype=np.uint8))
        st.do_time_steps(therm, beta)
        per_replica = st.do_time_steps(steps, beta, per_step_energies=True).mean(axis=1)
        mean, err = per_replica.mean(), per_replica.std(ddof=1) / math.sqrt(R)
        ref = exact.kaufman_energy(L, L, beta)
        assert abs(mean - ref) < 3.0 * err, (L, beta, mean, ref, err)   # within 3 sigma
        assert err < 3e-3 * abs(ref)


def test_k2_general_path_vs_exact_enumeration(capi, exact):
    """K2 on the GPU: the f64 CSR path on a 16-spin random +-J graph with fields against the exact Boltzmann
    averages (brute-force enumeration): <E> and <|M|> within 3 sigma, sigma from 512 independent replicas."""
    rng = np.random.default_rng(16)
    pairs = set()
    while len(pairs) < 28:
        a, b = (int(v) for v in rng.integers(0, 16, 2))
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    pairs = sorted(pairs)
    ea = np.array([p[0] for p in pairs], dtype=np.uint64)
    eb = np.array([p[1] for p in pairs], dtype=np.uint64)
    ej = rng.choice([-1.0, 1.0], size=len(pairs))
    h = rng.choice([-0.5, 0.0, 0.5], size=16)
    beta, R = 0.5, 512
    ex = exact.enumerate_graph(ea, eb, ej, 16, beta, h)
    g = capi.Graph(ea, eb, ej, nvars=16, biases=h)
    assert g.kind == capi.KIND_GENERAL
    st = capi.States(g, capi.make_seeds(41, R))
    st.do_time_steps(200, beta)
    e = st.do_time_steps(4000, beta, per_step_energies=True).mean(axis=1)
    mags = []
    for _ in range(800):
        st.do_time_steps(5, beta)
        mags.append(np.abs(st.magnetisations()))
    m = np.mean(mags, axis=0)
    for got, want, name in ((e, ex["E"], "E"), (m, ex["absM"], "|M|")):
        z = (got.mean() - want) / (got.std(ddof=1) / math.sqrt(R))
        assert abs(z) < 3.0, (name, z, got.mean(), want)


def test_observables_vs_reference_faithful_cpu_engine(capi, oracle, exact):
    """K6: <E> and <|M|> of the checkerboard kernel against the random-site sequential CPU engine
    (oracle engine A, the restatement of the reference's algorithm): two-sample z-tests."""
    W = H = 64
    beta = 0.42
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    g = capi.Graph(ea, eb, ej)
    st = capi.States(g, capi.make_seeds(3, 64))
    st.do_time_steps(3000, beta)
    ms, es = [], []
    for _ in range(80):
        st.do_time_steps(50, beta)
        ms.append(np.abs(st.magnetisations()))
        es.append(st.energies())
    gm, ge = np.mean(ms, axis=0), np.mean(es, axis=0)                       # one time average per replica
    _, _, eps = oracle.ref_run(ea, eb, ej, W * H, oracle.make_seeds(8, 16), [beta] * 4000, per_step=True)
    ce = eps[:, 2000:].mean(axis=1)
    z = (ge.mean() - ce.mean()) / math.sqrt(ge.var(ddof=1) / len(ge) + ce.var(ddof=1) / len(ce))
    assert abs(z) < 3.0, ("energy", z)
    _, finals = oracle.ref_run(ea, eb, ej, W * H, oracle.make_seeds(9, 64), [beta] * 3000)
    cpu_absm = np.abs(2.0 * finals.sum(axis=1) - W * H)
    gpu_absm = np.abs(st.magnetisations()).astype(np.float64)              # end-of-chain samples on both sides
    z = (gpu_absm.mean() - cpu_absm.mean()) / math.sqrt(gpu_absm.var(ddof=1) / 64 + cpu_absm.var(ddof=1) / 64)
    assert abs(z) < 3.0, ("|M|", z)
    assert abs(gm.mean() - cpu_absm.mean()) < 0.15 * W * H


def test_full_size_properties_4096(capi, exact):
    """BASELINE c2's lattice (4096^2), few replicas: properties that do not need the oracle."""
    L = 4096
    ea, eb, ej = exact.square_lattice_edges(L, L, -1.0)
    g = capi.Graph(ea, eb, ej)
    assert g.kind == capi.KIND_LATTICE2D
    seeds = capi.make_seeds(1, 3)
    st = capi.States(g, seeds)
    before = st.states()
    m0 = st.magnetisations()
    assert np.all(np.abs(m0) < 6 * L)                                       # random start: |M| ~ sqrt(N)
    st.do_time_steps(1, 0.0)                                                # beta = 0: every attempt accepted (K4)
    assert np.array_equal(st.states(), ~before)
    assert np.array_equal(st.magnetisations(), -m0)
    st.do_time_steps(30, 0.4407)
    spins = st.states()
    e = st.energies()
    s2 = spins.reshape(3, L, L).astype(np.int8) * 2 - 1                     # K1: recompute sum J s s on the host
    e_host = -(s2 * np.roll(s2, -1, axis=2)).sum(axis=(1, 2), dtype=np.int64) - (s2 * np.roll(s2, -1, axis=1)).sum(axis=(1, 2), dtype=np.int64)
    assert np.array_equal(e, e_host.astype(np.float64))
    assert np.array_equal(st.magnetisations(), s2.sum(axis=(1, 2), dtype=np.int64))
    assert np.all(e / L ** 2 < -1.2)                                         # relaxing towards -sqrt(2)
    cold = capi.States(g, seeds[:1], initial_state=np.ones(L * L, dtype=np.uint8))
    cold.do_time_steps(2, 20.0)                                             # ordered and cold: nothing moves (K4)
    assert cold.energies()[0] == -2.0 * L * L and cold.magnetisations()[0] == L * L
    # shard invariance at full size: replica 2 alone == replica 2 of the batch
    solo = capi.States(g, seeds[2:3])
    solo.do_time_steps(1, 0.0)
    solo.do_time_steps(30, 0.4407)
    assert np.array_equal(solo.packed()[0], st.packed()[2])


def test_tempering_matches_oracle_engine(capi, exact):
    """The classical ladder on the GPU against the same host logic driven by the CPU oracle engine."""
    from helpers import OracleLatEngine
    from pyisingmontecarlo_amd.tempering import ClassicalTempering
    W, H = 64, 8
    edges = exact.square_lattice_edges(W, H, -1.0)
    runs = []
    for factory in (None, lambda: OracleLatEngine(W, H)):
        pt = ClassicalTempering(edges, seed=5, engine_factory=factory)
        for b in np.linspace(0.38, 0.5, 7):
            pt.add_graph(b)
        pt.timesteps(5)
        states, energies = pt.timesteps_sample(24, replica_swap_freq=3, sampling_freq=6)
        runs.append((states, energies, pt.get_permutation(), pt.get_total_swaps()))
    assert runs[0][3] == runs[1][3] > 0                                     # K7: swaps happen, identically
    assert np.array_equal(runs[0][2], runs[1][2])
    assert np.array_equal(runs[0][1], runs[1][1])
    assert np.array_equal(runs[0][0], runs[1][0])
    assert runs[0][0].shape == (7, 4, W * H)


def test_on_stream_tempering_equals_host_swap_path(capi, exact):
    """timesteps(t, replica_swap_freq) keeps sweeps, measurement, exchange decisions and relabelling on the
    HIP stream; it must reproduce the host-side swap step (oracle engine) decision for decision."""
    from helpers import OracleLatEngine
    from pyisingmontecarlo_amd.tempering import ClassicalTempering
    W, H = 64, 8
    edges = exact.square_lattice_edges(W, H, -1.0)
    runs = []
    for factory in (None, lambda: OracleLatEngine(W, H)):
        pt = ClassicalTempering(edges, seed=11, engine_factory=factory)
        for b in np.linspace(0.36, 0.52, 9):
            pt.add_graph(b)
        pt.timesteps(4)
        pt.timesteps(33, replica_swap_freq=3)            # 11 exchange rounds
        states, energies = pt.timesteps_sample(6, replica_swap_freq=2, sampling_freq=3)
        runs.append((pt.get_permutation(), pt.get_total_swaps(), states, energies))
    assert runs[0][1] == runs[1][1] > 0
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)


def test_on_stream_tempering_two_shards_in_one_process(capi, exact):
    """The multi-GPU exchange protocol with both 'ranks' on this one GPU: two shard engines attach halves of
    the ladder; the all-gather between pt_measure and pt_swap is emulated by device copies issued on each
    engine's own stream (torch.cuda.ExternalStream, the object the RCCL collective is enqueued under).
    Result must equal the unsharded ladder: same permutation, swaps and configurations."""
    import torch
    W, H, G = 64, 8, 8
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(3, G)
    betas = np.linspace(0.38, 0.5, G)
    full = capi.States(g, seeds)
    full.pt_attach(betas, 0, G, 1, 99)
    shards = [capi.States(g, seeds[:4]), capi.States(g, seeds[4:])]
    for k, sh in enumerate(shards):
        sh.pt_attach(betas, 4 * k, 4, 2, 99)
    bufs = [sh.pt_buffers() for sh in shards]
    streams = [sh.pt_stream() for sh in shards]
    assert bufs[0][0].shape == (4,) and bufs[0][1].shape == (8,) and bufs[0][1].dtype == torch.float64
    for rnd in range(12):
        full.pt_time_steps(2); full.pt_measure(); full.pt_swap()
        for sh in shards:
            sh.pt_time_steps(2)
            sh.pt_measure()
        for sh in shards:
            sh.synchronize()
        for k in range(2):                                  # "all-gather": rank-major concatenation of the locals
            with torch.cuda.stream(streams[k]):
                bufs[k][1][:4].copy_(bufs[0][0])
                bufs[k][1][4:].copy_(bufs[1][0])
        for sh in shards:
            sh.pt_swap()
    perm, rounds, swaps = full.pt_state()
    assert rounds == 12 and swaps > 0
    for sh in shards:
        p, r, s = sh.pt_state()
        assert np.array_equal(p, perm) and r == rounds
    assert shards[0].pt_state()[2] == swaps                 # every rank counts the same accepted swaps
    assert np.array_equal(np.concatenate([sh.packed() for sh in shards]), full.packed())


@pytest.mark.parametrize("kind", ["lattice", "lattice_big", "general", "packed"])
def test_run_sampling_equals_step_by_step_loop(capi, exact, monkeypatch, kind):
    """isingmc_run_sampling (everything enqueued, one wait per chunk) against the loop it replaces."""
    if kind == "packed":
        monkeypatch.setenv("ISINGMC_FORCE_PACKED", "1")
    if kind == "lattice":
        ea, eb, ej = exact.square_lattice_edges(64, 16, -1.0); R = 5
    elif kind == "lattice_big":
        ea, eb, ej = exact.square_lattice_edges(1024, 64, 1.0, np.random.default_rng(1)); R = 3   # multi-launch path
    elif kind == "general":
        rng = np.random.default_rng(2)
        ea = rng.integers(0, 80, 200).astype(np.uint64); eb = rng.integers(0, 80, 200).astype(np.uint64)
        ej = rng.normal(size=200); R = 4
    else:
        ea, eb, ej = exact.cubic_lattice_edges(6, -1.0); R = 37
    seeds = capi.make_seeds(4, R)
    g = capi.Graph(ea, eb, ej)
    a = capi.States(g, seeds)
    e_a, s_a = a.run_sampling(0.45, 3, 2, 4)
    b = capi.States(g, seeds)
    b.do_time_steps(3, 0.45)
    for k in range(4):
        b.do_time_steps(2, 0.45)
        assert np.array_equal(s_a[:, k, :], b.states())
        np.testing.assert_allclose(e_a[:, k], b.energies(), rtol=1e-12, atol=1e-9)
    assert a.timestep == b.timestep == 11
    a.do_time_steps(2, 0.45); b.do_time_steps(2, 0.45)            # the temporary per-replica betas are gone again
    assert np.array_equal(a.states(), b.states())
    e0, s0 = a.run_sampling(0.45, 2, 3, 0)                        # no samples: thermalisation only
    assert e0.shape == (R, 0) and a.timestep == 15


@pytest.mark.parametrize("kind", ["lattice", "general", "packed"])
def test_sampling_pipeline_slabs(capi, exact, monkeypatch, kind):
    """isingmc_run_sampling works through slabs of samples, two in flight (device sweeps + copy-out of slab j while the
    host expands slab j-1).  One sample per slab, an odd number of slabs, and the default (everything in one slab)
    must give the same arrays."""
    if kind == "lattice":
        ea, eb, ej = exact.square_lattice_edges(1024, 64, -1.0)
    elif kind == "general":
        ea, eb, ej = exact.square_lattice_edges(30, 20, -1.0, np.random.default_rng(2))
    else:
        monkeypatch.setenv("ISINGMC_FORCE_PACKED", "1")
        ea, eb, ej = exact.cubic_lattice_edges(10, -1.0)
    g = capi.Graph(ea, eb, ej, force_general=kind != "lattice")
    seeds = capi.make_seeds(4, 5)
    out = []
    for slab_bytes in (None, "1", str(3 * 5 * g.state_words * 4)):      # default / 1 sample per slab / 3 per slab (7 = 3 + 3 + 1)
        if slab_bytes is None:
            monkeypatch.delenv("ISINGMC_SAMPLE_SLAB_BYTES", raising=False)
        else:
            monkeypatch.setenv("ISINGMC_SAMPLE_SLAB_BYTES", slab_bytes)
        st = capi.States(g, seeds)
        e, s = st.run_sampling(0.45, 3, 2, 7)
        e2, s2 = st.run_sampling(0.45, 0, 1, 2)                          # the buffers are reused by the next call
        out.append((e, s, e2, s2, st.energies()))
    for other in out[1:]:
        for a, b in zip(out[0], other):
            np.testing.assert_array_equal(a, b)
    assert np.array_equal(out[0][4], out[0][2][:, -1])                   # the last sample is the current configuration


@pytest.mark.parametrize("kind", ["csr_streaming", "packed"])
def test_per_step_energies_equal_step_by_step_measurements(capi, exact, monkeypatch, kind):
    """Energies after every timestep (lattice.rs:445-455) on the non-resident CSR path and on the replica-packed
    path: the measurements are enqueued behind their sweeps and read back per chunk; they must equal one
    get_energies call after every single step."""
    if kind == "packed":
        ea, eb, ej = exact.cubic_lattice_edges(40, -1.0)            # 64 000 sites: too big for the LDS-resident kernel
        R, beta = 40, 0.25
    else:
        monkeypatch.setenv("ISINGMC_DISABLE_RESIDENT", "1")
        monkeypatch.setenv("ISINGMC_DISABLE_PACKED", "1")
        rng = np.random.default_rng(7)
        ea = rng.integers(0, 3000, 9000).astype(np.uint64); eb = rng.integers(0, 3000, 9000).astype(np.uint64)
        ej = rng.normal(size=9000)
        R, beta = 5, 0.7
    seeds = capi.make_seeds(11, R)
    g = capi.Graph(ea, eb, ej)
    a = capi.States(g, seeds)
    eps = a.do_time_steps(7, beta, per_step_energies=True)
    b = capi.States(g, seeds)
    for k in range(7):
        b.do_time_steps(1, beta)
        np.testing.assert_allclose(eps[:, k], b.energies(), rtol=1e-12, atol=1e-9)
    assert np.array_equal(a.packed(), b.packed())


def test_bias_sign_compat_switch(mod, monkeypatch):
    """The bias sign is a crate-internal convention: E = sum J s s - sum h s by default (a strong positive
    bias aligns the spins with True); ISINGMC_COMPAT_BIAS_SIGN=-1 gives E = sum J s s + sum h s -- the mirrored chain."""
    edges = [((i, (i + 1) % 8), -0.1) for i in range(8)]
    lat = mod.Lattice(edges, 3)
    lat.set_global_bias(2.0)
    e, s = lat.run_monte_carlo(5.0, 50, 4)
    assert s.all() and np.allclose(e, -0.8 - 16.0)
    monkeypatch.setenv("ISINGMC_COMPAT_BIAS_SIGN", "-1")
    lat2 = mod.Lattice(edges, 3)
    lat2.set_global_bias(2.0)
    e2, s2 = lat2.run_monte_carlo(5.0, 50, 4)
    assert not s2.any() and np.array_equal(e2, e)


def test_cached_resources_can_be_released(capi, exact):
    """Freed device blocks, pinned buffers, streams and events wait for the next call (small calls are dominated by their
    creation otherwise); isingmc_release_cached_resources hands them back, and the next call simply allocates again."""
    ea, eb, ej = exact.square_lattice_edges(64, 16, -1.0)
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(3, 4)
    st = capi.States(g, seeds)
    st.do_time_steps(3, 0.4)
    want_e, want_s = st.energies(), st.states()
    del st
    assert capi.release_cached_resources() > 0
    assert capi.release_cached_resources() == 0
    st = capi.States(g, seeds)
    st.do_time_steps(3, 0.4)
    assert np.array_equal(st.energies(), want_e) and np.array_equal(st.states(), want_s)
