"""recognised lattices with a uniform field (Lattice.set_global_bias, lattice.rs:129-131; ClassicIsing's
longitudinal field, classicising.rs:69) or with open boundaries stay on the bit-sliced checkerboard path (multi-class
kernels, csrc/mc_kernels.hpp) and must reproduce oracle engine B -- spin by spin, field and missing bonds included --
bit for bit; energies E = sum J s s - h sum s from the integer counters."""
import math
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SEEDS = np.array([0x0123456789ABCDEF, 42, 2**64 - 1], dtype=np.uint64)


def _edges(ea, eb, ej):
    return [((int(a), int(b)), float(j)) for a, b, j in zip(ea, eb, ej)]


def _open_lattice(exact, W, H, J, rng, open_x, open_y):
    ea, eb, ej = exact.square_lattice_edges(W, H, J, rng)
    keep = np.ones(len(ea), dtype=bool)
    if open_x:
        keep &= ~((ea % W == W - 1) & (eb % W == 0))                # right bonds of the last column
    if open_y:
        keep &= ~((ea // W == H - 1) & (eb // W == 0))              # down bonds of the last row
    return ea, eb, ej, keep


def _check(capi, oracle, g, lat, ea, eb, ej, nvars, biases, T, betas, seeds=SEEDS):
    st = capi.States(g, seeds)
    ref = [lat.init(s) for s in seeds]
    np.testing.assert_array_equal(st.packed(), np.stack(ref))
    eps = st.do_time_steps(T, betas, per_step_energies=True)
    for r, s in enumerate(seeds):
        for t in range(T):
            lat.sweep(ref[r], s, t, betas[t] if np.ndim(betas) else betas)
            assert eps[r, t] == lat.energy_mag(ref[r])[0], (r, t)
    np.testing.assert_array_equal(st.packed(), np.stack(ref), err_msg="state after sweeps")
    st.do_time_steps(3, 0.6)                                          # no per-step energies: the plain launches
    for r, s in enumerate(seeds):
        for t in range(T, T + 3):
            lat.sweep(ref[r], s, t, 0.6)
    np.testing.assert_array_equal(st.packed(), np.stack(ref))
    em = [lat.energy_mag(x) for x in ref]
    np.testing.assert_array_equal(st.energies(), [e for e, _ in em])
    np.testing.assert_array_equal(st.magnetisations(), [m for _, m in em])
    spins = st.states()
    for r in range(len(seeds)):                                       # K1: energy recomputed from the returned spins
        np.testing.assert_allclose(st.energies()[r], oracle.energy(ea, eb, ej, nvars, spins[r], biases=biases), rtol=1e-12, atol=1e-9)
    return st


@pytest.mark.parametrize("W,H", [(256, 16), (512, 32), (1024, 8)])
@pytest.mark.parametrize("h", [0.25, -0.7, 2.0, -2.0])
@pytest.mark.parametrize("glass", [False, True])
def test_uniform_field_bit_exact(capi, oracle, exact, W, H, h, glass):
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0, np.random.default_rng(W) if glass else None)
    biases = np.full(W * H, h)
    g = capi.Graph(ea, eb, ej, biases=biases)
    assert g.kind == capi.KIND_LATTICE2D and g.info.fast_path == 1 and g.info.field == h
    if glass:
        lat = oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8), field=h)
    else:
        lat = oracle.Lat(W, H, 1.0, 0, field=h)
    _check(capi, oracle, g, lat, ea, eb, ej, W * H, biases, 5, np.array([0.0, 0.2, 0.4407, 0.9, 2.5]))


def test_field_scaled_coupling_and_per_replica_betas(capi, oracle, exact):
    W, H, J, h = 256, 32, 0.37, 0.5                                   # antiferromagnetic |J| = 0.37, h <= 2|J|
    ea, eb, ej = exact.square_lattice_edges(W, H, J)
    g = capi.Graph(ea, eb, ej, biases=np.full(W * H, h))
    assert g.info.fast_path == 1
    lat = oracle.Lat(W, H, J, 1, field=h)
    st = capi.States(g, SEEDS)
    betas = [0.3, 1.1, 4.0]
    st.set_betas(betas)
    st.do_time_steps(6)
    for r, s in enumerate(SEEDS):
        ref = lat.init(s)
        for t in range(6):
            lat.sweep(ref, s, t, betas[r])
        np.testing.assert_array_equal(st.packed()[r], ref)
        assert st.energies()[r] == lat.energy_mag(ref)[0]
    # a field beyond 2|J| (classes with fewer than two satisfied bonds would cost energy too) and site-dependent
    # biases take the general path
    assert capi.Graph(ea, eb, ej, biases=np.full(W * H, 0.75)).kind == capi.KIND_GENERAL
    b = np.full(W * H, h); b[5] = 0.0
    assert capi.Graph(ea, eb, ej, biases=b).kind == capi.KIND_GENERAL
    assert capi.Graph(ea, eb, ej, biases=np.zeros(W * H)).info.fast_path == 0     # h = 0: the two-class kernels


@pytest.mark.parametrize("W,H", [(256, 16), (512, 32)])
@pytest.mark.parametrize("open_x,open_y", [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize("glass", [False, True])
def test_open_boundaries_bit_exact(capi, oracle, exact, W, H, open_x, open_y, glass):
    ea, eb, ej, keep = _open_lattice(exact, W, H, -1.0, np.random.default_rng(H) if glass else None, open_x, open_y)
    perm = np.random.default_rng(1).permutation(int(keep.sum()))     # any edge order
    a, b, j = ea[keep][perm], eb[keep][perm], ej[keep][perm]
    r = capi.recognise_lattice2d(a, b, j, W * H)
    assert r["is_lattice"] and r["open_x"] == open_x and r["open_y"] == open_y
    g = capi.Graph(a, b, j, nvars=W * H)
    assert g.kind == capi.KIND_LATTICE2D and g.info.fast_path == 2 and (bool(g.info.open_x), bool(g.info.open_y)) == (open_x, open_y)
    if glass:
        lat = oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8), open_x=open_x, open_y=open_y)
    else:
        lat = oracle.Lat(W, H, 1.0, 0, open_x=open_x, open_y=open_y)
    _check(capi, oracle, g, lat, a, b, j, W * H, None, 5, np.array([0.0, 0.3, 0.4407, 1.2, 3.0]))


@pytest.mark.parametrize("W,H", [(256, 16), (512, 32)])
@pytest.mark.parametrize("open_x,open_y", [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize("h", [0.3, -0.75, 1.0])
@pytest.mark.parametrize("glass", [False, True])
def test_field_on_an_open_lattice_bit_exact(capi, oracle, exact, W, H, open_x, open_y, h, glass):
    """Open boundaries AND a uniform field |h| <= |J|: nine classes (m = sat - unsat in 0..4, spin along / against the field)."""
    ea, eb, ej, keep = _open_lattice(exact, W, H, -1.0, np.random.default_rng(W) if glass else None, open_x, open_y)
    a, b, j = ea[keep], eb[keep], ej[keep]
    biases = np.full(W * H, h)
    g = capi.Graph(a, b, j, nvars=W * H, biases=biases)
    assert g.kind == capi.KIND_LATTICE2D and g.info.fast_path == 4 and g.info.field == h and not g.info.field_signs
    kw = dict(field=h, open_x=open_x, open_y=open_y)
    lat = (oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8), **kw) if glass
           else oracle.Lat(W, H, 1.0, 0, **kw))
    _check(capi, oracle, g, lat, a, b, j, W * H, biases, 5, np.array([0.0, 0.3, 0.4407, 1.2, 3.0]))


@pytest.mark.parametrize("W,H", [(256, 16), (512, 32)])
@pytest.mark.parametrize("open_x,open_y", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("glass", [False, True])
def test_random_field_signs_bit_exact(capi, oracle, exact, W, H, open_x, open_y, glass):
    """Biases of one size and both signs (the bimodal random-field model, h_i = +-h): sign planes turn the spin bit into
    "along the site's field"; periodic (six classes) and open (nine); energy from the packed third counter."""
    h = 0.8
    ea, eb, ej, keep = _open_lattice(exact, W, H, -1.0, np.random.default_rng(W) if glass else None, open_x, open_y)
    a, b, j = ea[keep], eb[keep], ej[keep]
    rng = np.random.default_rng(H)
    biases = h * rng.choice([-1.0, 1.0], W * H)
    g = capi.Graph(a, b, j, nvars=W * H, biases=biases)
    assert g.kind == capi.KIND_LATTICE2D and g.info.fast_path == (4 if open_x or open_y else 1)
    assert g.info.field == h and g.info.field_signs
    kw = dict(field=h, open_x=open_x, open_y=open_y, field_neg=(biases < 0).astype(np.uint8))
    lat = (oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8), **kw) if glass
           else oracle.Lat(W, H, 1.0, 0, **kw))
    _check(capi, oracle, g, lat, a, b, j, W * H, biases, 5, np.array([0.0, 0.3, 0.4407, 1.2, 3.0]))
    # per-replica betas on the same graph
    st = capi.States(g, SEEDS)
    betas = [0.25, 0.9, -0.2]
    st.set_betas(betas)
    st.do_time_steps(4)
    for r, s in enumerate(SEEDS):
        ref = lat.init(s)
        for t in range(4):
            lat.sweep(ref, s, t, betas[r])
        np.testing.assert_array_equal(st.packed()[r], ref)
        assert st.energies()[r] == lat.energy_mag(ref)[0]


def test_fields_the_multi_class_kernels_cannot_take_are_general(capi, exact):
    W, H = 256, 16
    ea, eb, ej, keep = _open_lattice(exact, W, H, -1.0, None, True, False)
    a, b, j = ea[keep], eb[keep], ej[keep]
    assert capi.Graph(a, b, j, nvars=W * H, biases=np.full(W * H, 1.2)).kind == capi.KIND_GENERAL      # |h| > |J| on an open lattice
    bb = np.full(W * H, 0.5); bb[7] = -0.25                                                            # two sizes
    assert capi.Graph(a, b, j, nvars=W * H, biases=bb).kind == capi.KIND_GENERAL
    bz = np.full(W * H, 0.5); bz[9] = 0.0                                                              # a site without a field
    assert capi.Graph(ea, eb, ej, biases=bz).kind == capi.KIND_GENERAL


def test_open_lattice_with_a_missing_interior_bond_is_general(capi, exact):
    W, H = 256, 16
    ea, eb, ej, keep = _open_lattice(exact, W, H, -1.0, None, True, False)
    k2 = keep.copy(); k2[np.flatnonzero(keep)[7]] = False             # one interior bond gone: not a lattice
    assert not capi.recognise_lattice2d(ea[k2], eb[k2], ej[k2], W * H)["is_lattice"]
    k3 = keep.copy(); k3[np.flatnonzero(~keep)[0]] = True             # one wrap-around bond present, the others not
    assert not capi.recognise_lattice2d(ea[k3], eb[k3], ej[k3], W * H)["is_lattice"]


@pytest.mark.parametrize("mode", ["field", "open", "open_field", "field_signs", "open_field_signs", "aniso"])
def test_multi_class_kernels_equilibrium_against_the_general_path(capi, exact, mode):
    """Independent check of the multi-class kernels' physics: the thread-per-site CSR path (f64 local fields, other update
    order within a colour class, other random numbers; itself checked against exact enumeration) samples the same
    Boltzmann distribution -- two-sample z-test on <E> and <M> over 48 replicas each."""
    W, H, h, beta, R = 256, 16, 0.3, 0.35, 48
    ea, eb, ej, keep = _open_lattice(exact, W, H, -1.0, None, mode.startswith("open"), mode.startswith("open"))
    ea, eb, ej = ea[keep], eb[keep], ej[keep].copy()
    biases = np.full(W * H, h) if "field" in mode else None
    if mode.endswith("signs"):
        biases = h * np.random.default_rng(12).choice([-1.0, 1.0], W * H)
    if mode == "aniso":
        ej[1::2] *= 0.6
    obs = []
    for force in (False, True):
        g = capi.Graph(ea, eb, ej, nvars=W * H, biases=biases, force_general=force)
        assert (g.kind == capi.KIND_GENERAL) == force
        st = capi.States(g, capi.make_seeds(3 + force, R))
        st.do_time_steps(300, beta)
        e = st.do_time_steps(600, beta, per_step_energies=True).mean(axis=1)
        obs.append((e, st.magnetisations().astype(np.float64)))
    for a, b, name in ((obs[0][0], obs[1][0], "E"), (obs[0][1], obs[1][1], "M")):
        z = (a.mean() - b.mean()) / math.sqrt(a.var(ddof=1) / R + b.var(ddof=1) / R)
        assert abs(z) < 4.5, (name, z, a.mean(), b.mean())
    if mode in ("field", "open_field"):
        assert obs[0][1].mean() > 0.3 * W * H                        # the field magnetises the paramagnet


def test_python_api_keeps_field_and_open_lattices_on_the_fast_path(oracle, exact):
    import py_monte_carlo as mod
    W, H = 256, 16
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    lat = mod.Lattice.from_arrays(ea, eb, ej, seed_gen=9)
    lat.set_global_bias(0.25)                                         # lattice.rs:129-131
    info = lat.engine_info()
    assert info["kind"] == "lattice2d" and info["field"] == 0.25
    e, s = lat.run_monte_carlo(0.4, 12, 3)
    olat = oracle.Lat(W, H, 1.0, 0, field=0.25)
    for r, seed in enumerate(lat.make_seeds(3)):
        ref = olat.init(seed)
        for t in range(12):
            olat.sweep(ref, seed, t, 0.4)
        assert np.array_equal(s[r], olat.unpack(ref).astype(bool)) and e[r] == olat.energy_mag(ref)[0]
    es, ss = lat.run_monte_carlo_sampling(0.4, 6, 3, None, 2, 3)      # thermalisation 2, a sample every 3 steps
    ea2, _ = lat.run_monte_carlo_annealing_and_get_energies([(0, 0.1), (8, 0.8)], 8, 3)
    assert es.shape == (3, 2) and ss.shape == (3, 2, W * H) and ea2.shape == (3, 8)
    lat.set_individual_bias(3, 1.0)                                   # site-dependent: general path
    assert lat.engine_info()["kind"] == "general"
    ci = mod.ClassicIsing(_edges(ea, eb, ej), 0.5, 2, 7)              # longitudinal field (classicising.rs:69)
    ci.run_monte_carlo(0.3, 5)
    olat = oracle.Lat(W, H, 1.0, 0, field=0.5)
    for r, seed in enumerate(oracle.make_seeds(7, 2)):
        ref = olat.init(seed)
        for t in range(5):
            olat.sweep(ref, seed, t, 0.3)
        assert np.array_equal(ci.get_states()[r], olat.unpack(ref).astype(bool)) and ci.get_energies()[r] == olat.energy_mag(ref)[0]


def test_tempering_on_a_field_lattice_uses_the_host_swap_step(capi, exact):
    from helpers import OracleLatEngine  # noqa: F401  (the oracle engine of the plain lattice is not used here)
    from pyisingmontecarlo_amd.tempering import ClassicalTempering, HipEngine
    W, H = 256, 16
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)

    class FieldEngine(HipEngine):
        def __init__(self):
            self.graph = capi.Graph(ea, eb, ej, biases=np.full(W * H, 0.2))
            self.nvars = self.graph.nvars
            self.supports_on_stream_pt = self.graph.kind == capi.KIND_LATTICE2D and self.graph.info.fast_path == 0

    pt = ClassicalTempering((ea, eb, ej), seed=5, engine_factory=FieldEngine)
    for b in np.linspace(0.40, 0.44, 6):
        pt.add_graph(float(b))
    pt.timesteps(20, replica_swap_freq=2)
    assert not pt._on_stream and pt.get_total_swaps() > 0 and sorted(pt.get_permutation()) == list(range(6))


def _aniso_edges(exact, W, H, jx, jy, rng):
    """Right bonds +-jx, down bonds +-jy (signs: all negative, or random when rng)."""
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0, rng)
    ej = ej.copy()
    ej[0::2] *= jx
    ej[1::2] *= jy
    return ea, eb, ej


@pytest.mark.parametrize("W,H", [(256, 16), (512, 32), (1024, 8)])
@pytest.mark.parametrize("jx,jy", [(1.0, 0.5), (0.3, 1.7), (1.0, 3.0), (2.0, 1.0), (1.0, 0.0)])
@pytest.mark.parametrize("glass", [False, True])
def test_anisotropic_couplings_bit_exact(capi, oracle, exact, W, H, jx, jy, glass):
    """|J| differs between the horizontal and the vertical bonds: five classes (kx, ky) of the multi-class kernel,
    dE = 2|Jx|(sat_x - unsat_x) + 2|Jy|(sat_y - unsat_y); the energy from two bond counters."""
    ea, eb, ej = _aniso_edges(exact, W, H, jx, jy, np.random.default_rng(W + H) if glass else None)
    perm = np.random.default_rng(3).permutation(len(ea))
    a, b, j = ea[perm], eb[perm], ej[perm]
    r = capi.recognise_lattice2d(a, b, j, W * H)
    assert r["is_lattice"] and r.get("anisotropic") and r["jabs"] == jx
    g = capi.Graph(a, b, j)
    assert g.kind == capi.KIND_LATTICE2D and g.info.fast_path == 3 and (g.info.jabs, g.info.jabs_y) == (jx, jy)
    if glass:
        lat = oracle.Lat(W, H, jx, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8), jabs_y=jy)
    else:
        lat = oracle.Lat(W, H, jx, 0, jabs_y=jy)
    _check(capi, oracle, g, lat, a, b, j, W * H, None, 5, np.array([0.0, 0.3, 0.4407, 1.2, 3.0]))


def test_anisotropic_per_replica_betas_sampling_and_general_fallbacks(capi, oracle, exact):
    W, H, jx, jy = 256, 32, 0.8, 1.3
    ea, eb, ej = _aniso_edges(exact, W, H, jx, jy, None)
    ej = -ej                                                          # antiferromagnetic
    g = capi.Graph(ea, eb, ej)
    assert g.info.fast_path == 3
    lat = oracle.Lat(W, H, jx, 1, jabs_y=jy)
    st = capi.States(g, SEEDS)
    betas = [0.3, 1.1, -0.4]
    st.set_betas(betas)
    st.do_time_steps(6)
    for r, s in enumerate(SEEDS):
        ref = lat.init(s)
        for t in range(6):
            lat.sweep(ref, s, t, betas[r])
        np.testing.assert_array_equal(st.packed()[r], ref)
        assert st.energies()[r] == lat.energy_mag(ref)[0]
    # anisotropy together with a field or open boundaries, or a third |J|: the general path
    assert capi.Graph(ea, eb, ej, biases=np.full(W * H, 0.2)).kind == capi.KIND_GENERAL
    keep = ~((ea % W == W - 1) & (eb % W == 0))
    assert capi.Graph(ea[keep], eb[keep], ej[keep], nvars=W * H).kind == capi.KIND_GENERAL
    ej3 = ej.copy(); ej3[4] *= 2.0
    assert not capi.recognise_lattice2d(ea, eb, ej3, W * H)["is_lattice"]
    assert capi.Graph(ea, eb, ej3).kind == capi.KIND_GENERAL


def test_anisotropic_decoupled_chains_match_the_exact_energy(capi, exact):
    """Independent check of the anisotropic kernel's physics against an exact result: for |Jy| = 0 the rows are
    independent periodic chains, whose energy per bond is -|Jx| tanh(beta |Jx|) up to O(tanh^W) corrections."""
    W, H, jx, beta = 256, 64, 1.0, 0.7
    ea, eb, ej = _aniso_edges(exact, W, H, jx, 0.0, None)
    g = capi.Graph(ea, eb, ej)
    assert g.info.fast_path == 3
    st = capi.States(g, np.arange(16, dtype=np.uint64) + 5)
    st.do_time_steps(400, beta)
    acc = []
    for _ in range(40):
        st.do_time_steps(10, beta)
        acc.append(st.energies().mean())
    e_bond = np.mean(acc) / (W * H)
    expect = -jx * math.tanh(beta * jx)
    # 16 replicas x 40 samples x 16384 bonds; chain energy variance per bond = J^2 sech^2 ~ 0.63 -> sigma ~ 2.5e-4 (correlated: x3)
    assert abs(e_bond - expect) < 2.5e-3, (e_bond, expect)


@pytest.mark.parametrize("mode", ["field", "open", "open_field_signs", "aniso"])
def test_multi_class_resident_kernel_equals_per_colour_launches(capi, exact, monkeypatch, mode):
    """Small lattices: lat_mc_resident_kernel (planes in LDS, all timesteps in one launch) against the per-colour launches
    (ISINGMC_DISABLE_RESIDENT=1) -- same spins, same energies after every timestep; beta schedules, then per-replica betas.
    (The oracle comparisons of this file at 256 x 16 ... 1024 x 8 run the resident kernel; the per-colour launches are
    compared with the oracle at the wide geometries and through this equality.)"""
    W, H, R = 512, 64, 5
    ea, eb, ej, keep = _open_lattice(exact, W, H, -1.0, np.random.default_rng(3), mode.startswith("open"), mode.startswith("open"))
    ea, eb, ej = ea[keep], eb[keep], ej[keep].copy()
    biases = np.full(W * H, 0.4) if "field" in mode else None
    if mode.endswith("signs"):
        biases = 0.4 * np.random.default_rng(12).choice([-1.0, 1.0], W * H)
    if mode == "aniso":
        ea, eb, ej = _aniso_edges(exact, W, H, 1.0, 0.5, np.random.default_rng(3))
    out = []
    for disable in ("0", "1"):
        monkeypatch.setenv("ISINGMC_DISABLE_RESIDENT", disable)
        g = capi.Graph(ea, eb, ej, nvars=W * H, biases=biases)
        assert g.kind == capi.KIND_LATTICE2D and g.info.fast_path > 0
        st = capi.States(g, capi.make_seeds(4, R))
        st.do_time_steps(9, np.linspace(0.1, 1.0, 9))
        mid = st.packed().copy()
        eps = st.do_time_steps(6, np.linspace(1.0, 0.3, 6), per_step_energies=True)   # counters inside the launch vs a measurement pass per step
        st.set_betas(np.linspace(0.2, 0.9, R))
        st.do_time_steps(7)
        out.append((mid, eps, st.packed().copy(), st.energies(), st.magnetisations()))
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("W,H", [(4096, 8), (16384, 4), (2048, 16)])
@pytest.mark.parametrize("mode", ["field", "open_field", "aniso"])
def test_multi_class_streaming_kernels_at_wide_geometries(capi, oracle, exact, monkeypatch, W, H, mode):
    """The streaming instantiations with the 2^k quad mapping at BASELINE c2's width (cols_log2 = 4), at W >= 16384 (the
    branch of load_quad_uni in which a wave never leaves its row) and at 2048 -- lattices this small would otherwise run the
    LDS-resident kernel, so it is switched off -- against the oracle, spin by spin."""
    monkeypatch.setenv("ISINGMC_DISABLE_RESIDENT", "1")
    opn = mode.startswith("open")
    if mode == "aniso":
        ea, eb, ej = _aniso_edges(exact, W, H, 0.7, 1.9, np.random.default_rng(W))
        biases, kw = None, dict(jabs_y=1.9)
        jabs = 0.7
    else:
        ea, eb, ej, keep = _open_lattice(exact, W, H, -1.0, np.random.default_rng(W), opn, opn)
        ea, eb, ej = ea[keep], eb[keep], ej[keep]
        biases, kw, jabs = np.full(W * H, -0.6), dict(field=-0.6, open_x=opn, open_y=opn), 1.0
    g = capi.Graph(ea, eb, ej, nvars=W * H, biases=biases)
    assert g.kind == capi.KIND_LATTICE2D and g.info.fast_path == {"field": 1, "open_field": 4, "aniso": 3}[mode]
    full = exact.square_lattice_edges(W, H, -1.0, np.random.default_rng(W))[2] if mode != "aniso" else ej
    lat = oracle.Lat(W, H, jabs, 0, (full[0::2] > 0).astype(np.uint8), (full[1::2] > 0).astype(np.uint8), **kw)
    st = capi.States(g, SEEDS[:2])
    st.do_time_steps(3, 0.5)
    for r, sd in enumerate(SEEDS[:2]):
        ref = lat.init(sd)
        for t in range(3):
            lat.sweep(ref, sd, t, 0.5)
        np.testing.assert_array_equal(st.packed()[r], ref)
        assert st.energies()[r] == lat.energy_mag(ref)[0]
