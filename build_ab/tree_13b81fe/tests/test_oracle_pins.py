"""The oracle against everything that can pin it without the (unbuildable) reference:
published RNG known-answer vectors, hand-checked README energies, exact enumeration, Kaufman.
The reference ships no tests or fixtures: parity with the Rust crate itself is
UNPINNED; these are the pins that exist."""
import json
import math
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_philox_known_answers(oracle):
    """Random123 kat_vectors, philox4x32-10."""
    for v in _golden("philox_kat.json")["vectors"]:
        out = oracle.philox([int(x, 16) for x in v["ctr"]], [int(x, 16) for x in v["key"]])
        assert [int(x) for x in out] == [int(x, 16) for x in v["out"]]


def test_xoshiro256pp_reference_vector(oracle):
    g = _golden("xoshiro256pp.json")
    assert [int(x) for x in oracle.xoshiro_from_state(g["state"], len(g["out"]))] == g["out"]


def test_make_seeds_golden_and_prefix_property(oracle):
    g = _golden("make_seeds.json")
    for case in g["cases"]:
        assert [int(x) for x in oracle.make_seeds(case["seed_gen"], len(case["seeds"]))] == case["seeds"]
    # lattice.rs:83-91: the master rng is re-seeded per call, so n seeds are a prefix of n+k seeds
    assert list(oracle.make_seeds(99, 3)) == list(oracle.make_seeds(99, 7)[:3])


def test_readme_chain_energies(oracle):
    """README.md:45-53: E = J*Sza*Szb, positive J antiferromagnetic; edges ((0,1),1.0), ((1,2),-1.0)."""
    ea, eb, ej = oracle.split_edges([((0, 1), 1.0), ((1, 2), -1.0)])
    for state, e in [((1, 1, 1), 0.0), ((1, 0, 1), 0.0), ((1, 0, 0), -2.0), ((1, 1, 0), 2.0), ((0, 0, 0), 0.0)]:
        assert oracle.energy(ea, eb, ej, 3, np.array(state, dtype=np.uint8)) == e
    # bias convention of this build: E -= h*s
    assert oracle.energy(ea, eb, ej, 3, np.array([1, 1, 1], dtype=np.uint8), biases=[0.5, 0, -1]) == 0.5


def test_det_exp_accuracy_and_edges(oracle):
    xs = np.concatenate([-np.logspace(-12, math.log10(39.9), 400), [-1e-300, -0.0, 0.0, 3.0, -40.0000001, -1e9]])
    for x in xs:
        got, ref = oracle.det_exp(float(x)), math.exp(min(x, 0.0))
        if x < -40:
            assert got == 0.0
        else:
            assert abs(got - ref) <= 4e-16 * ref, x


def test_threshold_fixed_point(oracle):
    bits = 39  # N_PLANES (7) + 32
    assert oracle.threshold_fixed(0.4407, 0.0) == 1 << bits
    assert oracle.threshold_fixed(0.0, 8.0) == 1 << bits       # beta = 0: every move accepted (K4)
    assert oracle.threshold_fixed(-1.0, 4.0) == 1 << bits      # negative temperature never rejects
    for beta in (0.1, 0.4407, 1.0, 5.0):
        t3, t4 = oracle.threshold_fixed(beta, 4.0), oracle.threshold_fixed(beta, 8.0)
        assert t3 == math.floor(math.exp(-4 * beta) * 2 ** bits)
        # K5 detailed balance: p(k=4) = p(k=3)^2 up to the fixed-point floor
        assert abs(t4 / 2 ** bits - (t3 / 2 ** bits) ** 2) < 2 ** -(bits - 2)
    assert oracle.threshold_fixed(200.0, 8.0) == 0


def test_kaufman_matches_enumeration(exact):
    for (W, H) in [(4, 4), (6, 4)]:
        ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
        for beta in (0.2, 0.4407, 0.8):
            en = exact.enumerate_graph(ea, eb, ej, W * H, beta)
            assert abs(en["E"] - exact.kaufman_energy(W, H, beta)) < 1e-8


def test_kaufman_golden(exact):
    for c in _golden("kaufman.json")["cases"]:
        got = exact.kaufman_energy(c["W"], c["H"], c["beta"])
        assert abs(got - c["E"]) <= 1e-9 * abs(c["E"])
    # thermodynamic-limit check at beta_c: e = -sqrt(2) + O(1/L)
    e = exact.kaufman_energy(4096, 4096, 0.5 * math.log(1 + math.sqrt(2))) / 4096 ** 2
    assert abs(e + math.sqrt(2)) < 1e-3


def _blocked_err(x, nblocks=20):
    b = np.array_split(np.asarray(x, dtype=np.float64), nblocks)
    m = np.array([v.mean() for v in b])
    return m.std(ddof=1) / math.sqrt(nblocks)


def test_lattice_engine_vs_kaufman(oracle, exact):
    """K3: the checkerboard spec engine samples the right distribution (64x64, beta=0.3 and 0.4407)."""
    W = H = 64
    lat = oracle.Lat(W, H)
    for beta in (0.3, 0.4407):
        es = []
        for seed in (3, 4):
            st = lat.init(seed)
            for t in range(1500):
                lat.sweep(st, seed, t, beta)
                if t >= 300:
                    es.append(lat.energy_mag(st)[0])
        ref = exact.kaufman_energy(W, H, beta)
        err = _blocked_err(es)
        assert abs(np.mean(es) - ref) < 4 * err, (beta, np.mean(es), ref, err)
        assert err < 0.01 * abs(ref)


def test_lattice_engine_limits(oracle):
    """K4: beta=0 accepts every attempt; a cold ordered ferromagnet does not move."""
    lat = oracle.Lat(64, 8)
    st = lat.init(5)
    before = st.copy()
    lat.sweep(st, 5, 0, 0.0)
    assert np.array_equal(st, ~before)
    ordered = lat.pack(np.ones(64 * 8, dtype=np.uint8))
    e0 = lat.energy_mag(ordered)
    assert e0 == (-2.0 * 64 * 8, 64 * 8)
    for t in range(3):
        lat.sweep(ordered, 1, t, 50.0)
    assert lat.energy_mag(ordered) == e0


def test_lattice_antiferro_maps_to_ferro(oracle):
    """Gauge symmetry: flipping one sublattice maps J=+1 onto J=-1 with the same energy."""
    W, H = 64, 8
    rng = np.random.default_rng(0)
    spins = rng.integers(0, 2, (H, W)).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    gauged = spins ^ ((xx + yy) & 1).astype(np.uint8)
    e_f = oracle.Lat(W, H, 1.0, 0).energy_mag(oracle.Lat(W, H).pack(spins))[0]
    e_a = oracle.Lat(W, H, 1.0, 1).energy_mag(oracle.Lat(W, H).pack(gauged))[0]
    assert e_f == e_a


def test_general_engine_vs_enumeration(oracle, exact):
    """K2: general path on a random +-J graph with fields, against brute force."""
    rng = np.random.default_rng(3)
    n, m = 10, 22
    ea = rng.integers(0, n, m).astype(np.uint64)
    eb = (ea + rng.integers(1, n, m).astype(np.uint64)) % n
    ej = rng.choice([-1.0, 1.0], m) * rng.uniform(0.5, 1.5, m)
    biases = rng.normal(size=n) * 0.3
    beta = 0.6
    en = exact.enumerate_graph(ea, eb, ej, n, beta, biases)
    es, ms = [], []
    for seed in range(6):
        _, st, eps = oracle.gen_run(ea, eb, ej, n, seed + 100, [beta] * 6000, biases=biases, per_step=True)
        es += list(eps[500:])
    err = _blocked_err(es)
    assert abs(np.mean(es) - en["E"]) < 4 * err, (np.mean(es), en["E"], err)


def test_reference_faithful_engine_vs_enumeration(oracle, exact):
    """The random-site sequential engine (the timed CPU baseline) samples the same distribution."""
    ea, eb, ej = exact.square_lattice_edges(4, 4, -1.0)
    beta = 0.35
    en = exact.enumerate_graph(ea, eb, ej, 16, beta)
    _, _, eps = oracle.ref_run(ea, eb, ej, 16, oracle.make_seeds(5, 8), [beta] * 6000, per_step=True)
    es = eps[:, 500:].ravel()
    err = _blocked_err(es, 40)
    assert abs(es.mean() - en["E"]) < 4 * err


def test_lattice_sweep_golden_hashes(oracle):
    """Regression pin of the checkerboard spec: sha256 of the packed words."""
    import hashlib
    for c in _golden("lattice_sweeps.json")["cases"]:
        rng = np.random.default_rng(c["j_seed"]) if c["j_seed"] is not None else None
        if rng is None:
            lat = oracle.Lat(c["W"], c["H"], c["jabs"], c["jpos"])
        else:
            from oracle import exact as X
            _, _, ej = X.square_lattice_edges(c["W"], c["H"], c["jabs"], rng)
            lat = oracle.Lat(c["W"], c["H"], c["jabs"], 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8))
        st = lat.init(c["seed"])
        for t in range(c["T"]):
            lat.sweep(st, c["seed"], t, c["beta"])
        assert hashlib.sha256(st.tobytes()).hexdigest() == c["sha256"]
        assert lat.energy_mag(st)[0] == c["energy"]


def test_pt_swap_round_properties(oracle):
    betas = np.linspace(0.1, 1.0, 8)
    energies = -np.arange(8) * 5.0           # colder rungs hold lower energies: d = (b_i-b_j)(E_i-E_j) = -0.64
    perm = np.arange(8, dtype=np.uint32)
    total = 0
    for rnd in range(50):
        total += oracle.pt_swap_round(7, rnd, betas, energies, perm)
    assert sorted(perm) == list(range(8)) and total > 0
    # favourable swaps (d >= 0) are always accepted
    perm = np.arange(2, dtype=np.uint32)
    assert oracle.pt_swap_round(1, 0, np.array([0.2, 0.8]), np.array([-10.0, 5.0]), perm) == 1
    assert list(perm) == [1, 0]
