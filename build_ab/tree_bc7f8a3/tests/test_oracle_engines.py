"""Every engine of the CPU oracle against the one piece of the reference that IS in the tree, the Hamiltonian of README.md:45-46
(E = sum J s_a s_b - sum h s, evaluated edge by edge by `orc_energy`), on small inputs: K1 for each mode of the checkerboard
engine B (+-J, uniform and +-h fields, open boundaries, both, anisotropic couplings), for the coloured engine C, the
replica-packed engine D and the real-coupling engine E; K2 (exact enumeration) for engine D, whose spec c5 runs on.  CPU only:
tests/test_host_sanitizers.py re-runs this file against an ASan/UBSan build of the oracle."""
import numpy as np
import pytest


def _lattice_edges(W, H, jx, jy, open_x, open_y, signs=None):
    ea, eb, ej = [], [], []
    for y in range(H):
        for x in range(W):
            i = y * W + x
            if x + 1 < W or not open_x:
                ea.append(i); eb.append(y * W + (x + 1) % W)
                ej.append(jx * (signs[0][i] if signs else 1.0))
            if y + 1 < H or not open_y:
                ea.append(i); eb.append(((y + 1) % H) * W + x)
                ej.append(jy * (signs[1][i] if signs else 1.0))
    return np.array(ea, dtype=np.uint64), np.array(eb, dtype=np.uint64), np.array(ej, dtype=np.float64)


MODES = [
    dict(name="ferromagnet"),
    dict(name="antiferromagnet", jpos=1),
    dict(name="+-J glass", glass=True),
    dict(name="uniform field", field=0.75),
    dict(name="+-J glass in a field", glass=True, field=1.5),
    dict(name="random field +-h", field=0.5, random_field=True),
    dict(name="open in x", open_x=True),
    dict(name="open in both, glass", open_x=True, open_y=True, glass=True),
    dict(name="open with a field", open_y=True, field=0.8),
    dict(name="anisotropic", jabs_y=0.4),
    dict(name="anisotropic glass", jabs_y=2.5, glass=True),
]


@pytest.mark.parametrize("mode", MODES, ids=[m["name"] for m in MODES])
def test_checkerboard_engine_energy_equals_the_edge_list_hamiltonian(oracle, mode):
    """K1 for engine B: after sweeps at several betas, the engine's integer-counter energy and magnetisation equal the
    edge-by-edge sum over the unpacked spins; beta = 0 accepts every attempt (K4)."""
    W, H = 128, 6
    rng = np.random.default_rng(11)
    jabs, jabs_y = 1.25, mode.get("jabs_y")
    glass = mode.get("glass", False)
    jright = rng.integers(0, 2, W * H).astype(np.uint8) if glass else None   # 1 = J > 0
    jdown = rng.integers(0, 2, W * H).astype(np.uint8) if glass else None
    field = mode.get("field", 0.0)
    fneg = rng.integers(0, 2, W * H).astype(np.uint8) if mode.get("random_field") else None
    lat = oracle.Lat(W, H, jabs, mode.get("jpos", 0), jright, jdown, field=field, open_x=mode.get("open_x", False),
                     open_y=mode.get("open_y", False), jabs_y=jabs_y, field_neg=fneg)
    if glass:
        signs = (np.where(jright == 1, 1.0, -1.0), np.where(jdown == 1, 1.0, -1.0))
    else:
        s = 1.0 if mode.get("jpos", 0) else -1.0
        signs = (np.full(W * H, s), np.full(W * H, s))
    ea, eb, ej = _lattice_edges(W, H, jabs, jabs if jabs_y is None else jabs_y, mode.get("open_x", False), mode.get("open_y", False), signs)
    biases = None
    if field:
        biases = np.full(W * H, field) if fneg is None else np.where(fneg == 1, -field, field)
    st = lat.init(12345)
    for t, beta in enumerate([0.0, 0.3, 0.9, 3.0]):
        before = lat.unpack(st).copy()
        lat.sweep(st, 12345, t, beta)
        spins = lat.unpack(st)
        if beta == 0.0:
            assert np.array_equal(spins, 1 - before)           # every attempt accepted: the whole lattice flipped
        e, m = lat.energy_mag(st)
        assert m == 2 * int(spins.sum()) - W * H
        want = oracle.energy(ea, eb, ej, W * H, spins, biases)
        assert abs(e - want) <= 1e-9 * max(1.0, abs(want)), (mode["name"], beta, e, want)
    assert np.array_equal(lat.unpack(lat.pack(spins)), spins)


def _random_graph(rng, n, m, couplings):
    ea = rng.integers(0, n, m).astype(np.uint64)
    eb = (ea + rng.integers(1, n, m).astype(np.uint64)) % n
    return ea, eb, couplings(m)


def test_coloured_engine_energy_and_continuation(oracle):
    """Engine C: K1 with real couplings, biases and a self loop; a run split in two (t0, state) equals the run in one piece."""
    rng = np.random.default_rng(5)
    n = 70
    ea, eb, ej = _random_graph(rng, n, 160, lambda m: rng.normal(size=m))
    ea = np.append(ea, np.uint64(3)); eb = np.append(eb, np.uint64(3)); ej = np.append(ej, 0.75)   # self loop: a constant
    biases = rng.normal(size=n) * 0.4
    betas = np.linspace(0.2, 1.5, 12)
    e, st, eps = oracle.gen_run(ea, eb, ej, n, 77, betas, biases=biases, per_step=True)
    assert abs(e - oracle.energy(ea, eb, ej, n, st, biases)) < 1e-9 and eps[-1] == e
    e1, s1 = oracle.gen_run(ea, eb, ej, n, 77, betas[:5], biases=biases)
    e2, s2 = oracle.gen_run(ea, eb, ej, n, 77, betas[5:], biases=biases, t0=5, state=s1, initial=s1)
    assert np.array_equal(s2, st) and e2 == e
    nc, colours, pos = oracle.gen_colouring(ea, eb, ej, n)
    assert all(colours[int(a)] != colours[int(b)] for a, b in zip(ea, eb) if a != b) and len(set(pos.tolist())) == n


def test_packed_engines_energy_per_replica(oracle):
    """Engines D and E: K1 for every replica of a partial last group (R = 37), with per-replica betas."""
    rng = np.random.default_rng(9)
    n, R = 90, 37
    ea, eb, deg = [], [], np.zeros(n, dtype=np.int64)
    while len(ea) < 150:                               # degree <= 6 (engine D's limit), multi-edges allowed
        a, b = rng.integers(0, n, 2)
        if a != b and deg[a] < 6 and deg[b] < 6:
            ea.append(a); eb.append(b); deg[a] += 1; deg[b] += 1
    ea, eb = np.array(ea, dtype=np.uint64), np.array(eb, dtype=np.uint64)
    seeds = oracle.make_seeds(4, R)
    beta_r = np.linspace(0.1, 1.2, R)
    ej_d = 1.5 * rng.choice([-1.0, 1.0], len(ea))
    e, st = oracle.pk_run(ea, eb, ej_d, n, seeds, 5, beta_replica=beta_r)   # engine D: one |J|
    for r in range(R):
        assert abs(e[r] - oracle.energy(ea, eb, ej_d, n, st[r])) < 1e-9
    ej_e = rng.normal(size=len(ea))
    biases = rng.normal(size=n) * 0.3
    e, st, eps = oracle.rj_run(ea, eb, ej_e, n, seeds, 5, beta_replica=beta_r, biases=biases, per_step=True)
    for r in range(R):
        want = oracle.energy(ea, eb, ej_e, n, st[r], biases)
        assert abs(e[r] - want) < 1e-6 * max(1.0, abs(want))     # the energy of the ROUNDED couplings
        assert eps[r, -1] == e[r]


def _blocked_err(x, blocks=30):
    x = np.asarray(x, dtype=np.float64)
    b = x[: len(x) // blocks * blocks].reshape(blocks, -1).mean(axis=1)
    return b.std(ddof=1) / np.sqrt(blocks)


def test_packed_engine_d_against_exact_enumeration(oracle, exact):
    """K2 for engine D (the spec c5's kernel implements): a 12-spin +-J graph of degree <= 6 at beta = 0.5, <E> and <|M|>
    over 32 replicas x 3000 sweeps within 4 sigma of brute force."""
    rng = np.random.default_rng(21)
    n = 12
    ea = np.array([i for i in range(n)] + [i for i in range(n)], dtype=np.uint64)
    eb = np.array([(i + 1) % n for i in range(n)] + [(i + 5) % n for i in range(n)], dtype=np.uint64)
    ej = rng.choice([-1.0, 1.0], len(ea))
    beta = 0.5
    en = exact.enumerate_graph(ea, eb, ej, n, beta)
    seeds = oracle.make_seeds(8, 32)
    T, burn = 3000, 200
    e, st, eps = oracle.pk_run(ea, eb, ej, n, seeds, T, betas=np.full(T, beta), per_step=True)
    per_replica = eps[:, burn:].mean(axis=1)
    err = per_replica.std(ddof=1) / np.sqrt(len(per_replica))
    assert abs(per_replica.mean() - en["E"]) < 4 * err, (per_replica.mean(), en["E"], err)
    # magnetisation from repeated short continuations
    mags = []
    state = st
    for block in range(40):
        _, state = oracle.pk_run(ea, eb, ej, n, seeds, 5, betas=np.full(5, beta), states=state, t0=T + 5 * block)
        mags.append(np.abs(2 * state[:32].sum(axis=1).astype(np.int64) - n).mean())
    err_m = np.std(mags, ddof=1) / np.sqrt(len(mags))
    assert abs(np.mean(mags) - en["absM"]) < 4 * err_m + 0.02, (np.mean(mags), en["absM"], err_m)
