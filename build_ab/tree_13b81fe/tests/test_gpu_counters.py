"""The timestep counter across 2^32 on every kernel family, and exact resume.

A timestep's random numbers are Philox outputs of (key, t, ...): the low 32 bits of t sit in counter word 0, bits 32..47 beside
the colour / call index.  A persistent ClassicIsing on a small lattice passes 2^32 timesteps in a
few hours; a kernel that dropped the high bits would silently repeat its random stream.  isingmc_states_set_timestep places the
counter just below 2^32, four timesteps take it across, and the result must equal the oracle started at the same t0.  The same
entry point gives exact resume: seeds + set_state + set_timestep continue a stopped run on the trajectory it would have followed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T0 = 2 ** 32 - 2
STEPS = 4
BETAS = np.array([0.3, 0.6, 0.45, 0.9])


def _lattice_check(capi, oracle, W, H, R, **kw):
    glass = kw.pop("glass", False)
    rng = np.random.default_rng(W + H)
    ids = np.arange(W * H, dtype=np.uint64).reshape(H, W)
    sx = rng.choice(np.array([-1.0, 1.0]), size=(H, W)) if glass else -np.ones((H, W))
    sy = rng.choice(np.array([-1.0, 1.0]), size=(H, W)) if glass else -np.ones((H, W))
    ea = np.ascontiguousarray(np.stack([ids, ids], axis=-1).reshape(-1))
    eb = np.ascontiguousarray(np.stack([np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1))
    ej = np.ascontiguousarray(np.stack([sx, sy], axis=-1).reshape(-1))
    field = kw.get("field", 0.0)
    g = capi.Graph(ea, eb, ej, biases=np.full(W * H, field) if field else None)
    assert g.kind == capi.KIND_LATTICE2D
    lat = oracle.Lat(W, H, 1.0, 0, (sx.ravel() > 0).astype(np.uint8) if glass else None, (sy.ravel() > 0).astype(np.uint8) if glass else None,
                     field=field)
    seeds = capi.make_seeds(3, R)
    for per_step in (False, True):
        st = capi.States(g, seeds)
        st.timestep = T0
        assert st.timestep == T0
        out = st.do_time_steps(STEPS, BETAS, per_step_energies=per_step)
        assert st.timestep == T0 + STEPS
        packed = st.packed()
        for r in (0, R - 1):
            ref = lat.init(seeds[r])
            want = []
            for k in range(STEPS):
                lat.sweep(ref, seeds[r], T0 + k, BETAS[k])
                want.append(lat.energy_mag(ref)[0])
            assert np.array_equal(packed[r], ref), (W, H, per_step, r)
            if per_step:
                assert out[r].tolist() == want


@pytest.mark.parametrize("W,H,R,kw", [
    (64, 16, 3, {}),                       # LDS-resident kernel
    (1024, 1024, 8, {}),                   # persistent strips
    (2048, 2048, 3, {}),                   # streaming kernels
    (2048, 1024, 3, dict(glass=True)),     # sign planes
    (256, 16, 3, dict(field=0.5)),         # multi-class kernel, LDS-resident
    (2048, 1024, 3, dict(field=0.5)),      # multi-class kernel, streaming
], ids=["resident", "strips", "streaming", "+-J", "field resident", "field streaming"])
def test_checkerboard_kernels_across_two_to_the_32(capi, oracle, W, H, R, kw):
    _lattice_check(capi, oracle, W, H, R, **dict(kw))


def _cubic(L, rng, gaussian):
    ids = np.arange(L ** 3, dtype=np.uint64).reshape(L, L, L)
    ea = np.ascontiguousarray(np.stack([ids] * 3, axis=-1).reshape(-1))
    eb = np.ascontiguousarray(np.stack([np.roll(ids, -1, axis=2), np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1))
    ej = rng.normal(size=ea.shape) if gaussian else -np.ones(ea.shape)
    return ea, eb, ej


@pytest.mark.parametrize("kind", ["packed one-degree", "packed general", "real 7 slots", "real 4 slots"])
def test_replica_packed_kernels_across_two_to_the_32(capi, oracle, exact, monkeypatch, kind):
    rng = np.random.default_rng(8)
    R = 64
    biases = None
    if kind.startswith("packed"):
        monkeypatch.setenv("ISINGMC_FORCE_PACKED", "1")
        monkeypatch.setenv("ISINGMC_DISABLE_REAL", "1")
        ea, eb, ej = _cubic(12, rng, False)
        if "general" in kind:
            keep = rng.random(len(ea)) > 0.1
            ea, eb, ej = ea[keep], eb[keep], ej[keep]
        nvars = 12 ** 3
        run = lambda states, t0, **k: oracle.pk_run(ea, eb, ej, nvars, seeds, STEPS, betas=BETAS, states=states, t0=t0, **k)
    else:
        monkeypatch.setenv("ISINGMC_FORCE_REAL", "1")
        if "7" in kind:
            ea, eb, ej = _cubic(12, rng, True)
            nvars = 12 ** 3
        else:
            ea, eb, _ = exact.square_lattice_edges(48, 40, 1.0)
            ej = rng.normal(size=len(ea))
            nvars = 48 * 40
        biases = rng.normal(size=nvars) * 0.3
        run = lambda states, t0, **k: oracle.rj_run(ea, eb, ej, nvars, seeds, STEPS, betas=BETAS, biases=biases, states=states, t0=t0, **k)
    g = capi.Graph(ea, eb, ej, nvars=nvars, biases=biases, force_general=True)
    seeds = capi.make_seeds(13, R)
    for per_step in (False, True):
        st = capi.States(g, seeds)
        start = st.states().astype(np.uint8)               # the random start (drawn at t = 0)
        st.timestep = T0
        out = st.do_time_steps(STEPS, BETAS, per_step_energies=per_step)
        e_ref, s_ref, eps_ref = run(start.copy(), T0, per_step=True)
        assert np.array_equal(st.states().astype(np.uint8), s_ref[:R]), (kind, per_step)
        assert np.array_equal(st.energies(), e_ref)
        if per_step:
            assert np.array_equal(out, eps_ref)


@pytest.mark.parametrize("size", ["resident", "streaming"])
def test_csr_kernels_across_two_to_the_32(capi, oracle, exact, monkeypatch, size):
    monkeypatch.setenv("ISINGMC_DISABLE_REAL", "1")          # (from two experiments on, a big real-coupling graph would take the packed path)
    rng = np.random.default_rng(2)
    W, H = (20, 12) if size == "resident" else (700, 300)
    ea, eb, _ = exact.square_lattice_edges(W, H, 1.0)
    ej = rng.normal(size=len(ea))
    biases = rng.normal(size=W * H) * 0.2
    g = capi.Graph(ea, eb, ej, nvars=W * H, biases=biases, force_general=True)
    seeds = capi.make_seeds(4, 3)
    st = capi.States(g, seeds)
    start = st.states().astype(np.uint8)
    st.timestep = T0
    out = st.do_time_steps(STEPS, BETAS, per_step_energies=True)
    spins = st.states().astype(np.uint8)
    for r in range(3):
        e_ref, s_ref, eps_ref = oracle.gen_run(ea, eb, ej, W * H, seeds[r], BETAS, biases=biases, initial=start[r], t0=T0, per_step=True)
        assert np.array_equal(spins[r], s_ref)
        np.testing.assert_allclose(out[r], eps_ref, rtol=1e-12)


def test_exact_resume_from_seeds_state_and_counter(capi, exact):
    """A run stopped after 7 timesteps and rebuilt from (seeds, configurations, counter) continues bit for bit."""
    ea, eb, ej = exact.square_lattice_edges(512, 256, -1.0, np.random.default_rng(1))
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(99, 5)
    whole = capi.States(g, seeds)
    whole.do_time_steps(7, 0.5)
    saved_spins, saved_t = whole.states(), whole.timestep
    whole.do_time_steps(6, 0.7)
    resumed = capi.States(g, seeds, initial_state=None)
    for r in range(5):
        resumed.set_state(r, saved_spins[r])
    resumed.timestep = saved_t
    resumed.do_time_steps(6, 0.7)
    assert np.array_equal(resumed.packed(), whole.packed()) and np.array_equal(resumed.energies(), whole.energies())
    with pytest.raises(ValueError):
        resumed.timestep = 2 ** 48
