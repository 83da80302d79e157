"""Randomised parity (hypothesis, derandomised): random graphs / lattices / parameters on the GPU against
the oracle engines, bit for bit.  Covers ragged inputs the hand-written cases miss: duplicate edges, self
loops, isolated sites, odd replica counts, zero-length runs, extreme betas."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu
COMMON = dict(deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
SCALE = int(os.environ.get("ISINGMC_HYP_SCALE", "1"))  # a longer differential campaign: ISINGMC_HYP_SCALE=20 pytest tests/test_gpu_random.py


@st.composite
def random_graph(draw, uniform):
    n = draw(st.integers(2, 90))
    m = draw(st.integers(1, 3 * n))
    rng = np.random.default_rng(draw(st.integers(0, 2 ** 32 - 1)))
    ea = rng.integers(0, n, m).astype(np.uint64)
    eb = rng.integers(0, n, m).astype(np.uint64)
    if uniform:  # one |J|, degree <= 6, no self loops needed but allowed (they only shift the energy)
        keep, deg = [], np.zeros(n, dtype=int)
        for k in range(m):
            a, b = int(ea[k]), int(eb[k])
            if a == b or (deg[a] < 6 and deg[b] < 6):
                keep.append(k)
                if a != b:
                    deg[a] += 1; deg[b] += 1
        if not any(ea[k] != eb[k] for k in keep):
            keep = [0]; ea[0], eb[0] = 0, 1
        ea, eb = ea[keep], eb[keep]
        ej = rng.choice([-1.0, 1.0], len(ea)) * draw(st.sampled_from([1.0, 0.5, 2.25]))
    else:
        ej = np.round(rng.normal(size=m), 3)
    nvars = int(max(ea.max(), eb.max())) + 1 + draw(st.integers(0, 3))   # trailing isolated sites
    return ea, eb, ej, nvars


@settings(max_examples=25 * SCALE, **COMMON)
@given(g=random_graph(uniform=False), R=st.integers(1, 11), T=st.integers(0, 6),
       beta=st.sampled_from([0.0, 0.05, 0.4, 1.3, 7.0, -0.2]), with_bias=st.booleans(), seed=st.integers(0, 2 ** 64 - 1))
def test_random_general_graphs(capi, oracle, g, R, T, beta, with_bias, seed):
    ea, eb, ej, nvars = g
    biases = np.round(np.random.default_rng(seed % 2 ** 32).normal(size=nvars), 2) if with_bias else None
    seeds = capi.make_seeds(seed, R)
    graph = capi.Graph(ea, eb, ej, nvars=nvars, biases=biases)
    states = capi.States(graph, seeds)
    states.do_time_steps(T, beta)
    spins, energies = states.states(), states.energies()
    for r in range(R):
        e_ref, s_ref = oracle.gen_run(ea, eb, ej, nvars, seeds[r], [beta] * T, biases=biases)
        assert np.array_equal(spins[r].astype(np.uint8), s_ref)
        assert abs(energies[r] - e_ref) <= 1e-9 * max(1.0, abs(e_ref))


@settings(max_examples=25 * SCALE, **COMMON)
@given(g=random_graph(uniform=True), R=st.integers(1, 70), T=st.integers(0, 5),
       beta=st.sampled_from([0.0, 0.1, 0.5, 2.0, 30.0]), per_replica=st.booleans(), seed=st.integers(0, 2 ** 64 - 1))
def test_random_packed_graphs(capi, oracle, monkeypatch, g, R, T, beta, per_replica, seed):
    monkeypatch.setenv("ISINGMC_FORCE_PACKED", "1")
    ea, eb, ej, nvars = g
    seeds = capi.make_seeds(seed, R)
    graph = capi.Graph(ea, eb, ej, nvars=nvars)
    states = capi.States(graph, seeds)
    if per_replica:
        betas = np.linspace(0.0, 2.0, R)
        states.set_betas(betas)
        states.do_time_steps(T)
        e_ref, s_ref = oracle.pk_run(ea, eb, ej, nvars, seeds, T, beta_replica=betas)
    else:
        states.do_time_steps(T, beta)
        e_ref, s_ref = oracle.pk_run(ea, eb, ej, nvars, seeds, T, betas=[beta] * T)
    assert np.array_equal(states.states().astype(np.uint8), s_ref[:R])
    assert np.allclose(states.energies(), e_ref, rtol=1e-12, atol=1e-9)


@st.composite
def random_real_graph(draw):
    """any couplings / biases, degree <= 15: what the real-coupling packed path takes (duplicate bonds and self-loops included)"""
    n = draw(st.integers(2, 90))
    m = draw(st.integers(1, 3 * n))
    rng = np.random.default_rng(draw(st.integers(0, 2 ** 32 - 1)))
    ea = rng.integers(0, n, m).astype(np.uint64)
    eb = rng.integers(0, n, m).astype(np.uint64)
    maxdeg = draw(st.sampled_from([3, 4, 7, 10, 15]))
    keep, deg = [], np.zeros(n, dtype=int)
    for k in range(m):
        a, b = int(ea[k]), int(eb[k])
        if a == b or (deg[a] < maxdeg and deg[b] < maxdeg):
            keep.append(k)
            if a != b:
                deg[a] += 1; deg[b] += 1
    if not any(ea[k] != eb[k] for k in keep):
        keep = [0]; ea[0], eb[0] = 0, 1
    ea, eb = ea[keep], eb[keep]
    kind = draw(st.sampled_from(["gauss", "gauss_small", "integers", "halves"]))
    ej = {"gauss": rng.normal(size=len(ea)), "gauss_small": rng.normal(size=len(ea)) * 1e-3,
          "integers": rng.integers(1, 4, len(ea)) * rng.choice([-1.0, 1.0], len(ea)),
          "halves": rng.integers(1, 6, len(ea)) * 0.5 * rng.choice([-1.0, 1.0], len(ea))}[kind]
    nvars = int(max(ea.max(), eb.max())) + 1 + draw(st.integers(0, 3))
    scale = float(np.abs(ej).mean())
    biases = {0: None, 1: rng.normal(size=nvars) * scale, 2: np.full(nvars, 0.7 * scale)}[draw(st.integers(0, 2))]
    return ea, eb, ej, nvars, biases


@settings(max_examples=30 * SCALE, **COMMON)
@given(g=random_real_graph(), R=st.integers(1, 70), T=st.integers(0, 5), beta=st.sampled_from([0.0, 0.1, 0.5, 2.0, 30.0, -0.4]),
       per_replica=st.booleans(), per_step=st.booleans(), seed=st.integers(0, 2 ** 64 - 1))
def test_random_real_coupling_graphs(capi, oracle, g, R, T, beta, per_replica, per_step, seed):
    """The replica-packed real-coupling path against oracle engine E: spins, energies (exact integer sums
    scaled by 2^k: bit-equal), energies after every timestep."""
    ea, eb, ej, nvars, biases = g
    if not oracle.rj_eligible(ea, eb, ej, nvars, biases):
        return
    os.environ["ISINGMC_FORCE_REAL"] = "1"
    try:
        scale = float(np.abs(ej).mean())
        seeds = capi.make_seeds(seed, R)
        graph = capi.Graph(ea, eb, ej, nvars=nvars, biases=biases)
        if graph.info.real_slots == 0:   # one |J| and no biases: the bit-sliced packed path keeps such a graph
            assert biases is None and len(set(np.abs(ej[ea != eb]))) == 1
            return
        assert graph.info.real_slots in (4, 7, 11, 15)
        states = capi.States(graph, seeds)
        if per_replica:
            betas = np.linspace(0.0, 2.0, R) / scale
            states.set_betas(betas)
            eps = states.do_time_steps(T, per_step_energies=per_step)
            e_ref, s_ref, eps_ref = oracle.rj_run(ea, eb, ej, nvars, seeds, T, beta_replica=betas, biases=biases, per_step=True)
        else:
            eps = states.do_time_steps(T, beta / scale, per_step_energies=per_step)
            e_ref, s_ref, eps_ref = oracle.rj_run(ea, eb, ej, nvars, seeds, T, betas=[beta / scale] * T, biases=biases, per_step=True)
        assert np.array_equal(states.states().astype(np.uint8), s_ref[:R])
        assert np.array_equal(states.energies(), e_ref)
        if per_step:
            assert np.array_equal(eps, eps_ref)
    finally:
        os.environ.pop("ISINGMC_FORCE_REAL", None)


@settings(max_examples=20 * SCALE, **COMMON)
@given(wq=st.integers(1, 6), H=st.sampled_from([2, 4, 6, 8, 12, 16, 34]), pm=st.booleans(), R=st.integers(1, 5),
       T=st.integers(0, 5), beta=st.sampled_from([0.0, 0.2, 0.4407, 0.9, 4.0]), jabs=st.sampled_from([1.0, 0.3]),
       seed=st.integers(0, 2 ** 64 - 1))
def test_random_lattices(capi, oracle, exact, wq, H, pm, R, T, beta, jabs, seed):
    W = 64 * wq
    if (H * wq) % 4 or H < 4:
        H = 4 * H                                   # fast-path geometry: H * W/64 multiple of 4, H >= 4
    ea, eb, ej = exact.square_lattice_edges(W, H, jabs if pm else -jabs, np.random.default_rng(seed % 2 ** 32) if pm else None)
    graph = capi.Graph(ea, eb, ej)
    assert graph.kind == capi.KIND_LATTICE2D
    seeds = capi.make_seeds(seed, R)
    states = capi.States(graph, seeds)
    states.do_time_steps(T, beta)
    if pm:
        lat = oracle.Lat(W, H, jabs, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8))
    else:
        lat = oracle.Lat(W, H, jabs, 0)
    packed = states.packed()
    for r in range(R):
        ref = lat.init(seeds[r])
        for t in range(T):
            lat.sweep(ref, seeds[r], t, beta)
        assert np.array_equal(packed[r], ref)


@settings(max_examples=25 * SCALE, **COMMON)
@given(wq=st.sampled_from([4, 8, 16, 32]), rows=st.sampled_from([2, 4, 6, 16]), pm=st.booleans(), R=st.integers(1, 9),
       T=st.integers(2, 6), beta=st.sampled_from([0.0, 0.3, 0.4407, 1.5]), nw=st.sampled_from(["1", "4"]),
       per_step=st.booleans(), seed=st.integers(0, 2 ** 64 - 1))
def test_random_strip_lattices(capi, oracle, exact, wq, rows, pm, R, T, beta, nw, per_step, seed):
    """The persistent strip kernel on random geometries: W = 256 .. 2048, H = a few strips, 1 or 4 waves per strip."""
    W = 64 * wq
    S = 64 * int(nw) // (wq // 4)                    # rows per strip
    H = S * (rows if rows * S >= 4 else 4)
    if H % 2 or H < 4:
        H *= 2
    os.environ["ISINGMC_STRIP"], os.environ["ISINGMC_STRIP_NW"], os.environ["ISINGMC_DISABLE_RESIDENT"] = "1", nw, "1"
    try:
        ea, eb, ej = exact.square_lattice_edges(W, H, 1.0 if pm else -1.0, np.random.default_rng(seed % 2 ** 32) if pm else None)
        graph = capi.Graph(ea, eb, ej)
        seeds = capi.make_seeds(seed, R)
        states = capi.States(graph, seeds)
        eps = states.do_time_steps(T, beta, per_step_energies=per_step)
        lat = (oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8)) if pm
               else oracle.Lat(W, H, 1.0, 0))
        packed = states.packed()
        for r in sorted({0, R - 1}):
            ref = lat.init(seeds[r])
            for t in range(T):
                lat.sweep(ref, seeds[r], t, beta)
                if per_step:
                    assert eps[r, t] == lat.energy_mag(ref)[0]
            assert np.array_equal(packed[r], ref)
    finally:
        for k in ("ISINGMC_STRIP", "ISINGMC_STRIP_NW", "ISINGMC_DISABLE_RESIDENT"):
            os.environ.pop(k, None)


@settings(max_examples=40 * SCALE, **COMMON)
@given(wq=st.sampled_from([4, 8, 12]), H=st.sampled_from([4, 6, 16, 34]), pm=st.booleans(), R=st.integers(1, 4), T=st.integers(0, 5),
       beta=st.sampled_from([0.0, 0.2, 0.4407, 0.9, 4.0, -0.3]), jabs=st.sampled_from([1.0, 0.3]),
       mode=st.sampled_from(["field+", "field-", "field_max", "open_x", "open_y", "open_xy", "aniso_x", "aniso_y", "open_x_field",
                             "open_xy_field-", "signs", "open_y_signs"]),
       seed=st.integers(0, 2 ** 64 - 1))
def test_random_field_and_open_lattices(capi, oracle, exact, wq, H, pm, R, T, beta, jabs, mode, seed):
    """Multi-class checkerboard kernels on random geometries, couplings, fields, boundary conditions and anisotropies."""
    W = 64 * wq
    ea, eb, ej = exact.square_lattice_edges(W, H, jabs if pm else -jabs, np.random.default_rng(seed % 2 ** 32) if pm else None)
    jy = {"aniso_x": 0.45 * jabs, "aniso_y": 2.75 * jabs}.get(mode)         # |J| of the vertical bonds, when it differs
    if jy is not None:
        ej = ej.copy()
        ej[1::2] *= jy / jabs
    h = {"field+": 0.37 * jabs, "field-": -1.3 * jabs, "field_max": 2.0 * jabs, "open_x_field": 0.6 * jabs, "open_xy_field-": -1.0 * jabs,
         "signs": 1.7 * jabs, "open_y_signs": 0.45 * jabs}.get(mode, 0.0)
    ox, oy = mode.startswith(("open_x", "open_xy")), mode.startswith(("open_y", "open_xy"))
    signs = mode.endswith("signs")                                     # fields +-h from site to site
    keep = np.ones(len(ea), dtype=bool)
    if ox:
        keep &= ~((ea % W == W - 1) & (eb % W == 0))
    if oy:
        keep &= ~((ea // W == H - 1) & (eb // W == 0))
    biases = np.full(W * H, h) if h else None
    if signs:
        biases = h * np.random.default_rng(seed % 2 ** 31).choice([-1.0, 1.0], W * H)
        biases[:2] = [h, -h]                                           # both signs for sure
    graph = capi.Graph(ea[keep], eb[keep], ej[keep], nvars=W * H, biases=biases)
    assert graph.kind == capi.KIND_LATTICE2D
    assert graph.info.fast_path == ((4 if ox or oy else 1) if h else 3 if jy is not None else 2) and bool(graph.info.field_signs) == signs
    kw = dict(field=h, open_x=ox, open_y=oy, jabs_y=jy, field_neg=(biases < 0).astype(np.uint8) if signs else None)
    lat = (oracle.Lat(W, H, jabs, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8), **kw) if pm
           else oracle.Lat(W, H, jabs, 0, **kw))
    seeds = capi.make_seeds(seed, R)
    states = capi.States(graph, seeds)
    states.do_time_steps(T, beta)
    packed, energies = states.packed(), states.energies()
    for r in range(R):
        ref = lat.init(seeds[r])
        for t in range(T):
            lat.sweep(ref, seeds[r], t, beta)
        assert np.array_equal(packed[r], ref)
        assert energies[r] == lat.energy_mag(ref)[0]
