"""The persistent strip kernel (csrc/strip_kernels.hpp): all timesteps of a call in one launch, neighbour strips
exchanging one row per half-sweep through 8-byte {tag, word} granules.  Same Philox counters as the per-colour
launches, so everything must equal the oracle (and the streaming path) bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEEDS3 = np.array([0x0123456789ABCDEF, 42, 2**64 - 1], dtype=np.uint64)


def _oracle_lat(oracle, W, H, ej, glass):
    if glass:
        return oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8))
    return oracle.Lat(W, H, 1.0, 0)


@pytest.mark.parametrize("W,H", [(1024, 128), (512, 1024), (2048, 64), (8192, 32), (256, 512), (4096, 64)])
@pytest.mark.parametrize("glass", [False, True])
@pytest.mark.parametrize("nw", ["1", "4"])
def test_strip_kernel_bit_exact(capi, oracle, exact, monkeypatch, W, H, glass, nw):
    monkeypatch.setenv("ISINGMC_STRIP", "1")
    monkeypatch.setenv("ISINGMC_STRIP_NW", nw)                     # strip = one wavefront (no barrier) / a 256-thread workgroup
    monkeypatch.setenv("ISINGMC_DISABLE_RESIDENT", "1")           # the small cases would otherwise stay in one workgroup
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0, np.random.default_rng(W + H) if glass else None)
    g = capi.Graph(ea, eb, ej)
    assert g.kind == capi.KIND_LATTICE2D
    lat = _oracle_lat(oracle, W, H, ej, glass)
    T = 5
    betas = np.linspace(0.3, 0.6, T)
    for seeds in (SEEDS3, capi.make_seeds(W, 8)):                  # 8 replicas: the XCD-aware block -> strip map
        st = capi.States(g, seeds)
        eps = st.do_time_steps(T, betas, per_step_energies=True)   # per-step thresholds + energies after every step
        st.do_time_steps(4, 0.4407)                                # constant beta, no measurement, continues at t = T
        packed, energies, mags = st.packed(), st.energies(), st.magnetisations()
        for r in (0, len(seeds) - 1):
            ref = lat.init(seeds[r])
            for t in range(T):
                lat.sweep(ref, seeds[r], t, betas[t])
                assert eps[r, t] == lat.energy_mag(ref)[0], (r, t)
            for t in range(T, T + 4):
                lat.sweep(ref, seeds[r], t, 0.4407)
            np.testing.assert_array_equal(packed[r], ref, err_msg=f"replica {r}")
            assert (energies[r], mags[r]) == lat.energy_mag(ref)


def test_strip_kernel_equals_streaming_path_c3_geometry(capi, exact, monkeypatch):
    """BASELINE c3's lattice (1024^2, 16 strips per replica), per-replica betas as on a tempering ladder: default path
    selection must pick the strip kernel here and give the bits of the per-colour launches."""
    L, R, T = 1024, 16, 12
    ea, eb, ej = exact.square_lattice_edges(L, L, -1.0)
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(1, R)
    out = []
    for mode in (None, "0"):
        if mode is None:
            monkeypatch.delenv("ISINGMC_STRIP", raising=False)
        else:
            monkeypatch.setenv("ISINGMC_STRIP", mode)
        st = capi.States(g, seeds)
        st.set_betas(np.linspace(0.1, 1.0, R))
        st.do_time_steps(T)
        st.set_betas(None)
        eps = st.do_time_steps(3, 0.44, per_step_energies=True)
        out.append((st.packed(), st.energies(), eps))
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)
    spins = capi.States(g, seeds[:1])
    spins.do_time_steps(2, 0.3)                                    # K1 on the strip path: energy from the returned spins
    s2 = spins.states().reshape(L, L).astype(np.int8) * 2 - 1
    e_host = -(s2 * np.roll(s2, -1, axis=1)).sum(dtype=np.int64) - (s2 * np.roll(s2, -1, axis=0)).sum(dtype=np.int64)
    assert spins.energies()[0] == float(e_host)


def test_strip_kernel_in_several_passes(capi, oracle, exact, monkeypatch):
    """More workgroups than may be resident at once (4 per CU): the replicas go through in blocks, one launch each."""
    monkeypatch.setenv("ISINGMC_STRIP", "1")
    W, H, R, T = 1024, 128, 700, 3                                 # 8 one-wave strips x 700 replicas = 5600 waves > 4096
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(8, R)
    st = capi.States(g, seeds)
    st.do_time_steps(T, 0.5)
    packed, energies = st.packed(), st.energies()
    lat = oracle.Lat(W, H)
    for r in (0, 349, 350, 511, 512, R - 1):
        ref = lat.init(seeds[r])
        for t in range(T):
            lat.sweep(ref, seeds[r], t, 0.5)
        np.testing.assert_array_equal(packed[r], ref, err_msg=f"replica {r}")
        assert energies[r] == lat.energy_mag(ref)[0]


def test_on_stream_tempering_on_the_strip_path(capi, exact, monkeypatch):
    """Exchange rounds with the energies measured inside the last strip launch of each block of sweeps (no separate
    pass over the planes) against the per-colour launches + lat_measure_kernel: same permutation, swaps, spins."""
    from pyisingmontecarlo_amd.tempering import ClassicalTempering
    W, H, G = 1024, 256, 12
    edges = exact.square_lattice_edges(W, H, -1.0)
    runs = []
    # strip kernel with the exchange rounds inside the launch (rung-indexed mailboxes) / strip kernel, one launch per round /
    # per-colour launches + lat_measure_kernel
    for mode, in_kernel in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("ISINGMC_STRIP", mode)
        monkeypatch.setenv("ISINGMC_PT_IN_KERNEL", in_kernel)
        pt = ClassicalTempering(edges, seed=31)
        for b in np.linspace(0.4400, 0.4402, G):            # 262 144 spins: neighbouring rungs must be this close to exchange
            pt.add_graph(float(b))
        pt.timesteps(4)
        pt.timesteps(43, replica_swap_freq=4)                      # 10 exchange rounds on the stream + 3 sweeps
        pt.timesteps(9, replica_swap_freq=3)                       # the round counter continues (odd / even pairing)
        runs.append((pt.get_permutation(), pt.get_total_swaps(), pt._states.packed(), pt._states.energies()))
    assert runs[0][1] == runs[1][1] == runs[2][1] > 0
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            np.testing.assert_array_equal(a, b)


def test_in_kernel_exchange_against_the_oracle_engine(capi, exact, monkeypatch):
    """the exchange rounds INSIDE the strip launch (lat_strip_kernel<.., LAD = true>: mailboxes, arrival
    counter, strip_det_exp, relabelling) against the host swap step driven by the CPU oracle engine (the loop of
    tempering.rs:177-194): permutation, swap count, spins and energies must be equal."""
    from helpers import OracleLatEngine
    from pyisingmontecarlo_amd.tempering import ClassicalTempering
    W, H, G = 1024, 128, 8                                          # two strips of 64 rows per replica
    edges = exact.square_lattice_edges(W, H, -1.0)
    monkeypatch.setenv("ISINGMC_STRIP", "1")
    monkeypatch.setenv("ISINGMC_PT_IN_KERNEL", "1")
    runs = []
    for factory in (None, lambda: OracleLatEngine(W, H)):
        pt = ClassicalTempering(edges, seed=77, engine_factory=factory)
        for b in np.linspace(0.4400, 0.4403, G):                   # 131 072 spins: rungs this close exchange
            pt.add_graph(float(b))
        pt.timesteps(3)
        pt.timesteps(26, replica_swap_freq=4)                      # 6 exchange rounds (5 inside the launch) + 2 sweeps
        pt.timesteps(9, replica_swap_freq=3)                       # odd / even pairing continues
        st = pt._states
        runs.append((pt.get_permutation(), pt.get_total_swaps(), st.states(), st.energies()))
    assert runs[0][1] == runs[1][1] > 0
    for a, b in zip(runs[0], runs[1]):
        np.testing.assert_array_equal(a, b)


def test_in_kernel_exchange_at_c3_shape(capi, oracle, exact, monkeypatch):
    """BASELINE c3's per-GPU share under pytest -- 1024^2 x 64 rungs = 1024 workgroups, the residency
    limit: exchange rounds inside the launch == one launch per round; swaps happen; K1 on two rungs."""
    from pyisingmontecarlo_amd.tempering import ClassicalTempering
    L, G = 1024, 64
    ea, eb, ej = exact.square_lattice_edges(L, L, -1.0)
    runs = []
    for in_kernel in ("1", "0"):
        monkeypatch.setenv("ISINGMC_PT_IN_KERNEL", in_kernel)
        pt = ClassicalTempering((ea, eb, ej), seed=5)
        for b in np.linspace(0.44000, 0.44030, G):
            pt.add_graph(float(b))
        pt.timesteps(2)
        pt.timesteps(40, replica_swap_freq=10)
        st = pt._states
        runs.append((pt.get_permutation(), pt.get_total_swaps(), st.packed(), st.energies()))
    assert runs[0][1] == runs[1][1] > 0
    for a, b in zip(runs[0], runs[1]):
        np.testing.assert_array_equal(a, b)
    states = st.states()
    for r in (0, 63):                                                              # K1: host recomputation of the energy
        assert runs[1][3][r] == oracle.energy(ea, eb, ej, L * L, states[r].astype(np.uint8))


def test_strip_timeout_is_recovered(capi, oracle, exact, monkeypatch):
    """when a strip launch gives up (its workgroups were not all resident) a synchronous call repeats its
    work from the planes it started with, on the per-colour launches, and the object stays usable.  In-order dispatch makes
    a real timeout need a co-tenant, so the hook raises the kernel's error word before the object's first strip launch."""
    W, H, R, T = 1024, 256, 12, 5
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(21, R)
    monkeypatch.setenv("ISINGMC_STRIP", "0")
    ref = capi.States(g, seeds)
    ref.do_time_steps(T, 0.45)
    ref_eps = ref.do_time_steps(3, 0.45, per_step_energies=True)
    ref_e, ref_s = ref.run_sampling(0.45, 2, 2, 2)
    monkeypatch.setenv("ISINGMC_STRIP", "1")
    monkeypatch.setenv("ISINGMC_STRIP_TEST_FAIL_ONCE", "1")
    st = capi.States(g, seeds)
    st.do_time_steps(T, 0.45)                        # "times out", is repeated, must not raise
    eps = st.do_time_steps(3, 0.45, per_step_energies=True)   # the object took the per-colour path for good
    np.testing.assert_array_equal(eps, ref_eps)
    e2, s2 = st.run_sampling(0.45, 2, 2, 2)
    np.testing.assert_array_equal(e2, ref_e)
    np.testing.assert_array_equal(s2, ref_s)
    np.testing.assert_array_equal(st.packed(), ref.packed())
    # the same hook under the sampling call's own guard
    st2 = capi.States(g, seeds)
    ref2 = capi.States(g, seeds)
    monkeypatch.setenv("ISINGMC_STRIP_TEST_FAIL_ONCE", "0")
    monkeypatch.setenv("ISINGMC_STRIP", "0")
    e_ref, s_ref = ref2.run_sampling(0.45, 4, 3, 2)
    monkeypatch.setenv("ISINGMC_STRIP", "1")
    monkeypatch.setenv("ISINGMC_STRIP_TEST_FAIL_ONCE", "1")
    e_got, s_got = st2.run_sampling(0.45, 4, 3, 2)
    np.testing.assert_array_equal(e_got, e_ref)
    np.testing.assert_array_equal(s_got, s_ref)
    lat = oracle.Lat(W, H)
    o = lat.init(seeds[7])
    for t in range(T + 3 + 2 + 4):
        lat.sweep(o, seeds[7], t, 0.45)
    np.testing.assert_array_equal(st.packed()[7], o)
