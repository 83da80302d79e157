"""Host-side logic of libisingmc.so through the C ABI -- no GPU needed: seeds, annealing schedules,
lattice recogniser, colouring, tempering swap step, ABI surface."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(capi):
    header = open(os.path.join(ROOT, "include", "isingmc.h")).read()
    declared = set(re.findall(r"\b(isingmc_[a-z0-9_]+)\s*\(", header))
    declared -= {"isingmc_graph_info_t"}
    assert declared, "no declarations parsed"
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libisingmc.so does not export {name}"
    assert declared == set(capi.EXPORTED_SYMBOLS), declared ^ set(capi.EXPORTED_SYMBOLS)
    assert L.isingmc_abi_version() == 3


def test_no_cpu_fallback(capi, exact):
    """Without a HIP device every device entry point must fail loudly (never compute on the CPU)."""
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    ea, eb, ej = exact.square_lattice_edges(64, 4, -1.0)
    with pytest.raises(RuntimeError, match="no HIP device|no ROCm"):
        capi.Graph(ea, eb, ej)


def test_make_seeds_matches_oracle(capi, oracle):
    for seed in (0, 1, 1234, 2 ** 64 - 1):
        assert np.array_equal(capi.make_seeds(seed, 9), oracle.make_seeds(seed, 9))
    a, b = capi.make_seeds(None, 4), capi.make_seeds(None, 4)  # entropy
    assert not np.array_equal(a, b)


def _schedule_restatement(stops, T, compat):
    """lattice.rs:320-334 + 358-365 restated in Python (intended: i = step; compat: i = captured)."""
    betas = sorted(stops, key=lambda s: s[0])
    if not betas:
        betas = [(0, 1.0), (T, 1.0)]
    if betas[0][0] > 0:
        betas.insert(0, (0, betas[0][1]))
    i_captured = betas[-1][0]
    if betas[-1][0] < T:
        betas.append((T, betas[-1][1]))
    out, idx = [], 0
    for step in range(T):
        i = i_captured if compat else step
        while idx + 2 < len(betas) and i > betas[idx + 1][0]:
            idx += 1
        (ia, va), (ib, vb) = betas[idx], betas[idx + 1]
        out.append((vb - va) * (0.0 if ib == ia else (i - ia) / (ib - ia)) + va)
    return np.array(out)


@pytest.mark.parametrize("stops,T", [([], 10), ([(0, 0.1), (10, 2.0)], 10), ([(5, 1.0)], 12),
                                     ([(8, 2.0), (2, 0.5), (4, 1.0)], 10), ([(0, 0.2), (100, 3.0)], 7),
                                     ([(3, 0.5), (3, 0.9), (6, 1.0)], 9)])
def test_expand_schedule(capi, stops, T):
    for compat in (False, True):
        got = capi.expand_schedule(stops, T, compat)
        np.testing.assert_allclose(got, _schedule_restatement(list(stops), T, compat), rtol=0, atol=1e-15)
    if stops:
        # the reference's behaviour: beta constant = the last stop's beta
        last = sorted(stops, key=lambda s: s[0])[-1][1]
        compat = capi.expand_schedule(stops, T, True)
        assert np.allclose(compat, compat[0])
        if sorted(stops, key=lambda s: s[0])[-1][0] <= T:
            assert np.isclose(compat[0], last)


def test_expand_schedule_geometric_c4(capi):
    """BASELINE config c4: beta_t = 0.1 * 30^(t/(T-1)) given as T stops is reproduced step by step."""
    T = 50
    stops = [(t, 0.1 * 30 ** (t / (T - 1))) for t in range(T)]
    np.testing.assert_allclose(capi.expand_schedule(stops, T), [b for _, b in stops], rtol=1e-15)
    with pytest.raises(ValueError):
        capi.expand_schedule([(0, float("nan"))], 4)


def test_recogniser_accepts_any_edge_order_and_orientation(capi, exact):
    rng = np.random.default_rng(0)
    for (W, H) in [(64, 4), (128, 6), (8, 12), (256, 4)]:
        ea, eb, ej = exact.square_lattice_edges(W, H, -1.5)
        perm = rng.permutation(len(ea))
        flip = rng.integers(0, 2, len(ea)).astype(bool)
        a, b = np.where(flip, eb, ea)[perm], np.where(flip, ea, eb)[perm]
        r = capi.recognise_lattice2d(a, b, ej[perm], W * H)
        assert r == dict(is_lattice=True, width=W, height=H, jabs=1.5, uniform_sign=True)
    ea, eb, ej = exact.square_lattice_edges(64, 8, 1.0, np.random.default_rng(1))
    r = capi.recognise_lattice2d(ea, eb, ej, 512)
    assert r["is_lattice"] and not r["uniform_sign"] and r["jabs"] == 1.0


def test_recogniser_accepts_one_coupling_strength_per_direction(capi, exact):
    W, H = 64, 8
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0, np.random.default_rng(2))
    ej = ej.copy(); ej[0::2] *= 0.5; ej[1::2] *= 2.5              # right bonds |J| = 0.5, down bonds |J| = 2.5
    perm = np.random.default_rng(0).permutation(len(ea))
    r = capi.recognise_lattice2d(eb[perm], ea[perm], ej[perm], W * H)
    assert r["is_lattice"] and r["anisotropic"] and r["jabs"] == 0.5 and not r["uniform_sign"]
    ej2 = ej.copy(); ej2[6] = 2.5                                    # a horizontal bond with the vertical bonds' |J|
    assert not capi.recognise_lattice2d(ea, eb, ej2, W * H)["is_lattice"]
    ea1, eb1, ej1 = exact.square_lattice_edges(W, H, 1.0)
    assert "anisotropic" not in capi.recognise_lattice2d(ea1, eb1, ej1, W * H)


def test_recogniser_rejects_non_lattices(capi, exact):
    W, H = 64, 8
    ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
    N = W * H
    assert not capi.recognise_lattice2d(ea[:-1], eb[:-1], ej[:-1], N)["is_lattice"]          # missing bond
    ej2 = ej.copy(); ej2[5] = -2.0
    assert not capi.recognise_lattice2d(ea, eb, ej2, N)["is_lattice"]                        # non-uniform |J|
    eb2 = eb.copy(); eb2[7] = (eb2[7] + 3) % N
    assert not capi.recognise_lattice2d(ea, eb2, ej, N)["is_lattice"]                        # rewired bond
    ea3, eb3 = ea.copy(), eb.copy(); ea3[0], eb3[0] = ea3[2], eb3[2]
    assert not capi.recognise_lattice2d(ea3, eb3, ej, N)["is_lattice"]                       # duplicate bond
    assert not capi.recognise_lattice2d(ea, eb, ej, N + 1)["is_lattice"]                     # extra isolated site
    open_mask = ~((ea % W == W - 1) & (eb % W == 0))                                          # open boundary in x: a lattice
    r = capi.recognise_lattice2d(ea[open_mask], eb[open_mask], ej[open_mask], N)
    assert r["is_lattice"] and r["open_x"] and not r["open_y"] and (r["width"], r["height"]) == (W, H)
    half_open = open_mask.copy(); half_open[np.flatnonzero(~open_mask)[:3]] = True           # only some wrap-around bonds cut
    assert not capi.recognise_lattice2d(ea[half_open], eb[half_open], ej[half_open], N)["is_lattice"]
    ea5, eb5, ej5 = exact.cubic_lattice_edges(4, -1.0)
    assert not capi.recognise_lattice2d(ea5, eb5, ej5, 64)["is_lattice"]
    with pytest.raises(ValueError, match="Must supply some edges"):
        capi.recognise_lattice2d(np.zeros(0), np.zeros(0), np.zeros(0), 4)
    with pytest.raises(ValueError, match="Index out of bounds"):
        capi.recognise_lattice2d([0, 9], [1, 2], [1.0, 1.0], 4)


def test_colouring_is_proper_and_matches_oracle(capi, oracle, exact):
    rng = np.random.default_rng(4)
    cases = [exact.square_lattice_edges(6, 4, -1.0)[:2] + (24,), exact.cubic_lattice_edges(3, -1.0)[:2] + (27,)]
    n, m = 200, 700
    cases.append((rng.integers(0, n, m).astype(np.uint64), rng.integers(0, n, m).astype(np.uint64), n + 5))
    for ea, eb, nvars in cases:
        nc, colours = capi.colour_graph(ea, eb, nvars)
        onc, ocolours, _ = oracle.gen_colouring(ea, eb, np.ones(len(ea)), nvars)
        assert nc == onc and np.array_equal(colours, ocolours)
        proper = ea != eb
        assert np.all(colours[ea[proper].astype(int)] != colours[eb[proper].astype(int)])
    nc, colours = capi.colour_graph(*exact.square_lattice_edges(6, 4, -1.0)[:2], 24)
    assert nc == 2  # bipartite lattice -> checkerboard


def test_pt_swap_round_matches_oracle(capi, oracle):
    rng = np.random.default_rng(9)
    betas = np.linspace(0.1, 1.0, 11)
    perm_a = np.arange(11, dtype=np.uint32)
    perm_b = perm_a.copy()
    for rnd in range(40):
        e = rng.normal(size=11) * 3
        sa = capi.pt_swap_round(12345, rnd, betas, e, perm_a)
        sb = oracle.pt_swap_round(12345, rnd, betas, e, perm_b)
        assert sa == sb and np.array_equal(perm_a, perm_b)
    assert not np.array_equal(perm_a, np.arange(11))
    with pytest.raises(ValueError):
        capi.pt_swap_round(1, 0, betas, np.zeros(11), np.full(11, 99, dtype=np.uint32))


def test_shard_bounds():
    from pyisingmontecarlo_amd.distributed import block_size, shard_bounds
    for n in (0, 1, 7, 8, 100, 256, 513, 1024):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            per = block_size(n, world)
            assert per >= -(-n // world) and all(0 <= hi - lo <= per for lo, hi in blocks)
            if n // world >= 32:                       # big blocks start on multiples of 32 (packed-path groups)
                assert all(lo % 32 == 0 for lo, hi in blocks if hi > lo)
    assert shard_bounds(256, 8, 3) == (96, 128) and shard_bounds(100, 2, 1) == (64, 100)
