"""Regression tests for the device-block cache (isingmc.hip cached_malloc / cached_free, round 3) and the synchronisations that
went in with commit b935561: hipFree used to wait for the whole device; a recycled block may be handed to the next request at
once, so every owner must have drained the work that still uses it.  One scenario per recycling site (tests/_block_cache_scenarios.py):
reserve() behind enqueue-only tempering sweeps, the strip kernel's halo regrown behind a running strip launch, pk_append opening
a group behind enqueued exchange rounds, the real-coupling path's scale table, the sampling slabs, and two host threads creating
and destroying containers of equal sizes (the device fan-out of lattice.rs:192-197).  Every scenario must give the same bits with
the cache on (this process) and off (a child with ISINGMC_NO_ALLOC_CACHE=1), and where an oracle engine covers it, the oracle's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import _block_cache_scenarios as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def without_cache():
    """digests of every scenario from a child process that runs with hipMalloc / hipFree (one child: one torch-free import)"""
    env = dict(os.environ, ISINGMC_NO_ALLOC_CACHE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_block_cache_scenarios.py")], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("name", sorted(S.SCENARIOS))
def test_scenario_gives_the_same_bits_with_and_without_the_block_cache(capi, without_cache, name):
    for attempt in range(2):  # twice in this process: the second pass starts from a cache filled by the first
        got = S.digests(S.SCENARIOS[name](capi))
        assert got == without_cache[name], (name, attempt, [k for k in got if got[k] != without_cache[name][k]])


def test_reserve_behind_enqueued_sweeps_against_the_oracle(capi, oracle):
    W, H, R, T = 512, 256, 8, 300
    res = S.reserve_after_enqueued_tempering(capi, W, H, R, T)
    seeds = capi.make_seeds(21, 2 * R + 1)
    betas = np.linspace(0.2, 0.9, R)
    lat = oracle.Lat(W, H)
    for r in (0, R - 1):                                   # no exchange round was run: slot r kept rung r's beta
        ref = lat.init(int(seeds[r]))
        for t in range(T):
            lat.sweep(ref, int(seeds[r]), t, float(betas[r]))
        assert np.array_equal(res["a"][r], ref) and res["a_e"][r] == lat.energy_mag(ref)[0]
    assert np.array_equal(res["a"][R], lat.init(int(seeds[2 * R])))          # the appended replica: its random start
    for r in (0, R - 1):
        ref = lat.init(int(seeds[R + r]))
        for t in range(3):
            lat.sweep(ref, int(seeds[R + r]), t, 0.5)
        assert np.array_equal(res["b"][r], ref)


def test_real_path_tempering_then_new_betas_against_engine_e(capi, oracle):
    W, H, R, T = 160, 128, 40, 12
    res = S.real_path_set_betas_after_enqueued_tempering(capi, W, H, R, T)
    ea, eb, _ = S._lattice(W, H, 1.0)
    ej = np.random.default_rng(6).normal(size=len(ea))
    seeds = capi.make_seeds(24, R)
    ladder = np.linspace(0.3, 1.5, R)
    perm = np.arange(R, dtype=np.uint32)
    st, t = None, 0
    for rnd in range(4):
        beta_of_slot = np.empty(R)
        beta_of_slot[perm] = ladder
        e, st = oracle.rj_run(ea, eb, ej, W * H, seeds, T // 4, beta_replica=beta_of_slot, states=st, t0=t)
        t += T // 4
        capi.pt_swap_round(3, rnd, ladder, e, perm)
    e, st = oracle.rj_run(ea, eb, ej, W * H, seeds, 3, beta_replica=np.linspace(1.5, 0.3, R), states=st, t0=t)
    assert np.array_equal(res["a"].astype(np.uint8), st[:R]) and np.array_equal(res["a_e"], e)
