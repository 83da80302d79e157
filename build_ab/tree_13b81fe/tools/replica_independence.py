#!/usr/bin/env python3
"""Are the replicas independent?  For uncorrelated replicas the replica-mean of the energy fluctuates R times less than one
replica: ratio = R x Var_t(mean_r E) / mean_r Var_t(E_r) = 1 + (R - 1) rho for a mean pairwise correlation rho.  (Replica keys
are the 64-bit seeds the xoshiro stream hands out; Philox is keyed by them.)

    python tools/replica_independence.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402


def case(name, L, beta, R, therm, steps, force_general=False, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    ids = np.arange(L * L, dtype=np.uint64).reshape(L, L)
    ea = np.ascontiguousarray(np.stack([ids, ids], axis=-1).reshape(-1))
    eb = np.ascontiguousarray(np.stack([np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1))
    g = _capi.Graph(ea, eb, -np.ones(len(ea)), force_general=force_general)
    st = _capi.States(g, _capi.make_seeds(1, R))
    for k in (env or {}):
        del os.environ[k]
    st.do_time_steps(therm, beta)
    chunks = [st.do_time_steps(100, beta, per_step_energies=True) for _ in range(steps // 100)]
    e = np.concatenate(chunks, axis=1)                                  # [R, steps]
    # 16 time blocks give the error of the ratio
    ratios = []
    for blk in np.array_split(np.arange(e.shape[1]), 16):
        x = e[:, blk]
        ratios.append(R * x.mean(axis=0).var(ddof=1) / x.var(axis=1, ddof=1).mean())
    ratios = np.array(ratios)
    r, err = ratios.mean(), ratios.std(ddof=1) / 4.0
    print(f"{name:44s} L={L:5d} R={R:4d} steps={steps:7d}: ratio {r:.4f} +- {err:.4f}  ->  mean pairwise correlation {(r - 1) / (R - 1):+.1e} +- {err / (R - 1):.1e}", flush=True)


case("LDS-resident kernel", 64, 0.42, 256, 20000, 400000)
case("streaming kernels (c2 size)", 4096, 0.40, 256, 1500, 8000)
case("replica-packed bit-sliced (32 replicas per word)", 128, 0.42, 256, 20000, 100000, force_general=True, env={"ISINGMC_FORCE_PACKED": "1", "ISINGMC_DISABLE_REAL": "1"})
case("replica-packed real-coupling (32 per word)", 128, 0.42, 256, 20000, 100000, force_general=True, env={"ISINGMC_FORCE_REAL": "1"})
