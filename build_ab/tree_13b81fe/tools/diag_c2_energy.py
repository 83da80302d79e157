#!/usr/bin/env python3
"""Diagnostic: the 4096^2 lattice away from beta_c against Kaufman's exact energy (observables rows 'c2 lattice')."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi

def edges(W, H):
    ids = np.arange(W * H, dtype=np.uint64).reshape(H, W)
    ea = np.stack([ids, ids], axis=-1).reshape(-1)
    eb = np.stack([np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1)
    return np.ascontiguousarray(ea), np.ascontiguousarray(eb), np.full(ea.shape, -1.0)

def host_energy(spins, W, H):
    s = spins.reshape(H, W).astype(np.int8) * 2 - 1
    return -float((s * np.roll(s, -1, 0)).sum(dtype=np.int64) + (s * np.roll(s, -1, 1)).sum(dtype=np.int64))

EXACT = {(4096, 0.35): -14760696.059890712, (4096, 0.55): -31056906.500247616, (2048, 0.35): -14760696.059890712 / 4, (2048, 0.55): -31056906.500247616 / 4,
         (1024, 0.35): -14760696.059890712 / 16, (1024, 0.55): -31056906.500247616 / 16}
for L in (1024, 2048, 4096):
    ea, eb, ej = edges(L, L)
    g = _capi.Graph(ea, eb, ej)
    for beta in (0.35, 0.55):
        for R, init in ((32, True), (32, False), (256, True)):
            if init is False and beta == 0.55:
                continue
            st = _capi.States(g, _capi.make_seeds(113, R), initial_state=np.ones(L * L, dtype=np.uint8) if init else None)
            e0 = st.energies()
            st.do_time_steps(300, beta)
            per = st.do_time_steps(200, beta, per_step_energies=True)
            e_end = st.energies()
            spins = st.states()
            k1 = host_energy(spins[0], L, L)
            sampled = []
            for _ in range(40):
                st.do_time_steps(5, beta)
                sampled.append(st.energies())
            sampled = np.array(sampled).mean(axis=0)
            pm = per.mean(axis=1)
            ex = EXACT[(L, beta)]
            print(f"L={L} beta={beta} R={R} init={init}: E0/N={e0[0] / L / L:+.3f} per-step mean {pm.mean():.1f} +- {pm.std(ddof=1) / np.sqrt(R):.1f} (z {(pm.mean() - ex) / (pm.std(ddof=1) / np.sqrt(R)):+.2f})"
                  f"  sampled energies() {sampled.mean():.1f} +- {sampled.std(ddof=1) / np.sqrt(R):.1f} (z {(sampled.mean() - ex) / (sampled.std(ddof=1) / np.sqrt(R)):+.2f})"
                  f"  last per-step == energies(): {bool(np.array_equal(per[:, -1], e_end))}  K1 replica 0: {k1 == e_end[0]}", flush=True)
