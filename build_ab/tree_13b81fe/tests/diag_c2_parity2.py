#!/usr/bin/env python3
"""Diagnostic 2: where do per-step-energy runs leave the oracle's trajectory?  (lives under tests/ because it calls the oracle; run by hand: python tests/diag_c2_parity2.py L R [plain])"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi
from oracle import oracle as O

def edges(W, H):
    ids = np.arange(W * H, dtype=np.uint64).reshape(H, W)
    ea = np.stack([ids, ids], axis=-1).reshape(-1)
    eb = np.stack([np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1)
    return np.ascontiguousarray(ea), np.ascontiguousarray(eb), np.full(ea.shape, -1.0)

L, R, beta = int(sys.argv[1]), int(sys.argv[2]), 0.55
PER_STEP = not (len(sys.argv) > 3 and sys.argv[3] == 'plain')
ea, eb, ej = edges(L, L)
g = _capi.Graph(ea, eb, ej)
lat = O.Lat(L, L)
seeds = _capi.make_seeds(5, R)
for T in (2,):
    st = _capi.States(g, seeds, initial_state=np.ones(L * L, dtype=np.uint8))
    out = st.do_time_steps(T, beta, per_step_energies=PER_STEP)
    spins = st.states()
    bad = []
    for r in range(R):
        if r not in (0, 1, 2, 3, 64, R // 2 - 1, R // 2, R - 2, R - 1) and R > 8:
            continue
        ost = lat.pack(np.ones(L * L, dtype=np.uint8))
        for t in range(T):
            lat.sweep(ost, seeds[r], t, beta)
        d = np.nonzero(lat.unpack(ost) != spins[r])[0]
        ys, xs = d // L, d % L
        colours = ((xs + ys) & 1)
        print(f"[{os.environ.get('ISINGMC_DBG_FUSED', '-')}{'' if PER_STEP else ' plain'}] L={L} R={R} T={T} replica {r}: {len(d)} spins differ; colour-0 {int((colours == 0).sum())} colour-1 {int((colours == 1).sum())}"
              + (f"; rows {ys.min()}..{ys.max()} (distinct {len(set(ys.tolist()))}), cols {xs.min()}..{xs.max()}; first rows {sorted(set(ys.tolist()))[:12]}" if len(d) else ""), flush=True)
