#!/usr/bin/env python3
"""Diagnostic (lives under tests/ because it calls the oracle; run by hand: python tests/diag_c2_parity.py [L]): 4096^2 states and per-step energies against oracle engine B."""
import os, sys, time
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

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 3
ea, eb, ej = edges(L, L)
g = _capi.Graph(ea, eb, ej)
lat = O.Lat(L, L)
for R, per_step, beta in ((2, False, 0.55), (2, True, 0.55), (256, False, 0.55), (256, True, 0.55), (256, True, 0.35)):
    seeds = _capi.make_seeds(5, R)
    st = _capi.States(g, seeds, initial_state=np.ones(L * L, dtype=np.uint8))
    out = st.do_time_steps(T, beta, per_step_energies=per_step)
    e_end = st.energies()
    spins = st.states()
    for r in sorted(set((0, R - 1, R // 2 - 1, R // 2))):
        ost = lat.pack(np.ones(L * L, dtype=np.uint8))
        oes = []
        for t in range(T):
            lat.sweep(ost, seeds[r], t, beta)
            oes.append(lat.energy_mag(ost)[0])
        ospins = lat.unpack(ost)
        diff = int((ospins != spins[r]).sum())
        msg = f"L={L} R={R} per_step={per_step} beta={beta} replica {r}: spins differing {diff}; energies() {e_end[r]:.0f} oracle {oes[-1]:.0f}"
        if per_step:
            msg += f"; per-step {out[r].tolist()} oracle {oes}"
        print(msg, flush=True)
