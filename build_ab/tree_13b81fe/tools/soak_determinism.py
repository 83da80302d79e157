#!/usr/bin/env python3
"""Soak: every kernel family twice from the same seeds for thousands of timesteps at benchmark sizes; the two trajectories must
stay bit-identical (energies after every chunk, configurations at the end).  A race or a hazard that depends on timing shows as
a mismatch (the sweep+measure store bug of round 3 flipped a few thousand spins per launch, differently every run).

    python tools/soak_determinism.py [steps-scale]      (default 1.0: ~2 min on one MI355X)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402

SCALE = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0


def square(W, H, rng=None, gaussian=False):
    ids = np.arange(W * H, dtype=np.uint64).reshape(H, W)
    ea = np.stack([ids, ids], axis=-1).reshape(-1)
    eb = np.stack([np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1)
    if gaussian:
        ej = rng.normal(size=ea.shape)
    elif rng is not None:
        ej = rng.choice(np.array([-1.0, 1.0]), size=ea.shape)
    else:
        ej = -np.ones(ea.shape)
    return np.ascontiguousarray(ea), np.ascontiguousarray(eb), np.ascontiguousarray(ej)


def cubic(L):
    ids = np.arange(L ** 3, dtype=np.uint64).reshape(L, L, L)
    ea = np.stack([ids] * 3, axis=-1).reshape(-1)
    eb = np.stack([np.roll(ids, -1, axis=2), np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1)
    return np.ascontiguousarray(ea), np.ascontiguousarray(eb), -np.ones(ea.shape)


def soak(name, graph, R, steps, chunk, beta, per_step=False):
    steps = max(chunk, int(steps * SCALE) // chunk * chunk)
    seeds = _capi.make_seeds(1, R)
    a, b = _capi.States(graph, seeds), _capi.States(graph, seeds)
    t0 = time.time()
    bad = 0
    for k in range(steps // chunk):
        ea_ = a.do_time_steps(chunk, beta, per_step_energies=per_step)
        eb_ = b.do_time_steps(chunk, beta, per_step_energies=per_step)
        if per_step and not np.array_equal(ea_, eb_):
            bad += 1
        if not np.array_equal(a.energies(), b.energies()):
            bad += 1
    same = bool(np.array_equal(a.packed(), b.packed())) if hasattr(a, "packed") and graph.kind == _capi.KIND_LATTICE2D else bool(np.array_equal(a.states(), b.states()))
    print(f"{name:52s} R={R:4d} steps={steps:6d} x2  mismatching chunks {bad}  final configurations identical: {same}  ({time.time() - t0:.1f} s)", flush=True)
    return bad == 0 and same


def main():
    rng = np.random.default_rng(7)
    ok = True
    g = _capi.Graph(*square(4096, 4096))
    ok &= soak("c2 4096^2 uniform J (2 lanes)", g, 256, 4000, 500, 0.4407)
    ok &= soak("c2 4096^2 uniform J, energies after every step", g, 64, 600, 100, 0.4407, per_step=True)
    g = _capi.Graph(*square(2048, 2048, rng))
    ok &= soak("c4 2048^2 +-J", g, 128, 6000, 1000, 1.0)
    ok &= soak("c4 2048^2 +-J, energies after every step", g, 128, 1000, 200, 1.0, per_step=True)
    g = _capi.Graph(*square(1024, 1024))
    ok &= soak("c3 1024^2 (persistent strips)", g, 64, 20000, 2000, 0.44)
    ea, eb, ej = square(2048, 2048)
    g = _capi.Graph(ea, eb, ej, biases=np.full(2048 * 2048, 0.5))
    ok &= soak("2048^2 uniform field (multi-class kernel)", g, 128, 3000, 500, 0.4)
    ea, eb, ej = cubic(256)
    g = _capi.Graph(ea, eb, ej, force_general=True)
    ok &= soak("c5 256^3 general path (bit-sliced packed)", g, 64, 1500, 250, 0.2217)
    ea, eb, ej = square(2048, 2048, rng, gaussian=True)
    g = _capi.Graph(ea, eb, ej, force_general=True)
    ok &= soak("2048^2 Gaussian glass (real-coupling packed)", g, 128, 1500, 250, 0.8)
    ok &= soak("2048^2 Gaussian glass, energies after every step", g, 128, 300, 100, 0.8, per_step=True)
    ok &= soak("2048^2 Gaussian glass x 4 (f64 CSR path)", g, 4, 300, 100, 0.8)
    print("SOAK", "OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
