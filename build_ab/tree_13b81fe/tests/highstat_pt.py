#!/usr/bin/env python3
"""High-statistics check of parallel tempering (lives under tests/: it takes Kaufman's exact energies from oracle/exact.py).

Every rung of a ladder must sample ITS Boltzmann distribution whatever the exchange rounds do: a wrong sign or a stale energy in
the swap rule biases the rungs' energies.  K independent ladders (different seeds), exchange rounds every few sweeps on the
device, the energy of every rung after every block of sweeps, mean and standard error over the ladders, z against Kaufman's exact
finite-torus energy for each rung.

    64 x 64, 32 rungs across beta_c            the LDS-resident kernel, exchange rounds on the stream
    1024 x 128, 64 rungs, beta 0.30 .. 0.38    the persistent strip kernel with the exchange inside the launch (c3's path)
    60 x 60, 64 rungs across beta_c            the general path forced onto the replica-packed bit-sliced / real-coupling kernels

    python tests/highstat_pt.py [out.txt]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import exact as X  # noqa: E402
from pyisingmontecarlo_amd.tempering import ClassicalTempering  # noqa: E402


def run(name, W, H, betas, swap_freq, ladders, therm, blocks, block, env=None):
    for key, value in (env or {}).items():
        os.environ[key] = value
    try:
        return _run(name, W, H, betas, swap_freq, ladders, therm, blocks, block)
    finally:
        for key in (env or {}):
            del os.environ[key]


def _run(name, W, H, betas, swap_freq, ladders, therm, blocks, block):
    ea, eb, ej = X.square_lattice_edges(W, H, -1.0)
    edges = [((int(a), int(b)), float(j)) for a, b, j in zip(ea, eb, ej)]
    G = len(betas)
    per_ladder = np.zeros((ladders, G))
    swaps = 0
    t0 = time.time()
    for k in range(ladders):
        pt = ClassicalTempering(edges, seed=int(os.environ.get("HIGHSTAT_SEED", "1000")) + k)
        for b in betas:
            pt.add_graph(float(b))
        pt.timesteps(therm, swap_freq)
        acc = np.zeros(G)
        for _ in range(blocks):
            pt.timesteps(block, swap_freq)
            perm = pt.get_permutation()                       # rung -> slot
            acc += pt._states.energies()[perm]
        per_ladder[k] = acc / blocks
        swaps += pt.get_total_swaps()
        on_stream = pt._on_stream
    mean = per_ladder.mean(axis=0)
    sigma = per_ladder.std(axis=0, ddof=1) / np.sqrt(ladders)
    exact = np.array([X.kaufman_energy(W, H, float(b)) for b in betas])
    z = (mean - exact) / sigma
    lines = [f"# {name}: {W} x {H}, {G} rungs beta {betas[0]:.4f} .. {betas[-1]:.4f}, exchange round every {swap_freq} sweeps, {ladders} ladders x "
             f"({therm} + {blocks} x {block}) sweeps, {swaps} accepted swaps ({swaps / (ladders * (therm + blocks * block) / swap_freq * (G - 1) / 2):.2f} per attempted pair), "
             f"exchange on the device stream: {on_stream}, {time.time() - t0:.0f} s"]
    for r in range(G):
        lines.append(f"rung {r:2d} beta {betas[r]:.5f}  <E> {mean[r]:14.2f} +- {sigma[r]:8.2f}  exact {exact[r]:14.2f}  z = {z[r]:+.2f}")
    lines.append(f"# {name}: rms z = {np.sqrt((z ** 2).mean()):.2f}, mean z = {z.mean():+.2f}, max |z| = {np.abs(z).max():.2f}, "
                 f"within 1/2/3 sigma: {int((np.abs(z) <= 1).sum())}/{int((np.abs(z) <= 2).sum())}/{int((np.abs(z) <= 3).sum())} of {G}")
    return lines


def main():
    out = []
    out += run("LDS-resident kernel, rounds on the stream", 64, 64, np.linspace(0.38, 0.50, 32), 5, 64, 20000, 400, 100)
    out += run("persistent strips, exchange inside the launch", 1024, 128, np.linspace(0.30, 0.38, 64), 10, 32, 3000, 300, 50)
    # 60 x 60 is not 64-wide: the edge list takes the general path; forced onto the two replica-packed paths (tempering on them
    # relabels per-slot thresholds / acceptance scales on the device, round 3)
    out += run("replica-packed bit-sliced path (ISINGMC_FORCE_PACKED=1)", 60, 60, np.linspace(0.36, 0.52, 64), 5, 16, 20000, 300, 100,
               env={"ISINGMC_FORCE_PACKED": "1", "ISINGMC_DISABLE_REAL": "1"})
    out += run("replica-packed real-coupling path (ISINGMC_FORCE_REAL=1)", 60, 60, np.linspace(0.36, 0.52, 64), 5, 16, 20000, 300, 100,
               env={"ISINGMC_FORCE_REAL": "1"})
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
