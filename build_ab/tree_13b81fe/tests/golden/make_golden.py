#!/usr/bin/env python3
"""Generates the fixtures in this directory.  No reference code is (or can be) involved: the reference
ships no tests, fixtures or golden vectors and cannot be built here (SURVEY.md sections 4, 8c).

  philox_kat.json      Random123 (D. E. Shaw Research) kat_vectors entries for philox4x32-10: published
                       known answers, typed in from the Random123 distribution -- NOT computed here.
  xoshiro256pp.json    Blackman & Vigna's xoshiro256++ from state [1,2,3,4] (the vector the rand crate
                       tests against) -- published, not computed here.
  make_seeds.json      oracle make_seeds (SplitMix64 -> xoshiro256++), regression pin for oracle == library
  kaufman.json         exact finite-torus energies from oracle/exact.py (Kaufman 1949)
  lattice_sweeps.json  sha256 of oracle checkerboard configurations: regression pin of the spec
  real_path.json       real-coupling packed path: sha256 of the log2 table (it comes out of libm's log2: the pin
                       detects a libm that rounds an entry the other way) and Lambda_q(u) known answers; a sha256 of engine E
                       configurations on a small Gaussian glass: regression pin of the spec
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import exact as X  # noqa: E402
from oracle import oracle as O  # noqa: E402


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")


def main():
    dump("philox_kat.json", {"source": "Random123 kat_vectors: philox4x32 10", "vectors": [
        {"ctr": ["00000000"] * 4, "key": ["00000000"] * 2,
         "out": ["6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8"]},
        {"ctr": ["ffffffff"] * 4, "key": ["ffffffff"] * 2,
         "out": ["408f276d", "41c83b0e", "a20bc7c6", "6d5451fd"]},
        {"ctr": ["243f6a88", "85a308d3", "13198a2e", "03707344"], "key": ["a4093822", "299f31d0"],
         "out": ["d16cfe09", "94fdcceb", "5001e420", "24126ea1"]}]})
    dump("xoshiro256pp.json", {"source": "xoshiro256++ reference implementation, state [1,2,3,4]",
                               "state": [1, 2, 3, 4],
                               "out": [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205,
                                       9973669472204895162, 14011001112246962877, 12406186145184390807,
                                       15849039046786891736, 10450023813501588000]})
    dump("make_seeds.json", {"source": "oracle orc_make_seeds",
                             "cases": [{"seed_gen": s, "seeds": [int(x) for x in O.make_seeds(s, 6)]}
                                       for s in (0, 1, 1234, 2 ** 63 + 5)]})
    cases = [(16, 16, 0.3), (64, 64, 0.3), (64, 64, 0.4407), (256, 256, 0.4407), (1024, 1024, 0.1),
             (1024, 1024, 1.0), (4096, 4096, 0.4407)]
    dump("kaufman.json", {"source": "oracle/exact.py kaufman_energy, E = -sum s s on the periodic W x H lattice",
                          "cases": [{"W": W, "H": H, "beta": b, "E": X.kaufman_energy(W, H, b)} for W, H, b in cases]})
    sweeps = []
    for W, H, jabs, jpos, j_seed, beta, seed, T in [
            (64, 64, 1.0, 0, None, 0.4407, 11, 12), (256, 8, 1.0, 1, None, 0.7, 12, 10),
            (128, 16, 0.5, 0, None, 1.3, 2 ** 64 - 3, 8), (256, 16, 1.0, 0, 2024, 0.9, 13, 10)]:
        if j_seed is None:
            lat = O.Lat(W, H, jabs, jpos)
        else:
            _, _, ej = X.square_lattice_edges(W, H, jabs, np.random.default_rng(j_seed))
            lat = O.Lat(W, H, jabs, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8))
        st = lat.init(seed)
        for t in range(T):
            lat.sweep(st, seed, t, beta)
        sweeps.append({"W": W, "H": H, "jabs": jabs, "jpos": jpos, "j_seed": j_seed, "beta": beta, "seed": seed,
                       "T": T, "sha256": hashlib.sha256(st.tobytes()).hexdigest(), "energy": lat.energy_mag(st)[0]})
    dump("lattice_sweeps.json", {"source": "oracle engine B (checkerboard spec), N_PLANES = 7", "cases": sweeps})
    lt = O.rj_log_table()
    rng = np.random.default_rng(99)
    us = [0, 1, 2, 3, 255, 256, 65535, 65536, 2 ** 24 - 1, 2 ** 24, 2 ** 24 + 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32 - 129,
          2 ** 32 - 128] + [int(x) for x in rng.integers(0, 2 ** 32, 40, dtype=np.uint64)]
    ea, eb, _ = X.square_lattice_edges(12, 10, 1.0)
    grng = np.random.default_rng(7)
    ej, h = grng.normal(size=len(ea)), grng.normal(size=120) * 0.3
    e, st = O.rj_run(ea, eb, ej, 120, O.make_seeds(5, 40), 6, betas=[0.8] * 6, biases=h)
    dump("real_path.json", {"source": "oracle engine E (real-coupling packed spec)",
                            "log_table_sha256": hashlib.sha256(lt.tobytes()).hexdigest(),
                            "lambda": [[u, O.rj_lambda(u)] for u in us],
                            "glass_12x10": {"sha256": hashlib.sha256(st[:40].tobytes()).hexdigest(), "energy0": float(e[0])}})


if __name__ == "__main__":
    main()
