#!/usr/bin/env python3
"""Observables table (north_star: "<E>, <|M|> agree within Monte-Carlo statistical error").

Run on the GPU box:  python tests/observables.py [out.json]     (committed result: profiles/r03_observables.json)

For every configuration: the HIP engine's <E> and <|M|> with the standard error from independent replicas, the reference
value -- EXACT where one exists (Kaufman's finite-torus energy, exact enumeration on <= 16 spins), otherwise oracle engine A
(the CPU restatement of the reference's random-site Metropolis; its own standard error enters the z score) -- and
z = (measured - reference) / sqrt(sigma^2 + sigma_ref^2).  Lives under tests/ because it calls the oracle (test
infrastructure: only tests/, smoke() and bench.py's cpu_baseline leg may)."""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import exact as X  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pyisingmontecarlo_amd import _capi  # noqa: E402

rows = []


def add(config, path, beta, name, per_replica, ref, ref_sigma, ref_kind, sweeps):
    per_replica = np.asarray(per_replica, dtype=np.float64)
    mean, sigma = per_replica.mean(), per_replica.std(ddof=1) / math.sqrt(len(per_replica))
    z = (mean - ref) / math.sqrt(sigma ** 2 + ref_sigma ** 2) if sigma > 0 or ref_sigma > 0 else 0.0
    rows.append({"config": config, "path": path, "beta": beta, "observable": name, "measured": mean, "sigma": sigma,
                 "reference": ref, "reference_sigma": ref_sigma, "reference_kind": ref_kind, "z": z,
                 "replicas": len(per_replica), "sweeps_measured": sweeps})
    print(f"{config:34s} {name:5s} {mean:14.6f} +- {sigma:10.6f}   ref {ref:14.6f} +- {ref_sigma:9.6f} ({ref_kind})  z = {z:+.2f}", flush=True)


def gpu_run(graph, R, therm, steps, beta, seed, every=5, initial=None):
    st = _capi.States(graph, _capi.make_seeds(seed, R), initial_state=initial)
    st.do_time_steps(therm, beta)
    e = st.do_time_steps(steps, beta, per_step_energies=True).mean(axis=1)
    mags = []
    for _ in range(steps // every):
        st.do_time_steps(every, beta)
        mags.append(np.abs(st.magnetisations()))
    return e, np.mean(mags, axis=0)


def cpu_absm(ea, eb, ej, n, R, therm, steps, beta, seed, biases=None, initial=None):
    """<|M|> per chain from oracle engine A (random-site sequential Metropolis, the reference's algorithm)."""
    return O.ref_averages(ea, eb, ej, n, O.make_seeds(seed, R), beta, therm, steps, biases=biases, initial=initial)[1]


def lattice_case(name, L, beta, R, therm, steps, seed, absm_cpu=None):
    ea, eb, ej = X.square_lattice_edges(L, L, -1.0)
    g = _capi.Graph(ea, eb, ej)
    path = "lattice (checkerboard, bit-sliced)" if g.kind == _capi.KIND_LATTICE2D else "general"
    e, m = gpu_run(g, R, therm, steps, beta, seed, initial=np.ones(L * L, dtype=np.uint8))
    add(name, path, beta, "E", e, X.kaufman_energy(L, L, beta), 0.0, "exact (Kaufman 1949)", steps)
    if absm_cpu:
        Rc, thc, stc = absm_cpu
        t0 = time.time()
        mc = cpu_absm(ea, eb, ej, L * L, Rc, thc, stc, beta, seed + 1, initial=np.ones(L * L, dtype=np.uint8))
        add(name, path, beta, "|M|", m, mc.mean(), mc.std(ddof=1) / math.sqrt(Rc), f"oracle engine A ({Rc} chains, {time.time() - t0:.0f} s)", steps)


def enum_case(name, ea, eb, ej, n, beta, biases, R, therm, steps, seed, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    g = _capi.Graph(ea, eb, ej, nvars=n, biases=biases)
    path = {0: "general", 1: "lattice"}[g.kind] + (f" (real-coupling packed, slots {g.info.real_slots})" if env and "ISINGMC_FORCE_REAL" in env else
                                                   " (f64 CSR, LDS-resident)")
    e, m = gpu_run(g, R, therm, steps, beta, seed)
    for k in (env or {}):
        del os.environ[k]
    ex = X.enumerate_graph(ea, eb, ej, n, beta, biases)
    add(name, path, beta, "E", e, ex["E"], 0.0, "exact (enumeration)", steps)
    add(name, path, beta, "|M|", m, ex["absM"], 0.0, "exact (enumeration)", steps)


def engine_a_case(name, path, ea, eb, ej, n, beta, R, therm, steps, seed, cpu, env=None, want_m=True, biases=None, fast_path=None):
    """No exact result: the reference is oracle engine A (the reference's algorithm) on the same couplings, its error included."""
    for k, v in (env or {}).items():
        os.environ[k] = v
    g = _capi.Graph(ea, eb, ej, nvars=n, biases=biases)
    if fast_path is not None:
        assert g.kind == _capi.KIND_LATTICE2D and g.info.fast_path == fast_path, (name, g.kind, g.info.fast_path)
    e, m = gpu_run(g, R, therm, steps, beta, seed)
    for k in (env or {}):
        del os.environ[k]
    Rc, thc, stc = cpu
    t0 = time.time()
    ec, mc = O.ref_averages(ea, eb, ej, n, O.make_seeds(seed + 1, Rc), beta, thc, stc, biases=biases)
    kind = f"oracle engine A ({Rc} chains, {time.time() - t0:.0f} s)"
    add(name, path, beta, "E", e, ec.mean(), ec.std(ddof=1) / math.sqrt(Rc), kind, steps)
    if want_m:
        add(name, path, beta, "|M|", m, mc.mean(), mc.std(ddof=1) / math.sqrt(Rc), kind, steps)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "observables.json")
    # c1: 16 x 16, beta = 0.3 (BASELINE config 1; the general path's LDS-resident kernel: 16 is not 64-wide)
    lattice_case("c1 16x16", 16, 0.3, 256, 500, 4000, 101, absm_cpu=(64, 500, 4000))
    # 64 x 64 at beta_c (critical slowing down: tau ~ L^2.17 sweeps)
    lattice_case("64x64 at beta_c", 64, 0.4407, 128, 20000, 40000, 102, absm_cpu=(32, 10000, 20000))
    lattice_case("256x256", 256, 0.3, 64, 300, 2000, 103, absm_cpu=(16, 200, 600))
    lattice_case("256x256", 256, 0.6, 64, 500, 2000, 104, absm_cpu=(16, 300, 600))
    # end rungs of c3's ladder: 1024 x 1024 at beta = 0.1 and 1.0
    lattice_case("c3 end rung 1024x1024", 1024, 0.1, 32, 100, 400, 105)
    lattice_case("c3 end rung 1024x1024", 1024, 1.0, 32, 300, 400, 106)
    # exact enumeration: 4 x 4 torus; 16-spin random +-J graph with fields -- on the CSR path and on the real-coupling path
    ea, eb, ej = X.square_lattice_edges(4, 4, -1.0)
    enum_case("4x4 torus", ea, eb, ej, 16, 0.35, None, 512, 200, 4000, 107)
    rng = np.random.default_rng(16)
    pairs = set()
    while len(pairs) < 28:
        a, b = (int(v) for v in rng.integers(0, 16, 2))
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    pairs = sorted(pairs)
    ga = np.array([p[0] for p in pairs], dtype=np.uint64)
    gb = np.array([p[1] for p in pairs], dtype=np.uint64)
    gj = rng.choice([-1.0, 1.0], size=len(pairs))
    gh = rng.choice([-0.5, 0.0, 0.5], size=16)
    enum_case("16-spin +-J graph with fields", ga, gb, gj, 16, 0.5, gh, 512, 200, 4000, 108)
    deg = np.bincount(np.concatenate([ga, gb]).astype(np.int64), minlength=16).max()
    if deg <= 15:
        enum_case("16-spin +-J graph with fields", ga, gb, gj, 16, 0.5, gh, 512, 200, 4000, 109, env={"ISINGMC_FORCE_REAL": "1"})
    gj2 = rng.normal(size=len(pairs))
    gh2 = rng.normal(size=16) * 0.4
    if deg <= 15:
        enum_case("16-spin Gaussian graph with fields", ga, gb, gj2, 16, 0.5, gh2, 512, 200, 4000, 110, env={"ISINGMC_FORCE_REAL": "1"})
    # Sherrington-Kirkpatrick glass on 16 spins: the complete graph, degree 15 -- the four-nibble kernel of the real-coupling path
    ka, kb = np.triu_indices(16, 1)
    kj = rng.normal(size=len(ka)) / 4.0
    enum_case("16-spin SK glass (complete graph)", ka.astype(np.uint64), kb.astype(np.uint64), kj, 16, 1.0, None, 512, 200, 4000, 111,
              env={"ISINGMC_FORCE_REAL": "1"})
    # 12-spin graph of degree <= 11 with fields: the three-nibble kernel
    pairs11 = [(a, b) for a in range(12) for b in range(a + 1, 12)]
    ga11 = np.array([p[0] for p in pairs11], dtype=np.uint64)
    gb11 = np.array([p[1] for p in pairs11], dtype=np.uint64)
    enum_case("12-spin complete graph with fields", ga11, gb11, rng.normal(size=len(pairs11)) / 3.0, 12, 0.8, rng.normal(size=12) * 0.3,
              512, 200, 4000, 112, env={"ISINGMC_FORCE_REAL": "1"})
    # the headline lattice at its full size, away from beta_c (at beta_c a 4096^2 lattice needs ~10^7 sweeps to equilibrate):
    # Kaufman's exact finite-torus energy; 32 replicas from the ordered start
    lattice_case("c2 lattice 4096x4096", 4096, 0.35, 32, 300, 200, 113)
    lattice_case("c2 lattice 4096x4096", 4096, 0.55, 32, 300, 200, 114)
    # closer to beta_c = 0.4407 on both sides (correlation length ~ 6 lattice spacings: still equilibrated in ~10^2 sweeps)
    lattice_case("c2 lattice 4096x4096", 4096, 0.40, 64, 1000, 400, 122)
    lattice_case("c2 lattice 4096x4096", 4096, 0.48, 64, 1000, 400, 123)
    # c5's kernel (uniform-degree replica-packed path) on a 16^3 cubic lattice at the 3-d critical point
    ca, cb, cj = X.cubic_lattice_edges(16, -1.0)
    engine_a_case("16^3 cubic at beta_c (c5's kernel)", "general (replica-packed bit-sliced, degree 6)", ca, cb, cj, 4096, 0.2217, 256, 3000, 10000, 115,
                  cpu=(64, 2000, 6000), env={"ISINGMC_FORCE_PACKED": "1"})
    # c4's kernel (+-J sign planes on the checkerboard path) on a 64 x 64 +-J glass at beta = 1
    ga4, gb4, gj4 = X.square_lattice_edges(64, 64, 1.0, rng=np.random.default_rng(2024))
    engine_a_case("64x64 +-J glass (c4's kernel)", "lattice (checkerboard, +-J sign planes)", ga4, gb4, gj4, 4096, 1.0, 128, 5000, 10000, 116,
                  cpu=(64, 3000, 6000), want_m=False)
    # the multi-class checkerboard kernels (fields, open boundaries, anisotropic couplings) on a 256 x 16 lattice: their class
    # tables are bit-exact against oracle engine B's spin-by-spin formula in the parity tests; here the resulting chain against
    # the reference's algorithm (engine A) on the same Hamiltonian
    W, H = 256, 16
    mrng = np.random.default_rng(77)

    def lattice_edges(jx, jy, open_x, open_y):
        ids = np.arange(W * H, dtype=np.uint64).reshape(H, W)
        ea_ = np.stack([ids, ids], axis=-1).reshape(-1)
        eb_ = np.stack([np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1)
        ej_ = np.stack([np.full((H, W), -jx), np.full((H, W), -jy)], axis=-1).reshape(-1)
        keep = np.ones(len(ea_), dtype=bool)
        if open_x:
            keep &= ~((ea_ % W == W - 1) & (eb_ % W == 0))
        if open_y:
            keep &= ~((ea_ // W == H - 1) & (eb_ // W == 0))
        return np.ascontiguousarray(ea_[keep]), np.ascontiguousarray(eb_[keep]), np.ascontiguousarray(ej_[keep])

    label = "lattice (checkerboard, multi-class kernel)"
    cpu = (64, 1500, 5000)
    a, b, j = lattice_edges(1.0, 1.0, False, False)
    engine_a_case("256x16 uniform field h = 0.3", label, a, b, j, W * H, 0.4, 128, 2000, 8000, 117, cpu, biases=np.full(W * H, 0.3), fast_path=1)
    engine_a_case("256x16 random field +-0.5", label, a, b, j, W * H, 0.4, 128, 2000, 8000, 118, cpu,
                  biases=np.where(mrng.integers(0, 2, W * H) == 1, -0.5, 0.5))
    a, b, j = lattice_edges(1.0, 1.0, True, True)
    engine_a_case("256x16 open boundaries", label, a, b, j, W * H, 0.4, 128, 2000, 8000, 119, cpu, fast_path=2)
    engine_a_case("256x16 open boundaries, h = 0.4", label, a, b, j, W * H, 0.4, 128, 2000, 8000, 120, cpu, biases=np.full(W * H, 0.4))
    a, b, j = lattice_edges(1.0, 0.4, False, False)
    engine_a_case("256x16 anisotropic Jy = 0.4 Jx", label, a, b, j, W * H, 0.5, 128, 2000, 8000, 121, cpu)
    zs = np.array([r["z"] for r in rows])
    summary = {"rows": len(rows), "max_abs_z": float(np.abs(zs).max()), "rms_z": float(np.sqrt((zs ** 2).mean())),
               "within_1_sigma": int((np.abs(zs) <= 1).sum()), "within_2_sigma": int((np.abs(zs) <= 2).sum()),
               "within_3_sigma": int((np.abs(zs) <= 3).sum()),
               "note": "independent estimates scatter as a unit normal: ~68 % of rows within 1 sigma, ~95 % within 2"}
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as f:
        json.dump({"summary": summary, "rows": rows}, f, indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
