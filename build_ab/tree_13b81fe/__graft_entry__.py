"""Driver entry points: build() compiles every native part for gfx950, smoke() runs one small
invocation of the hot path on cuda:0 and checks it against the CPU oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    """hipcc --offload-arch=gfx950 for libisingmc.so, g++/pybind11 for the host shim, gcc for the
    oracle (building the checker is not using it).  The reference itself (Rust + the un-vendored
    `qmc` crate) cannot be built in this image: there is no oracle/_ref."""
    from pyisingmontecarlo_amd import build as b
    b.build_all()
    from oracle import oracle as O
    O.build()
    import pyisingmontecarlo_amd  # noqa: F401
    from pyisingmontecarlo_amd import _capi
    _capi.lib()
    import py_monte_carlo  # noqa: F401


def smoke() -> None:
    """256x64 ferromagnet + a +-J glass, 3 replicas, 8 timesteps on cuda:0: packed spin words and
    energies must equal the oracle's bit for bit; then the Python API end to end."""
    import numpy as np
    from oracle import exact as X
    from oracle import oracle as O
    from pyisingmontecarlo_amd import _capi

    if _capi.device_count() < 1:
        raise RuntimeError("smoke() needs a HIP device: the engine has no CPU fallback")
    seeds = _capi.make_seeds(7, 3)
    W, H, beta, T = 256, 64, 0.4407, 8
    for rng in (None, np.random.default_rng(2024)):
        ea, eb, ej = X.square_lattice_edges(W, H, -1.0, rng)
        g = _capi.Graph(ea, eb, ej, device=0)
        assert g.kind == _capi.KIND_LATTICE2D
        st = _capi.States(g, seeds)
        st.do_time_steps(T, beta)
        if rng is None:
            lat = O.Lat(W, H, 1.0, 0)
        else:
            lat = O.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8))
        for r, s in enumerate(seeds):
            ref = lat.init(s)
            for t in range(T):
                lat.sweep(ref, s, t, beta)
            assert np.array_equal(st.packed()[r], ref), "HIP checkerboard sweep differs from the oracle"
            assert st.energies()[r] == lat.energy_mag(ref)[0]

    import py_monte_carlo
    edges = [((int(a), int(b)), float(j)) for a, b, j in zip(*X.square_lattice_edges(16, 16, -1.0))]
    lat = py_monte_carlo.Lattice(edges, seed_gen=1234)
    energies, states = lat.run_monte_carlo(0.3, 100, 4)  # BASELINE config c1's shape
    assert energies.shape == (4,) and states.shape == (4, 256) and states.dtype == np.bool_
    ea, eb, ej = O.split_edges(edges)
    for r, s in enumerate(lat.make_seeds(4)):
        e_ref, s_ref = O.gen_run(ea, eb, ej, 256, s, [0.3] * 100)
        assert np.array_equal(states[r].astype(np.uint8), s_ref) and energies[r] == e_ref
    print("smoke ok")


if __name__ == "__main__":
    build()
    if "--smoke" in sys.argv:
        smoke()
