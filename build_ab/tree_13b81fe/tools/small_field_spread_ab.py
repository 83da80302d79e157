#!/usr/bin/env python3
"""A/B of the multi-class LDS-resident kernel (small lattices with a field / open boundaries / anisotropic couplings): one lane per
quad (ISINGMC_RESIDENT_SPREAD=0) against eight lanes per quad.  Run once per setting."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import square  # noqa: E402

for (W, H), R, what in (((256, 16), 4, "field"), ((256, 64), 64, "field"), ((256, 64), 64, "random field"), ((256, 64), 64, "open"), ((256, 64), 64, "aniso"),
                        ((256, 128), 64, "field"), ((256, 64), 2048, "field")):
    ea, eb, ej = square(W, H)
    biases = None
    if what == "field":
        biases = np.full(W * H, 0.5)
    elif what == "random field":
        biases = np.where(np.random.default_rng(1).integers(0, 2, W * H) == 1, -0.5, 0.5)
    elif what == "open":
        keep = ~((ea % W == W - 1) & (eb % W == 0)) & ~((ea // W == H - 1) & (eb // W == 0))
        ea, eb, ej = ea[keep], eb[keep], ej[keep]
    else:
        ej = np.where(np.arange(len(ej)) % 2 == 1, 0.4 * ej, ej)        # down bonds weaker
    g = _capi.Graph(ea, eb, ej, nvars=W * H, biases=biases)
    assert g.kind == _capi.KIND_LATTICE2D and g.info.fast_path != 0, what
    st = _capi.States(g, _capi.make_seeds(1, R))
    st.do_time_steps(50, 0.44)
    T = 2000 if R <= 64 else 300
    ms = min(st.do_time_steps_timed(T, 0.44) for _ in range(3))
    print(f"spread={os.environ.get('ISINGMC_RESIDENT_SPREAD', '1')} {W:4d}x{H:<4d} x {R:5d} {what:13s}: {ms / T * 1e3:7.2f} us/step", flush=True)
