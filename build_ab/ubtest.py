import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import cubic
for L in (32, 64):
    ea, eb, ej = cubic(L)
    ej = ej * np.random.default_rng(7).choice([-1.0, 1.0], len(ej))
    g = _capi.Graph(ea, eb, ej, nvars=L ** 3)
    for mode in ("one beta", "per-replica betas", "per-replica, calls of 10"):
        st = _capi.States(g, _capi.make_seeds(1, 64))
        if mode != "one beta":
            st.set_betas(np.linspace(0.2, 1.6, 64))
        if mode.endswith("10"):
            import time
            st.do_time_steps(20, None); 
            t0 = time.perf_counter()
            for _ in range(40): st.do_time_steps(10, None)
            st.energies()
            us = (time.perf_counter() - t0) / 400 * 1e6
        else:
            b = 0.9 if mode == "one beta" else None
            st.do_time_steps(20, b)
            us = min(st.do_time_steps_timed(400, 0.9 if b else 0.0) if b else _timed(st) for _ in range(2)) if False else None
            import ctypes as C
            ms = C.c_float()
            best = 1e9
            for _ in range(2):
                if b is None:
                    _capi._check(_capi.lib().isingmc_do_time_steps_timed(st._h, 400, None, 0, C.byref(ms)))
                else:
                    ms.value = st.do_time_steps_timed(400, b)
                best = min(best, ms.value)
            us = best / 400 * 1e3
        print(f"{os.path.basename(os.getcwd()):16s} {L}^3 +-J x64 {mode:26s} {us:7.2f} us/step", flush=True)
