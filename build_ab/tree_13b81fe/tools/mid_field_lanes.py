import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import square
for L, R in ((1024, 64), (2048, 64), (512, 256), (256, 256), (256, 1024)):
    ea, eb, ej = square(L, L)
    g = _capi.Graph(ea, eb, ej, nvars=L * L, biases=np.full(L * L, 0.25))
    st = _capi.States(g, _capi.make_seeds(1, R))
    st.do_time_steps(70, 0.44)
    for lanes in ("1", "2"):
        os.environ["ISINGMC_STREAMS"] = lanes
        ms = min(st.do_time_steps_timed(400, 0.44) for _ in range(2))
        print(f"field h=0.25 {L}^2 x {R}: lanes={lanes} {ms / 400 * 1e3:7.2f} us/step {R * L * L * 400 / (ms * 1e-3):.3e} attempts/s", flush=True)
