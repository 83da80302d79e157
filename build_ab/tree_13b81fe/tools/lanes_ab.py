import sys, time, os, numpy as np
sys.path.insert(0, '.')
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import square
for (W, H, R) in ((1024, 1024, 64), (1024, 1024, 16), (2048, 2048, 32), (512, 1024, 64), (4096, 4096, 16)):
    g = _capi.Graph(*square(W, H), nvars=W*H); st = _capi.States(g, _capi.make_seeds(1, R)); st.do_time_steps(20, 0.4)
    ms = st.do_time_steps_timed(300, 0.4)
    print(f"streams={os.environ.get('ISINGMC_STREAMS','auto')} {W}x{H} R={R}: {ms/300*1e3:.1f} us/step, {R*W*H*300/(ms*1e-3):.3e} attempts/s", flush=True)
