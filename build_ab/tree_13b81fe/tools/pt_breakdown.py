import sys, time, numpy as np
sys.path.insert(0, '.')
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import square
L, G = 1024, 64
g = _capi.Graph(*square(L, L), nvars=L*L)
st = _capi.States(g, _capi.make_seeds(1, G))
betas = np.linspace(0.1, 1.0, G)
st.set_betas(betas)
st.do_time_steps(50)
for T in (10, 100, 1000):
    t0 = time.perf_counter(); st.do_time_steps(T); dt = time.perf_counter() - t0
    print(f"do_time_steps({T}): {dt*1e6/T:.1f} us/step")
ms = st.do_time_steps_timed(1000, 0.4); print("timed 1000 steps device ms/step", ms/1000*1e3, "us")
t0 = time.perf_counter()
for _ in range(100): e = st.energies()
print("energies():", (time.perf_counter()-t0)*1e4, "us")
t0 = time.perf_counter()
for _ in range(100): st.set_betas(betas)
print("set_betas():", (time.perf_counter()-t0)*1e4, "us")
perm = np.arange(G, dtype=np.uint32)
t0 = time.perf_counter()
for i in range(100): _capi.pt_swap_round(1, i, betas, e, perm)
print("pt_swap_round():", (time.perf_counter()-t0)*1e4, "us")
# small lattices: launch-bound regime
for (W, H, R) in ((64, 64, 64), (256, 256, 64), (512, 512, 256)):
    g2 = _capi.Graph(*square(W, H), nvars=W*H); s2 = _capi.States(g2, _capi.make_seeds(1, R)); s2.do_time_steps(20, 0.4)
    t0 = time.perf_counter(); s2.do_time_steps(500, 0.4); dt = time.perf_counter() - t0
    print(f"{W}x{H} R={R}: {dt*1e6/500:.1f} us/step, {R*W*H*500/dt:.3e} attempts/s")
# general path, tiny graphs (BASELINE c1 = 16x16, 4 experiments)
for (W, H, R) in ((16, 16, 4), (16, 16, 256), (32, 32, 64)):
    g2 = _capi.Graph(*square(W, H), nvars=W*H); s2 = _capi.States(g2, _capi.make_seeds(1, R)); s2.do_time_steps(20, 0.3)
    t0 = time.perf_counter(); s2.do_time_steps(1000, 0.3); dt = time.perf_counter() - t0
    print(f"general {W}x{H} R={R}: {dt*1e6/1000:.1f} us/step, {R*W*H*1000/dt:.3e} attempts/s")
