mkdir -p gpurun_out/r02j
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02j/probe8b.txt
import os, sys
sys.path.insert(0, '.')
import numpy as np
for lib, nw in (("ab/strip2_w4", "8"), ("libisingmc", "8"), ("libisingmc", "4")):
    import subprocess
    code = f"""
import os, sys; sys.path.insert(0, '.')
import numpy as np
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import square
L=1024; g=_capi.Graph(*square(L,L), nvars=L*L)
for R in (8,16,32):
    st=_capi.States(g,_capi.make_seeds(1,R)); st.set_betas(np.linspace(0.1,1.0,R)); st.do_time_steps(50)
    ms=min(st.do_time_steps_timed(400,0.4) for _ in range(3)); print('{lib} NW={nw} R=%d: %.2f us/step' % (R, ms/400*1e3), flush=True)
"""
    env = dict(os.environ, ISINGMC_STRIP="1", ISINGMC_STRIP_NW=nw, ISINGMC_LIB_PATH=os.path.abspath(f"pyisingmontecarlo_amd/lib/{lib}.so"))
    subprocess.run([sys.executable, "-c", code], env=env)
PY
