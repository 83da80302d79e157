mkdir -p gpurun_out/r02h
timeout -k 10 600 python -m pytest tests/test_gpu_strip.py tests/test_gpu_api.py -q -x -k "tempering or strip" > gpurun_out/r02h/pt.log 2>&1; tail -15 gpurun_out/r02h/pt.log
(for rep in 1 2; do for ik in 1 0; do echo "== in-kernel=$ik"; ISINGMC_PT_IN_KERNEL=$ik python tools/bench_configs.py c3 --steps 400; done; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/r02h/c3.txt; cat gpurun_out/r02h/c3.txt
