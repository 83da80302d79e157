mkdir -p gpurun_out/r02g
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_api.py -q -k "packed or sampling or per_step" > gpurun_out/r02g/packed2.log 2>&1; tail -4 gpurun_out/r02g/packed2.log
(for rep in 1 2; do for v in libisingmc ab/pk_no_uni; do echo "== $v"; ISINGMC_LIB_PATH=$PWD/pyisingmontecarlo_amd/lib/$v.so python tools/bench_configs.py c5 --steps 20; done; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/r02g/c5_ab2.txt; cut -c1-200 gpurun_out/r02g/c5_ab2.txt
