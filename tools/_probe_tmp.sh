mkdir -p gpurun_out/r02c
timeout -k 10 400 python -m pytest tests/test_gpu_strip.py -x -q > gpurun_out/r02c/strip5.log 2>&1; tail -3 gpurun_out/r02c/strip5.log
(for nw in 1 4; do for v in libisingmc ab/strip_nopair ab/strip_nopoll; do echo "== NW=$nw $v"; ISINGMC_STRIP_NW=$nw ISINGMC_LIB_PATH=$PWD/pyisingmontecarlo_amd/lib/$v.so python tools/strip_probe.py 1024 400; done; done
for nw in 1 4; do echo "== c3 NW=$nw"; ISINGMC_STRIP_NW=$nw python tools/bench_configs.py c3 --steps 400; done) > gpurun_out/r02c/probe3.txt 2>&1; grep -v amdgpu.ids gpurun_out/r02c/probe3.txt
