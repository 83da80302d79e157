#!/bin/bash
# The library's HOST code under UndefinedBehaviorSanitizer on the GPU box (device code is not instrumented: GPU sanitizers are not
# available on the pool; hipcc ignores -fsanitize=undefined for amdgcn).  Build here, run through gpurun:
#   bash tools/build_variant.sh ubsan -fsanitize=undefined -fno-sanitize=vptr,function
#   gpurun -- 'bash tools/ubsan_host.sh'
# The C-ABI-level parity suites run against that build with clang's UBSan runtime preloaded; any report aborts the run.
# (Tests that go through the pybind11 shim are left out: the shim links the regular library by rpath.)
cd "$(dirname "$0")/.."
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
LD_PRELOAD=$RT ISINGMC_LIB_PATH=$PWD/pyisingmontecarlo_amd/lib/ab/ubsan.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_real.py tests/test_gpu_strip.py tests/test_gpu_field_open.py \
  tests/test_gpu_underload.py tests/test_gpu_counters.py tests/test_gpu_sequences.py -q -x -k "not classic and not python_api"
