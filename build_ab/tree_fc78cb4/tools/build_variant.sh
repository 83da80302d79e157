#!/bin/bash
# A/B library builds: tools/build_variant.sh <name> [-DFLAG ...]  ->  build_ab/<name>.so (git-ignored, travels to the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_ab
S=pyisingmontecarlo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" -o build_ab/$name.so \
  $S/isingmc.hip $S/strip_kernels.hip $S/spread_kernels.hip $S/mc_kernels.hip $S/packed_uni_kernels.hip $S/real_kernels.hip $S/host_logic.cpp
echo built build_ab/$name.so
