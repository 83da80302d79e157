#!/bin/bash
# A/B library builds: tools/build_variant.sh <name> [-DFLAG ...]  ->  pyisingmontecarlo_amd/lib/ab/<name>.so
# (select one with ISINGMC_LIB_PATH=...; the variants are git-ignored but travel to the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p pyisingmontecarlo_amd/lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" \
  -o pyisingmontecarlo_amd/lib/ab/$name.so pyisingmontecarlo_amd/csrc/isingmc.hip pyisingmontecarlo_amd/csrc/strip_kernels.hip pyisingmontecarlo_amd/csrc/spread_kernels.hip pyisingmontecarlo_amd/csrc/mc_kernels.hip pyisingmontecarlo_amd/csrc/packed_uni_kernels.hip pyisingmontecarlo_amd/csrc/real_kernels.hip pyisingmontecarlo_amd/csrc/host_logic.cpp
echo built pyisingmontecarlo_amd/lib/ab/$name.so
