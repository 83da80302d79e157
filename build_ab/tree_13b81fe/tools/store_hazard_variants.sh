#!/bin/bash
# Round 3 diagnostics: the sweep+measure kernel with the store where rounds 1-2 had it (ISINGMC_DIAG_OLD_FUSED_STORE, see
# csrc/lattice_kernels.hpp) -- as shipped (0), with the counting pushed >= 10 instructions behind the store plus s_nop (1, 8),
# with an immediate soffset (m1) -- against the oracle on 4096^2 x 256 (tests/diag_c2_parity2.py).  Build the variants first:
#   for v in 0 1 8 -1; do bash tools/build_variant.sh oldfused_${v/-/m} -DISINGMC_DIAG_OLD_FUSED_STORE=$v; done
cd "$(dirname "$0")/.."
for v in 0 1 8 m1; do
  echo "== variant oldfused_$v"
  ISINGMC_LIB_PATH=$PWD/pyisingmontecarlo_amd/lib/ab/oldfused_$v.so timeout -k 10 200 python tests/diag_c2_parity2.py 4096 256 2>&1 | grep -v amdgpu.ids | cut -c1-110
done
echo "== the shipped library"
timeout -k 10 200 python tests/diag_c2_parity2.py 4096 256 2>&1 | grep -v amdgpu.ids | cut -c1-110
