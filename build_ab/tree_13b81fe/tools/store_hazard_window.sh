#!/bin/bash
# Round 3 diagnostics: how long after a register-soffset buffer_store_dwordx4 may its data registers not be written?  The old
# sweep+measure kernel with the third data register complemented (and restored) K wait states behind the store
# (ISINGMC_DIAG_OLD_FUSED_STORE = 20 + K; build: for v in 20 21 22 23 24 26; do bash tools/build_variant.sh oldfused_w$v -DISINGMC_DIAG_OLD_FUSED_STORE=$v; done)
cd "$(dirname "$0")/.."
for v in 20 21 22 23 24 26; do
  echo "== complemented $((v - 20)) wait state(s) behind the store"
  ISINGMC_LIB_PATH=$PWD/pyisingmontecarlo_amd/lib/ab/oldfused_w$v.so timeout -k 10 200 python tests/diag_c2_parity2.py 4096 256 2>&1 | grep -v amdgpu.ids | cut -c1-90 | awk '{n += ($7 > 0)} END {print NR " replicas compared with the oracle, " n " with wrong spins"}'
done
