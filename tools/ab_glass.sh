#!/bin/bash
# A/B of library builds on the 3-d lattices of the replica-packed path (ferromagnet = c5's kernel, +-J glass = the sign-mask one) at
# latency-bound and throughput-bound sizes: tools/ab_glass.sh lib1.so lib2.so ...   (two interleaved rounds)
for round in 1 2; do
  for lib in "$@"; do
    for size in "64 400" "128 100" "256 20"; do
      ISINGMC_LIB_PATH=$lib python3 tools/glass3d_bench.py $size 2>/dev/null | awk -v l="$lib" -v r=$round '{print l, "round", r, $1, $2, $3, $6, $7, $8, $9}'
    done
  done
done
