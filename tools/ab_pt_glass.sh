#!/bin/bash
# A/B of library builds on tempering of the 3-d +-J glasses (bit-sliced packed path, exchange rounds on the stream every 10 timesteps):
# tools/ab_pt_glass.sh lib1.so lib2.so ...   (two interleaved rounds; the +-J lines of tools/real_pt_bench.py)
for round in 1 2; do
  for lib in "$@"; do
    ISINGMC_LIB_PATH=$lib python3 tools/real_pt_bench.py 400 2>/dev/null | grep -e "+-J" | grep "on the stream" | awk -v l="$lib" -v r=$round '{print l, "round", r, $0}'
  done
done
