#!/bin/bash
# A/B of library builds on the real-coupling path (tools/real_bench.py): tools/ab_real.sh lib1.so lib2.so ...   (two interleaved rounds)
for round in 1 2; do
  for lib in "$@"; do
    ISINGMC_LIB_PATH=$lib python3 tools/real_bench.py 40 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$lib round $round %-46s %.4g attempts/s' % (d['case'], d['attempts_per_s']))"
  done
done
