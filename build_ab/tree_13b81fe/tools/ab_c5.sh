#!/bin/bash
# A/B of library builds on BASELINE c5 (256^3 x 64 replicas, one-degree packed kernel): tools/ab_c5.sh lib1.so lib2.so ... (interleaved, 2 rounds)
for round in 1 2; do
  for lib in "$@"; do
    ISINGMC_LIB_PATH=$lib python3 tools/bench_configs.py c5 --steps 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'round $round', '%.4g attempts/s  %.1f us/step  e/site=%.5f' % (d['attempts_per_s'], d['ms_per_step']*1e3, d['energy_per_site']))"
  done
done
