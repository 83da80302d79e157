#!/bin/bash
# A/B of library builds on BASELINE c4's kernel (2048^2 +-J x 128 replicas, geometric beta schedule): tools/ab_c4.sh lib1.so lib2.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    ISINGMC_LIB_PATH=$lib python3 tools/bench_configs.py c4 --steps 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib round $round c4 %.4g attempts/s  %.2f us/step  e/site %.6f' % (d['attempts_per_s'], d['ms_per_step']*1e3, d['final_energy_per_site']))"
  done
done
