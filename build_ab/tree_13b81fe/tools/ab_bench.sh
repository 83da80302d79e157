#!/bin/bash
# A/B the headline bench across library builds: tools/ab_bench.sh lib1.so lib2.so ... (interleaved, 2 rounds)
for round in 1 2; do
  for lib in "$@"; do
    ISINGMC_LIB_PATH=$lib python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'round $round', '%.4g attempts/s  frac=%.4f  launch=%.1f us  e/site=%.5f' % (d['device_attempts_per_s'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['energy_per_site']))"
  done
done
