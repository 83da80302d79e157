#!/bin/bash
# quads per thread of the looping sweep kernel (ISINGMC_SWEEP_ITERS = 1: one-quad kernel)
for round in 1 2; do
for k in 1 2 4 8; do
  ISINGMC_SWEEP_ITERS=$k python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('iters=$k round $round', '%.4g attempts/s  launch=%.1f us e/site=%.5f' % (d['device_attempts_per_s'], d['roofline']['avg_launch_us'], d['energy_per_site']))"
done
done
