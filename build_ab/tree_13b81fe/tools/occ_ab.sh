#!/bin/bash
# occupancy sweep of the headline kernel: unused LDS per workgroup limits the waves per SIMD
for lds in 0 24000 30000 40000 60000; do
  ISINGMC_DEBUG_SWEEP_LDS=$lds python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('lds=$lds', '%.4g attempts/s  launch=%.1f us' % (d['device_attempts_per_s'], d['roofline']['avg_launch_us']))"
done
