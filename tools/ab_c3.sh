#!/bin/bash
# A/B of library builds on the persistent strip kernel: tools/strip_probe.py (one launch, no exchange rounds) and BASELINE c3 (64 rungs, a round every 10 timesteps)
for round in 1 2; do
  for lib in "$@"; do
    ISINGMC_LIB_PATH=$lib python3 tools/strip_probe.py 1024 400 2>/dev/null | awk -v l="$lib" -v r=$round '{print l, "round", r, $1, $2, $3, $4, $5, $6}'
    ISINGMC_LIB_PATH=$lib python3 tools/bench_configs.py c3 --steps 400 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib round $round c3 %.4g attempts/s  %.2f us/step swaps %d' % (d['attempts_per_s'], d['ms_per_step']*1e3, d['total_swaps']))"
  done
done
