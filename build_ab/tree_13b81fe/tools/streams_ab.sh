#!/bin/bash
# replica lanes (streams) for the headline config: do the launch tails of one replica block overlap the other's work?
for round in 1 2; do
for n in 1 2 4; do
  ISINGMC_STREAMS=$n python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('streams=$n round $round', '%.4g attempts/s  ms/step=%.4f e/site=%.5f' % (d['device_attempts_per_s'], d['ms_per_step'], d['energy_per_site']))"
done
done
