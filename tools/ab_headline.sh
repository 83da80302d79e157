#!/bin/bash
# A/B of library builds on the headline (bench.py, 4096^2 x 256) and c4 (2048^2 +-J x 128 here): tools/ab_headline.sh lib1.so lib2.so ...
for round in 1 2; do
  for lib in "$@"; do
    ISINGMC_LIB_PATH=$lib python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib round $round c2 %.4g attempts/s  %.4f ms/step  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
    ISINGMC_LIB_PATH=$lib python3 tools/bench_configs.py c4 --steps 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib round $round c4 %.4g attempts/s  %.4f ms/step' % (d['attempts_per_s'], d['ms_per_step']))"
  done
done
