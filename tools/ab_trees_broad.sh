#!/bin/bash
# A broad set of the repository's own benchmarks in whole source trees side by side (the regression net of a round):
# tools/ab_trees_broad.sh dir1 dir2 ...   ("." = this tree; an older tree: git archive <commit> | tar -x -C build_ab/tree_<commit>, build it there)
for d in "$@"; do
  (cd $d
   for cmd in "tools/small_configs.py" "tools/general_bench.py" "tools/field_bench.py 30" "tools/few_replicas.py" "tools/glass3d_bench.py 64 200" "tools/glass3d_bench.py 128 50" \
              "tools/strip_probe.py 1024 200" "tools/real_small.py" "tools/small_call_overhead.py" "tools/sampling_timing.py" "tools/per_step_cost.py"; do
     timeout -k 10 300 python3 $cmd 2>/dev/null | grep -v "^$" | cut -c1-230 | awk -v l="$d" -v c="$cmd" '{print l " | " c " | " $0}'
   done
   timeout -k 10 300 python3 tools/bench_configs.py c3 c4 c5 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$d | bench_configs | %s %.4g attempts/s %.2f us/step' % (d['config'], d['attempts_per_s'], d['ms_per_step']*1e3))"
   timeout -k 10 300 python3 tools/real_bench.py 30 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$d | real_bench | %s %.4g attempts/s' % (d['case'], d['attempts_per_s']))"
  )
done
