#!/bin/bash
# The tempering benchmark (tools/real_pt_bench.py of each tree) in whole source trees side by side: tools/ab_trees_pt.sh [filter] dir1 dir2 ...  ("." = this tree)
filter=$1; shift
for round in 1 2; do
  for d in "$@"; do
    (cd $d && python3 tools/real_pt_bench.py 400 2>/dev/null | grep "on the stream" | grep -e "$filter" | awk -v l="$d" -v r=$round '{print l, "round", r, $0}')
  done
done
