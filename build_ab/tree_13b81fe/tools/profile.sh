#!/bin/bash
# rocprofv3 passes for the headline bench (run on the GPU box through gpurun):
#   1. --kernel-trace --stats   per-kernel durations
#   2..4. --pmc passes          HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes: TCC slots) + SQ mix
# Counters are never combined with tracing options other than --kernel-trace (gpurun rule).
# usage: tools/profile.sh <tag> [bench args...]
set -euo pipefail
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
ARGS=${*:---steps 50 --warmup 5 --no-cpu-baseline}   # PMC passes: counters are per launch
TRACE_ARGS=${TRACE_ARGS:---no-cpu-baseline}          # trace pass: the default bench (1000 steps + 100 warm-up)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $TRACE_ARGS > "$OUT/trace.log" 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
echo "sq done"
python3 "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG" > "$OUT/summary.json"
cat "$OUT/summary.json"
