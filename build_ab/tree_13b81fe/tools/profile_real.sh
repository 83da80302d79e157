#!/bin/bash
# rocprofv3 evidence for the real-coupling packed kernel (2048^2 Gaussian glass x 128 replicas): kernel stats, SQ counters,
# HBM traffic.  usage: tools/profile_real.sh <tag>   (on the GPU box; results under gpurun_out/real_<tag>/)
set -uo pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/real_$TAG
CASE="2048^2 gaussian x128"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/real_bench.py" 50 "$CASE" > "$OUT/trace.log" 2>&1
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
echo "trace done"
KFILTER=rj_sweep bash "$ROOT/tools/pmc_sq.sh" real_$TAG tools/real_bench.py 10 "$CASE" > "$OUT/sq_counters.json" 2> "$OUT/sq.err"
echo "sq done"
bash "$ROOT/tools/pmc_traffic.sh" real_$TAG tools/real_bench.py 10 "$CASE" > "$OUT/traffic.txt" 2> "$OUT/traffic.err"
echo "traffic done"
cat "$OUT/kernel_stats.csv" | head -8; cat "$OUT/sq_counters.json"; cat "$OUT/traffic.txt"
