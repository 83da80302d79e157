#!/bin/bash
# Round-3 evidence files that DESIGN.md cites: exchange-round cost by swap frequency and the kernel /
# gap timeline of a tempering round.  usage (GPU box): tools/evidence_r03.sh ; results under gpurun_out/evidence_r03/
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/evidence_r03
mkdir -p "$OUT"
python3 "$ROOT/tools/pt_round_cost.py" > "$OUT/pt_round_cost.txt" 2>&1
echo "pt_round_cost done"
cd /tmp && export TMPDIR=/tmp
ISINGMC_PT_IN_KERNEL=0 rocprofv3 --kernel-trace --output-format csv -d "$OUT/c3_trace" -- python3 "$ROOT/tools/bench_configs.py" c3 --steps 200 > "$OUT/c3_trace.log" 2>&1
python3 "$ROOT/tools/trace_gaps.py" "$OUT/c3_trace" 40 > "$OUT/c3_round_timeline.txt" 2>&1
echo "timeline done"
cat "$OUT/pt_round_cost.txt" "$OUT/c3_round_timeline.txt"
