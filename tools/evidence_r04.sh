#!/bin/bash
# Round-4 counter evidence on ONE binary: kernel trace + FETCH_SIZE / WRITE_SIZE / SQ passes of the headline
# kernel, the c3 strip kernel, c4's +-J instantiation, c5's one-degree packed kernel and the real-coupling kernel, then
# tools/evidence_r04.py condenses them into profiles/ (sq_latest.json, traffic_*.json, r04_*), each stamped with the sha256 of the
# libisingmc.so and of the kernel sources they were measured on.  usage (GPU box): bash tools/evidence_r04.sh
# Counters are collected in their own rocprofv3 runs, with --kernel-trace only (gpurun rule); the program follows `--` directly.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/evidence_r04
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile.sh r04 > "$OUT/profile.log" 2>&1; echo "headline profile done"
KFILTER=lat_sweep_loop bash tools/pmc_sq.sh r04_c2 > "$OUT/sq_c2.json" 2> "$OUT/sq_c2.err"; echo "sq c2 done"
bash tools/pmc_traffic.sh r04_c3 tools/bench_configs.py c3 --steps 200 > "$OUT/traffic_c3.txt" 2>&1; echo "traffic c3 done"
KFILTER=lat_strip bash tools/pmc_sq.sh r04_c3 tools/bench_configs.py c3 --steps 200 > "$OUT/sq_c3.json" 2> "$OUT/sq_c3.err"; echo "sq c3 done"
bash tools/pmc_traffic.sh r04_c4 tools/bench_configs.py c4 --steps 20 > "$OUT/traffic_c4.txt" 2>&1; echo "traffic c4 done"
KFILTER=lat_sweep_loop bash tools/pmc_sq.sh r04_c4 tools/bench_configs.py c4 --steps 20 > "$OUT/sq_c4.json" 2> "$OUT/sq_c4.err"; echo "sq c4 done"
bash tools/pmc_traffic.sh r04_c5 tools/bench_configs.py c5 --steps 10 > "$OUT/traffic_c5.txt" 2>&1; echo "traffic c5 done"
KFILTER=pk_sweep_uni bash tools/pmc_sq.sh r04_c5 tools/bench_configs.py c5 --steps 10 > "$OUT/sq_c5.json" 2> "$OUT/sq_c5.err"; echo "sq c5 done"
bash tools/pmc_traffic.sh r04_real tools/real_bench.py 20 "2048^2 gaussian x128" > "$OUT/traffic_real.txt" 2>&1; echo "traffic real done"
KFILTER=rj_sweep bash tools/pmc_sq.sh r04_real tools/real_bench.py 20 "2048^2 gaussian x128" > "$OUT/sq_real.json" 2> "$OUT/sq_real.err"; echo "sq real done"
python3 tools/evidence_r04.py "$OUT" > "$OUT/condense.log" 2>&1; cat "$OUT/condense.log"
# the configurations' lines WITH the fresh counter files in place
python3 tools/bench_configs.py c3 c4 c5 > "$OUT/configs.jsonl" 2> "$OUT/configs.err"; cat "$OUT/configs.jsonl"
python3 tools/real_bench.py 50 > "$OUT/real_bench.jsonl" 2> "$OUT/real_bench.err"; tail -8 "$OUT/real_bench.jsonl"
