#!/bin/bash
# HBM traffic of every kernel of a command (FETCH_SIZE / WRITE_SIZE in separate passes): tools/pmc_traffic.sh <tag> <python script> [args]
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/traffic_$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -- python3 "$ROOT/$1" "${@:2}" > "$OUT/$c.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(os.path.join(sys.argv[1], c, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for path in glob.glob(os.path.join(sys.argv[1], c, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in acc.items():
    rd = 2 * 1024 * sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1); wr = 1024 * sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    us = sum(dur[k]) / len(dur[k]) / 1e3
    print(f"{k:60s} launches {len(v['FETCH_SIZE']):5d}  {us:9.1f} us  read {rd/1e6:9.1f} MB  write {wr/1e6:8.1f} MB  => {(rd+wr)/us/1e6:6.2f} TB/s")
PY
