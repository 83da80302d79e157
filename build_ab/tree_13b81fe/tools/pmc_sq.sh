#!/bin/bash
# SQ counter passes for the sweep kernel (per-launch means printed as JSON): tools/pmc_sq.sh <tag> [<python script> [args]]
# default command: bench.py --steps 20 --warmup 2 --no-cpu-baseline; KFILTER=<substring of the kernel name> (default lat_sweep)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_sq_$1; shift
if [ $# -eq 0 ]; then set -- bench.py --steps 20 --warmup 2 --no-cpu-baseline; fi
export KFILTER=${KFILTER:-lat_sweep}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/$1" "${@:2}" > "$OUT/p$i.log" 2>&1
  echo "pass $i done"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))  # kernel name (up to its argument list) -> counter -> values
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if os.environ["KFILTER"] in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {name: {k: sum(v) / len(v) for k, v in sorted(c.items())} for name, c in sorted(acc.items())}
print(json.dumps(next(iter(out.values())) if len(out) == 1 else out, indent=1))
PY
