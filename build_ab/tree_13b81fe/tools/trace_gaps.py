#!/usr/bin/env python3
"""Per-kernel durations and the gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV.
  python tools/trace_gaps.py <dir or kernel_trace.csv> [skip_first_n]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
rows = rows[skip:]
dur = defaultdict(list)
gap_after = defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    dur[n].append(e - s)
    if i + 1 < len(rows):
        gap_after[n].append(rows[i + 1][0] - e)
total = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print(f"{len(rows)} kernels, span {total / 1e3:.1f} us, busy {busy / 1e3:.1f} us ({100 * busy / total:.1f} %)")
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[n], gap_after[n] or [0]
    print(f"{n:60s} calls {len(d):6d}  avg {sum(d) / len(d) / 1e3:8.2f} us  gap after avg {sum(g) / len(g) / 1e3:7.2f} us  max {max(g) / 1e3:8.1f}")
