#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of tools/profile.sh into one JSON summary (committed under profiles/).

HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB-like units of
1024 B (bytes = value * 1024), and on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads
by exactly 2x => read bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return hits[0] if hits else None


def kernel_stats(d):
    path = find(d, "*kernel_stats.csv")
    rows = []
    if path:
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append({"name": r.get("Name"), "calls": int(r.get("Calls", 0)),
                             "total_ns": float(r.get("TotalDurationNs", 0)),
                             "avg_ns": float(r.get("AverageNs", 0)), "pct": float(r.get("Percentage", 0))})
    return rows


def counters(d):
    """mean counter value per dispatch, per kernel name"""
    path = find(d, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if path:
        with open(path) as f:
            for r in csv.DictReader(f):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    summary = {"tag": tag, "kernel_stats": kernel_stats(os.path.join(out_dir, "trace"))}
    fetch = counters(os.path.join(out_dir, "pmc_fetch"))
    write = counters(os.path.join(out_dir, "pmc_write"))
    sq = counters(os.path.join(out_dir, "pmc_sq"))
    per_kernel = {}
    for name in set(fetch) | set(write) | set(sq):
        e = {}
        if name in fetch and "FETCH_SIZE" in fetch[name]:
            e["FETCH_SIZE_raw"] = fetch[name]["FETCH_SIZE"]
            e["read_bytes_corrected"] = 2 * fetch[name]["FETCH_SIZE"] * 1024
        if name in write and "WRITE_SIZE" in write[name]:
            e["WRITE_SIZE_raw"] = write[name]["WRITE_SIZE"]
            e["write_bytes"] = write[name]["WRITE_SIZE"] * 1024
        if "read_bytes_corrected" in e and "write_bytes" in e:
            e["hbm_bytes_per_launch"] = e["read_bytes_corrected"] + e["write_bytes"]
        e.update(sq.get(name, {}))
        per_kernel[name] = e
    summary["counters_per_launch"] = per_kernel
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
