#!/usr/bin/env python3
"""Condense the passes of tools/evidence_r04.sh into the counter files bench.py / tools/bench_configs.py read
(gpurun_out/evidence_r04/profiles/*: copy into profiles/), every one stamped with the binary and source hashes it belongs to."""
import datetime
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import library_sha256, source_sha256  # noqa: E402

out = sys.argv[1]
dst = os.path.join(out, "profiles")
os.makedirs(dst, exist_ok=True)
stamp = {"library_sha256": library_sha256(), "source_sha256": source_sha256(),
         "measured": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ"), "round": 4}


def dump(name, d):
    d = dict(d, **stamp)
    with open(os.path.join(dst, name), "w") as f:
        json.dump(d, f, indent=1)
    print(name, json.dumps(d)[:400])


def load(path):
    try:
        with open(path) as f:
            txt = f.read()
        return json.loads(txt[txt.index("{"):])
    except (OSError, ValueError):
        return None


def traffic_rows(path):
    rows = {}
    try:
        for line in open(path):
            m = re.match(r"^(.{60})\s+launches\s+(\d+)\s+([\d.]+) us\s+read\s+([\d.]+) MB\s+write\s+([\d.]+) MB", line)
            if m:
                rows[m.group(1).strip()] = dict(launches=int(m.group(2)), us=float(m.group(3)), read_mb_x2=float(m.group(4)), write_mb=float(m.group(5)))
    except OSError:
        pass
    return rows


# ---- headline: tools/profile.sh r04 -> gpurun_out/prof_r04/summary.json
summ = load(os.path.join(ROOT, "gpurun_out", "prof_r04", "summary.json"))
if summ:
    with open(os.path.join(dst, "r04_summary.json"), "w") as f:
        json.dump(dict(summ, **stamp), f, indent=1)
    for name, c in summ["counters_per_launch"].items():
        if "lat_sweep_loop_kernel<false>" in name and "hbm_bytes_per_launch" in c:
            dump("traffic_latest.json", {
                "source": "profiles/r04_summary.json (tools/evidence_r04.sh -> tools/profile.sh r04: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, of "
                          "bench.py --steps 50; read bytes = 2 x FETCH_SIZE x 1024 per MI355X_MICROARCH.md HBM section).  One colour half-sweep of the 256 "
                          "replicas = TWO dispatches of 128 replicas (two stream lanes): the figure is 2 x the per-dispatch mean",
                "kernel": "lat_sweep_loop_kernel<false>", "dispatches_per_launch": 2, "hbm_bytes_per_dispatch": c["hbm_bytes_per_launch"],
                "hbm_bytes_per_launch": 2 * c["hbm_bytes_per_launch"], "read_bytes": 2 * c["read_bytes_corrected"], "write_bytes": 2 * c["write_bytes"]})
sq = load(os.path.join(out, "sq_c2.json"))
if sq:
    if "SQ_WAVES" not in sq:  # several instantiations matched: take the uniform-J looping kernel
        sq = next(v for k, v in sq.items() if "Lb0E" in k or "<false>" in k)
    waves = sq["SQ_WAVES"]
    dump("sq_latest.json", {
        "source": "tools/evidence_r04.sh: rocprofv3 --pmc SQ passes of `bench.py --steps 20`, per DISPATCH of lat_sweep_loop_kernel<false> (128 replicas = one of the "
                  "two stream lanes; SQ_WAVES waves of 64 lanes x 2 quads): SQ_INSTS_VALU / wave-quads; SQ_THREAD_CYCLES_VALU x 4 (quad-cycle units) / 64 lanes / wave-quads",
        "kernel": "lat_sweep_loop_kernel<false>", "waves_per_dispatch": waves,
        "valu_insts_per_quad": sq["SQ_INSTS_VALU"] / (2 * waves), "salu_insts_per_quad": sq["SQ_INSTS_SALU"] / (2 * waves),
        "valu_busy_cycles_per_quad": sq["SQ_THREAD_CYCLES_VALU"] * 4 / 64 / (2 * waves), "raw": sq})

# ---- c3 / c4 / c5 / real path: tools/pmc_traffic.sh tables.  Reads = 2 x FETCH_SIZE x 1024 for EVERY kernel here: the guide's
# correction is stated for 16-byte-per-lane streaming loads, and tools/ubench/fetch_calib.hip (round 4, profiles/r04_fetch_calibration.txt)
# measures exactly the same factor for the packed kernels' 4-byte-per-lane loads (FETCH_SIZE x 1024 = 0.50001 of 1 GiB read once, both
# widths).  Rounds 2-3 took the RAW figure for those kernels and under-reported their read traffic by 2.
# (tag, kernel-name needle, timesteps the matched dispatches cover in the script's command, attempts per timestep)
RUNS = (("c3", "lat_strip_kernel<false, 4, true>", 240, 64 * 1024 * 1024, "bench_configs.py c3 --steps 200: the 40- and the 200-timestep ladder calls (exchange rounds in the launch)"),
        ("c4", "lat_sweep_loop_kernel<true>", 30, 128 * 2048 * 2048, "bench_configs.py c4 --steps 20: 10 + 20 timesteps x 2 colours, one dispatch of 128 replicas each"),
        ("c5", "pk_sweep_uni_kernel", 12, 64 * 256 ** 3, "bench_configs.py c5 --steps 10: 2 + 10 timesteps x 2 classes (x 2 group lanes in the timed call)"),
        ("real", "rj_sweep_kernel", 65, 128 * 2048 * 2048, "real_bench.py 20 '2048^2 gaussian x128': 3 + 2 x 20 + 2 + 20 timesteps x 2 classes (x 2 group lanes in the timed calls)"))
for tag, needle, timesteps, per_step, what in RUNS:
    rows = traffic_rows(os.path.join(out, f"traffic_{tag}.txt"))
    hit = [(k, v) for k, v in rows.items() if needle in k]
    if not hit:
        print("no rows for", tag, list(rows)[:5])
        continue
    k, v = max(hit, key=lambda kv: kv[1]["launches"])
    read, write = v["read_mb_x2"] * 1e6, v["write_mb"] * 1e6
    total = v["launches"] * (read + write)
    dump(f"traffic_{tag}.json", {"source": f"tools/evidence_r04.sh: tools/pmc_traffic.sh r04_{tag} ({what}); read = 2 x FETCH_SIZE x 1024 (calibrated: "
                                           "profiles/r04_fetch_calibration.txt), write = WRITE_SIZE x 1024",
                                 "kernel": k, "dispatches": v["launches"], "avg_us": v["us"], "read_bytes_per_dispatch": read, "write_bytes_per_dispatch": write,
                                 "timesteps_covered": timesteps, "attempts_covered": timesteps * per_step, "hbm_bytes_total": total,
                                 "hbm_bytes_per_attempt": total / (timesteps * per_step)})
    sq_ = load(os.path.join(out, f"sq_{tag}.json"))
    if sq_:
        dump(f"r04_sq_{tag}.json", {"source": f"tools/evidence_r04.sh: tools/pmc_sq.sh r04_{tag}, per-dispatch means", "kernel_filter": needle, "counters": sq_})
