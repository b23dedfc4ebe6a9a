#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh) into tracked files:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (all launches)
   profiles/<tag>_summary.json       per-kernel mean duration of full-work launches, HBM-side
                                     traffic per launch from the PMC passes, and the bench line.
FETCH_SIZE on gfx950 counts 64 B per 128-B request of a wide coalesced read, i.e. half the
bytes; it is doubled here (calibrated on k_dot, which reads a known 4 N bytes once).
WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Both are in KiB."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return g[-1] if g else None   # the newest run if several were merged into the folder


shutil.copy(one("stats/*/*kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
trace = list(csv.DictReader(open(one("stats/*/*kernel_trace.csv"))))
out = {"tag": tag, "kernels": {}}
names = sorted({r["Kernel_Name"] for r in trace if "cgamd::k_" in r["Kernel_Name"]})
pmc = {}
for kind in ("fetch", "write"):
    f = one(f"{kind}/*/*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            pmc.setdefault((kind, r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
for n in names:
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace if r["Kernel_Name"] == n]
    # drop no-op launches (past maxit or after a stop).  Kernel B launches of a solver that updates x in
    # pairs alternate between two amounts of work (3 and 6 words per point): both are full work, the
    # means below are over whole pairs
    # (against the 90th percentile, not the maximum: a first launch can take several times as long)
    ref = sorted(dur)[int(0.9 * (len(dur) - 1))]
    full = [d for d in dur if 0.2 * ref < d < 3 * ref]
    k = {"calls": len(dur), "full_work_calls": len(full), "mean_us_full_work": round(statistics.mean(full) / 1e3, 3),
         "min_us": round(min(dur) / 1e3, 3)}
    for kind in ("fetch", "write"):
        v = [x for x in pmc.get((kind, n), []) if x > 0.2 * max(pmc[(kind, n)])]
        if v:
            kib = statistics.mean(v)
            k[f"{kind}_size_KiB_raw"] = round(kib, 1)
            k[f"{kind}_bytes"] = int(kib * 1024 * (2 if kind == "fetch" else 1))
    if "fetch_bytes" in k and "write_bytes" in k:
        k["traffic_bytes"] = k["fetch_bytes"] + k["write_bytes"]
    out["kernels"][n] = k
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl) and os.path.getsize(bl):
    out["bench_line"] = json.load(open(bl))
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
