#!/usr/bin/env python3
"""Side-by-side table of the counters tools/pmc_passes.sh collected for kernels A and B:
median over the full-work launches of each kernel.  Writes profiles/<tag>_pmc.json.
usage: tools/pmc_table.py <tag>"""
import csv
import glob
import json
import os
import re
import statistics
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"pmc_{tag}")
table = {}
durs = {}
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"cgamd::(k_\w+)<(\d),", row.get("Kernel_Name", ""))
        if not m or m.group(2) not in "12":
            continue
        k = ("A " if m.group(2) == "1" else "B ") + m.group(1)
        table.setdefault(row["Counter_Name"], {}).setdefault(k, []).append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(src, "sq1", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"cgamd::(k_\w+)<(\d),", row.get("Kernel_Name", ""))
        if m and m.group(2) in "12":
            k = ("A " if m.group(2) == "1" else "B ") + m.group(1)
            durs.setdefault(k, []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
kernels = sorted({k for v in table.values() for k in v})
out = {"tag": tag, "kernels": kernels, "counters": {}, "duration_us_under_counters": {}}
for k in kernels:
    if k in durs:
        full = [d for d in durs[k] if d > 0.5 * max(durs[k])]
        out["duration_us_under_counters"][k] = round(statistics.median(full) / 1e3, 1)
print(f"{'counter':40s}" + "".join(f"{k:>22s}" for k in kernels))
for c in sorted(table):
    vals = []
    for k in kernels:
        v = table[c].get(k, [])
        full = [x for x in v if x >= 0.5 * max(v)] if v and max(v) > 0 else v
        vals.append(statistics.median(full) if full else float("nan"))
    out["counters"][c] = dict(zip(kernels, vals))
    print(f"{c:40s}" + "".join(f"{v:22.4g}" for v in vals))
print("duration under counters (us):", out["duration_us_under_counters"])
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
