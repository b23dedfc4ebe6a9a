#!/usr/bin/env python3
"""Where the time of a short timed region goes: reads a rocprofv3 kernel trace of `bench.py --steps K` and prints,
for every gap of more than 3 us between consecutive kernels, the kernels on both sides and the gap.
usage: region_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    if prev is not None:
        gap = (s - prev[1]) / 1e3
        if gap > 3.0:
            print(f"{gap:9.1f} us idle   after {prev[2]:<62} before {name}  (that kernel: {(e - s) / 1e3:.1f} us)")
    prev = (s, e, name)
