#!/usr/bin/env python
"""Kernels of a rocprofv3 kernel trace longer than MIN us, in time order, over a window of WINDOW ms ending BACK ms before the
trace's end (what a multi-view iteration of the seg step is made of).  Usage: trace_long_kernels.py DIR [min_us=80] [window_ms=40] [back_ms=60]"""
import csv, glob, sys
src = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 80.0
window = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
back = float(sys.argv[4]) if len(sys.argv) > 4 else 60.0
f = src if src.endswith(".csv") else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[-1]["End_Timestamp"])
lo, hi = end - int((back + window) * 1e6), end - int(back * 1e6)
t0 = None
tot = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < lo or s > hi:
        continue
    if t0 is None:
        t0 = s
    name = r["Kernel_Name"].replace("void ", "")[:70]
    tot[name] = tot.get(name, 0) + (e - s)
    if (e - s) / 1e3 >= min_us:
        print("%10.1f %8.1f  q=%-3s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
print("--- totals over the window (ms)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print("%8.3f  %s" % (v / 1e6, k))
