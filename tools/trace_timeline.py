#!/usr/bin/env python
"""Kernel-by-kernel timeline of ONE step in the middle of a rocprofv3 kernel_trace.csv: start offset, duration, hardware
queue, kernel.  (The last steps of a bench run hold its untimed extras, so a step eight before the end is printed.)"""
import csv, glob, sys
src = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = src if src.endswith(".csv") else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_preprocess" in r["Kernel_Name"] and "bwd" not in r["Kernel_Name"]]
back = min(back, len(marks) - 1)
lo, hi = marks[-back - 1], marks[-back]
t0 = int(rows[lo]["Start_Timestamp"])
print("   start us   dur us  queue  kernel   (one step: from one geometry pass to the next)")
for r in rows[lo:hi + 2]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f %8.1f  q=%-3s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
