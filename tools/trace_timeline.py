#!/usr/bin/env python
"""Kernel-by-kernel timeline of the last full step of a rocprofv3 kernel_trace.csv (start offset, duration, gap)."""
import csv, glob, sys
src = sys.argv[1]
f = src if src.endswith(".csv") else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_preprocess" in r[2] and "bwd" not in r[2]]
lo, hi = marks[-2], marks[-1]
t0, end = rows[lo][0], rows[lo][0]
for s, e, k in rows[lo:hi]:
    print("%9.1f us  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3, k[:80]))
    end = max(end, e)
