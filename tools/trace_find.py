#!/usr/bin/env python
"""List the launches of a rocprofv3 kernel_trace.csv whose name contains PATTERN and that ran longer than MIN_US, with the
step (count of geometry passes before them), queue and what ran right before / after on the same queue.
usage: trace_find.py DIR_OR_CSV PATTERN [MIN_US]"""
import csv, glob, sys
src, pat = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
f = src if src.endswith(".csv") else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
step = 0
byq = {}
for i, r in enumerate(rows):
    if "k_preprocess" in r["Kernel_Name"] and "bwd" not in r["Kernel_Name"]:
        step += 1
    r["_step"] = step
    byq.setdefault(r.get("Queue_Id", "?"), []).append(i)
print("total launches", len(rows), "geometry passes", step)
for q, idx in byq.items():
    for k, i in enumerate(idx):
        r = rows[i]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if pat in r["Kernel_Name"] and d >= min_us:
            prev = rows[idx[k - 1]]["Kernel_Name"][:40] if k else "-"
            nxt = rows[idx[k + 1]]["Kernel_Name"][:40] if k + 1 < len(idx) else "-"
            print("t=%10.1f us  dur %8.1f us  q=%s  step %d  after [%s]  before [%s]" % (
                (int(r["Start_Timestamp"]) - t0) / 1e3, d, q, r["_step"], prev, nxt))
