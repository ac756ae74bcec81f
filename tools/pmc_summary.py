#!/usr/bin/env python
"""Aggregate a rocprofv3 counter_collection.csv per kernel (mean per dispatch)."""
import collections, csv, glob, sys
d = sys.argv[1]
f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k + " | " + " ".join("%s=%.4g(n=%d)" % (c, sum(x) / len(x), len(x)) for c, x in sorted(v.items())))
