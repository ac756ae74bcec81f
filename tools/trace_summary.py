#!/usr/bin/env python
"""Per-step kernel breakdown from a rocprofv3 kernel_trace.csv of `bench.py`.

usage: trace_summary.py <dir-or-csv> [n_last_steps]
A step starts at each k_preprocess launch; the last n steps are averaged."""
import collections, csv, glob, sys
src = sys.argv[1]
f = src if src.endswith(".csv") else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_preprocess" in r[2] and "bwd" not in r[2]]
lo, hi = marks[-n - 1], marks[-1]
sel = rows[lo:hi]
busy = sum(e - s for s, e, _ in sel)
span = rows[hi][0] - rows[lo][0]
print("%d steps: kernel-busy %.1f us/step, wall span %.1f us/step, %d launches/step" % (n, busy / n / 1e3, span / n / 1e3, len(sel) // n))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in sel:
    agg[k[:90]][0] += e - s
    agg[k[:90]][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%9.1f us/step  n/step %5.1f  %s" % (t / n / 1e3, c / n, k))
# idle time on the device, attributed to the kernel that follows the gap
gaps = collections.defaultdict(lambda: [0, 0])
end = sel[0][1]
for s, e, k in sel[1:]:
    if s > end:
        gaps[k[:70]][0] += s - end
        gaps[k[:70]][1] += 1
    end = max(end, e)
print("idle before kernel (us/step, count/step):")
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%9.1f  %5.1f  %s" % (t / n / 1e3, c / n, k))
