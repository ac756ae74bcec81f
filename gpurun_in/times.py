import ctypes, json, os, sys, shutil
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
shutil.copy(R + "/gpurun_in/lib_dbgt.so", R + "/instascene_amd/libinstascene_hip.so")
import torch
from instascene_amd import harness, scenes
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "6", "--warmup", "3", "--view-cache-gb", "8"]
bench.main()
torch.cuda.synchronize()
L = ctypes.CDLL(R + "/instascene_amd/libinstascene_hip.so")
buf = np.zeros(8 * 16384, dtype=np.uint64)
rc = L.isr_debug_times(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(-1, 8)
t = t[t[:, 0] > 0]
b, p, e = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.int64)
leneff = (t[:, 3] >> np.uint64(32)).astype(np.int64); ns = (t[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
t0 = b.min()
live = p > 0
print("rc", rc, "tiles", len(t), "with work", live.sum())
tick = 0.01  # us per tick at 100 MHz
print("kernel span (first begin -> last end) us:", (max(e.max(), b.max()) - t0) * tick)
print("begin times pct us:", np.percentile((b - t0) * tick, [0, 10, 50, 90, 99, 100]))
d = (e[live] - b[live]) * tick
print("wave lifetime us pct:", np.percentile(d, [0, 10, 50, 90, 99, 100]), "mean", d.mean())
pr = (p[live] - b[live]) * tick
print("prologue us pct:", np.percentile(pr, [0, 10, 50, 90, 99, 100]), "mean", pr.mean())
print("end times pct us:", np.percentile((e[live] - t0) * tick, [0, 10, 50, 90, 99, 100]))
print("len_eff pct:", np.percentile(leneff[live], [0, 10, 50, 90, 99, 100]), "samples/tile pct:", np.percentile(ns[live], [0, 50, 90, 99, 100]))
idx = np.argsort(-d)[:10]
print("slowest:", [(float(d[i]), int(leneff[live][i]), int(ns[live][i]), float((b[live][i]-t0)*tick)) for i in idx])
tl = t[live]
m_, c_, s_ = tl[:, 4].astype(np.int64) * tick, tl[:, 5].astype(np.int64) * tick, tl[:, 6].astype(np.int64) * tick
p1 = (tl[:, 7] >> np.uint64(32)).astype(np.int64) * tick; nch = (tl[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
print("mean per tile us: lifetime %.2f prologue %.2f pass1 %.2f memwait %.2f compute %.2f store-issue %.2f chunks %.2f" % (d.mean(), pr.mean(), p1.mean(), m_.mean(), c_.mean(), s_.mean(), nch.mean()))
print("per chunk us: memwait %.2f compute %.2f store-issue %.2f" % (m_.sum() / nch.sum(), c_.sum() / nch.sum(), s_.sum() / nch.sum()))
for lo, hi in [(1, 1), (2, 2), (3, 4), (5, 9)]:
    mm = (ns[live] >= lo) & (ns[live] <= hi)
    if mm.any(): print("samples %d-%d: tiles %d lifetime %.1f chunks %.1f memwait/chunk %.2f compute/chunk %.2f store/chunk %.2f" % (lo, hi, mm.sum(), d[mm].mean(), nch[mm].mean(), m_[mm].sum() / max(1, nch[mm].sum()), c_[mm].sum() / max(1, nch[mm].sum()), s_[mm].sum() / max(1, nch[mm].sum())))
# concurrency over time
ev = np.concatenate([b[live] - t0, e[live] - t0]); sg = np.concatenate([np.ones(live.sum()), -np.ones(live.sum())])
o = np.argsort(ev, kind="stable"); c = np.cumsum(sg[o]); tt = ev[o] * tick
for q in range(0, int(tt.max()) + 1, 10):
    m = (tt >= q) & (tt < q + 10)
    if m.any(): print("t=%3d..%3d us resident waves mean %.0f" % (q, q + 10, c[m].mean()))
