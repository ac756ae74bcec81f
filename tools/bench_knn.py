#!/usr/bin/env python
"""distCUDA2 (iso_dist2_3nn) timing at SfM-cloud sizes: the default query (ring walk from global memory) and the opt-in
LDS-bucketed one (ISO_KNN_LDS=1 in a second process), plus a brute-force check on a subset.  usage: bench_knn.py [--sizes 100000,300000,1000000,1500000]"""
import argparse, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instascene_amd.knn import distCUDA2

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="100000,300000,1000000,1500000")
ap.add_argument("--child", action="store_true")
a = ap.parse_args()
out = {}
for P in (int(x) for x in a.sizes.split(",")):
    for dist in ("uniform", "clustered"):
        g = torch.Generator().manual_seed(P)
        pts = torch.rand(P, 3, generator=g) * 3 - 1.5
        if dist == "clustered":          # an SfM-like cloud: dense surfaces + sparse outliers
            pts[: P * 9 // 10] = (torch.randn(P * 9 // 10, 3, generator=g) * torch.tensor([0.6, 0.05, 0.6])
                                  + torch.randint(-2, 3, (P * 9 // 10, 1), generator=g).float() * 0.4)
        pts = pts.cuda()
        d = distCUDA2(pts); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            d = distCUDA2(pts)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        # brute force on 256 query points
        idx = torch.randint(0, P, (256,), generator=g).cuda()
        dd = ((pts[idx][:, None, :] - pts[None, :, :]) ** 2).sum(-1)
        dd[torch.arange(256, device="cuda"), idx] = float("inf")
        ref = dd.topk(3, dim=1, largest=False).values.mean(1)
        ok = bool(torch.allclose(d[idx], ref, rtol=1e-6, atol=0))
        out[f"{P}:{dist}"] = {"ms": round(ms, 3), "Mpoints_per_s": round(P / ms / 1e3, 1), "matches_brute_force": ok}
if a.child:
    print(json.dumps(out))
else:
    env = dict(os.environ, ISO_KNN_LDS="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--sizes", a.sizes], env=env, capture_output=True, text=True)
    old = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-500:]}
    print(json.dumps({"default_global_ring_walk": out, "opt_in_lds_bucketed": old}, indent=1))
