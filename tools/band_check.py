"""Guard-band evidence (GPU box): for N fuzz scenes (tests/test_gpu_fuzz.py's generator; cases first .. first + N - 1) run the FAST
forward with the STATS build of the blend kernel and compare its decisions with the CPU oracle's.

Per scene: (wave, splat) evaluations, how many took the EXACT path, counter 7 = pairs OUTSIDE the guard bands whose decision differs
from EXACT's (must be 0: the band's bound holds), pixels whose last / median contributor differs from the oracle's, and how many of
those are explained by the one decision FAST cannot replay (the T < 1e-4 stop; oracle margin [3] below T_TOL) or by the oracle's
second build (FMA contraction + libm expf) disagreeing with the first on that pixel.
Usage: python tools/band_check.py [N=40] [first=0] > report.jsonl"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("ISR_MODE", "exact")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_gpu_fuzz as Z  # noqa: E402
from helpers import oracle_forward  # noqa: E402
from instascene_amd import _lib  # noqa: E402

T = Z.T
T_TOL = 1e-4


def check(inp, cam, tag):
    st = oracle_forward(inp, cam, margins=True)
    st2 = oracle_forward(inp, cam, fma=True)
    counters = torch.zeros(16, dtype=torch.int64, device="cuda")
    _lib.lib().isr_forward_set_counters(ctypes.c_void_p(counters.data_ptr()))
    args, out = T.hip_forward(inp, cam, mode=T.MODE_FAST)
    torch.cuda.synchronize()
    c = counters.tolist()
    dbg = T.rz.debug_state(st["P"], st["W"], st["H"], out[0], out[5], out[6], out[7])
    N = st["W"] * st["H"]
    diff = (dbg["n_contrib"] != st["n_contrib"]).any(axis=0).reshape(-1)
    img_bad = np.zeros(N, bool)
    for got, want in ((out[1], st["color"]), (out[2][:5], st["others"][:5]), (out[4], st["extra"])):
        if want.size == 0:
            continue
        g = got.cpu().numpy().reshape(want.shape[0], -1)
        w = want.reshape(want.shape[0], -1)
        sc = np.abs(w).max(axis=1, keepdims=True) + 1e-30
        img_bad |= (np.abs(g - w) > 1e-4 * sc).any(axis=0)
    tm = np.minimum(st["margins"][3], st["margins"][4])          # the two T decisions (stop at 1e-4, median at 0.5)
    tstop = tm < T_TOL
    two = (st2["n_contrib"] != st["n_contrib"]).any(axis=0)
    bad = diff | img_bad
    rec = dict(tag=tag, P=st["P"], N=N, R=st["R"], evals=c[1], blends=c[2], exact_path=c[6], outside_band_decision_differs=c[7],
               px_contrib_differs=int(diff.sum()), px_image_beyond_1e4=int(img_bad.sum()),
               explained_T_stop=int((bad & tstop).sum()), explained_two_builds=int((bad & ~tstop & two).sum()),
               unexplained=int((bad & ~tstop & ~two).sum()), px_two_builds_differ=int(two.sum()),
               largest_T_margin_of_a_differing_pixel=float(tm[bad].max()) if bad.any() else 0.0,
               forced_splats=int(np.isinf(dbg["records"][:, 19][st["radii"] > 0]).sum()), visible=int((st["radii"] > 0).sum()))
    print(json.dumps(rec), flush=True)
    return rec


def main():
    if len(sys.argv) > 1 and sys.argv[1] in ("C1", "C2", "C3", "C5"):          # a full-size bench scene, one view
        from instascene_amd import scenes
        sc, cams, cfg = scenes.config_scene(sys.argv[1])
        check(scenes.activated_inputs(sc), cams[int(sys.argv[2]) if len(sys.argv) > 2 else 0], sys.argv[1])
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tot = {}
    for case in range(first, first + n):
        inp, cam, F = Z._scene(case)
        r = check(inp, cam, f"fuzz{case}")
        for k, v in r.items():
            if isinstance(v, int):
                tot[k] = tot.get(k, 0) + v
    print(json.dumps(dict(tag="TOTAL", scenes=n, **tot)))


if __name__ == "__main__":
    main()
