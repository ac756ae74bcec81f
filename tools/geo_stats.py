#!/usr/bin/env python
"""Lane-utilisation table of k_render_bwd_geo (the train.py step's dominant kernel): one RgbTrainer step with the STATS build's
counters armed (isr_backward_set_counters), printed as JSON.

    python tools/geo_stats.py [C2|C3|C1] [mode]
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from instascene_amd import rasterizer, scenes  # noqa: E402
from instascene_amd._lib import lib  # noqa: E402
from instascene_amd.harness import RgbTrainer  # noqa: E402
from instascene_amd.render import render  # noqa: E402


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "C2"
    mode = sys.argv[2] if len(sys.argv) > 2 else "fast_reflists"
    rasterizer.set_mode(mode)
    dev = torch.device("cuda", 0)
    scene, cams, cfg = scenes.config_scene(config)
    scene.seg_feature = None
    tr = RgbTrainer(scene, cams[:4], [torch.zeros(3, 8, 8)] * 4, device=dev)
    with torch.no_grad():
        g = torch.Generator(device=dev).manual_seed(5)
        tr.targets = [(render(c, tr.model, tr.pipe, tr.bg)["render"] + 0.05 * torch.randn(3, cfg["H"], cfg["W"], device=dev, generator=g)).clamp(0, 1)
                      for c in tr.cams]
    L = lib()
    for it in range(3):
        tr.step(it)
    torch.cuda.synchronize()
    counters = torch.zeros(16, dtype=torch.int64, device=dev)
    fwd = torch.zeros(16, dtype=torch.int64, device=dev)
    L.isr_forward_set_counters(ctypes.c_void_p(fwd.data_ptr()))
    L.isr_backward_set_counters(ctypes.c_void_p(counters.data_ptr()))
    tr.step(3)
    torch.cuda.synchronize()
    c = [int(v) for v in counters.tolist()]
    f = [int(v) for v in fwd.tolist()]
    names = ["chunks", "splat_slots_filled", "chunk_rows", "chunk_rows_reached", "iterations_with_a_candidate", "candidate_lanes",
             "iterations_with_a_blending_lane", "blending_lanes", "chunks_with_a_pre_evaluated_exact_splat", "partial_rows_stored"]
    out = dict(zip(names, c))
    np_wave = 64 if cfg["W"] * cfg["H"] > 12000 * 64 else 32          # pixels per wave: two waves per block on small grids
    out.update({
        "config": config, "mode": mode, "R": int(rasterizer.LAST_NUM_RENDERED),
        "forward_wave_splat_pairs_evaluated": f[1], "forward_blending_lanes": f[3],
        "slot_fill": round(c[1] / max(1, 64 * c[0]), 4),
        "pixel_iterations_per_chunk": round(c[4] / max(1, c[0]), 2),
        "candidate_lane_fraction": round(c[5] / max(1, 64 * c[4]), 4),
        "blending_iterations_of_candidate_iterations": round(c[6] / max(1, c[4]), 4),
        "blending_lane_fraction_of_blending_iterations": round(c[7] / max(1, 64 * c[6]), 4),
        "rows_skipped_fraction": round(1.0 - c[3] / max(1, c[2]), 4),
    })
    print(json.dumps(out))


if __name__ == "__main__":
    main()
