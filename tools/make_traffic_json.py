#!/usr/bin/env python
"""profiles/roofline_traffic.json from the PMC passes of tools/refresh_profiles.sh: HBM bytes per launch and kernel,
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 - FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950
(calibrated in round 1 on torch copy / add kernels over a 192 MB tensor: WRITE_SIZE = 1.00 x bytes, FETCH_SIZE = 0.50 x).
usage: make_traffic_json.py gpurun_out/prof_TAG [tag-for-the-source-note]"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import DEFAULT_MODE as MODE          # the passes run bench.py / run_raster.py in the library's default mode
d = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(d.rstrip("/"))
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `python bench.py --config C --step S --steps 3 "
                 f"--warmup 2 --submodes ''` (tools/refresh_profiles.sh {tag}); bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024; "
                 f"raw counters: profiles/{tag}_pmc_FETCH_SIZE_*.txt, profiles/{tag}_pmc_WRITE_SIZE_*.txt"}
short = {"k_render_fwd_fast_w": "k_render_fwd", "k_render_fwd_fast": "k_render_fwd", "k_render_fwd": "k_render_fwd", "k_pack_hits": "k_pack_hits", "k_render_bwd_geo": "k_render_bwd",
         "k_render_bwd_sparse": "k_render_bwd_sparse", "k_render_bwd": "k_render_bwd", "k_preprocess_bwd": "k_preprocess_bwd",
         "k_preprocess": "k_preprocess", "k_scatter": "k_scatter", "k_tile_sort": "k_tile_sort", "k_tile_sort_wave": "k_tile_sort_wave",
         "k_feature_rows_step": "k_feature_rows_step", "gaussian_adam_kernel": "gaussian_adam_kernel", "ssim_fwd": "ssim_fwd",
         "ssim_bwd": "ssim_bwd", "pp_maps": "pp_maps", "pp_surf_normal": "pp_surf_normal"}
for cfg, step in (("C3", "seg"), ("C2", "rgb"), ("C5", "seg")):
    vals, seen, inst = {}, {}, {}
    for k in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(d, f"pmc_{k}_{cfg}_{step}.txt")
        if not os.path.exists(f):
            continue
        for line in open(f):
            if "|" not in line:
                continue
            name, rest = line.split("|", 1)
            m = re.search(k + r"=([0-9.e+]+)\(n=(\d+)\)", rest)
            if not m:
                continue
            full = re.sub(r"^(void )?(isr|iso)::", "", name.strip())
            base = re.split(r"[<(]", full)[0]
            key = short.get(base)
            if key is None:
                continue
            # several template instances share a key (the STATS-instrumented k_render_fwd_fast<.., true, ..> runs once per bench
            # for the work counters): the record is the instance that was launched most often, i.e. the production kernel
            n = int(m.group(2))
            if re.match(r"k_render_fwd_fast(_w)?<\w+, true", full):
                continue
            if n >= seen.get((key, k), 0):
                seen[(key, k)] = n
                vals.setdefault(key, {})[k] = float(m.group(1))
                inst.setdefault(key, full.split("(")[0])
    rec = {k: int((2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024) for k, v in vals.items()}
    if rec:
        out[f"{cfg}:{step}:{MODE}"] = rec
        out[f"{cfg}:{step}:{MODE}:instances"] = inst
# the blend kernel's vector-instruction count per launch (its binding roofline is instruction issue): tools/pmc_fwd.sh's passes
issue = os.path.join(d, "pmc_issue_k_render_fwd.txt")
if os.path.exists(issue):
    cnt = {}
    for line in open(issue):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            m = re.search(r"\b" + k + r"=([0-9.e+]+)\(n=", line)
            if m and "k_render_fwd_fast_w" in line:
                cnt[k] = float(m.group(1))
    if cnt:
        out[f"C3:seg:{MODE}:k_render_fwd:issue"] = dict(cnt, source=f"rocprofv3 --pmc, one counter group per run, tools/pmc_fwd.sh over tools/run_raster.py "
                                                                  f"--config C3 (profiles/{tag}_pmc_issue_k_render_fwd.txt); wave-instructions per launch")
# the tree the counters were taken on: bench.py prints "traffic_stale": true when csrc/ has changed since
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_tree_hash
out["csrc_tree_hash"] = csrc_tree_hash()
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "roofline_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:2000])
