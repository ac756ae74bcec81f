#!/bin/bash
# Collect the evidence kept under profiles/ (round 3; rerun after the forward blend was re-decomposed: TAG r03b): the default bench line (with all sub_records), timeout 500 rocprofv3 kernel
# stats + per-step trace of the headline and of C2 rgb / C5 seg / the multi-view step, PMC traffic passes (FETCH_SIZE /
# WRITE_SIZE, each alone; --pmc with --kernel-trace only), the blend kernels' issue counters, kernels timed alone.
# Usage (on the GPU box, from the repo root):  bash tools/refresh_profiles.sh TAG  -> gpurun_out/prof_TAG/ ; then
# python tools/make_traffic_json.py gpurun_out/prof_TAG TAG  rebuilds profiles/roofline_traffic.json
set -u
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 20 --warmup 5 2>$OUT/bench_c3.err | tail -1 > $OUT/bench_c3.json
ONE="--no-cpu-baseline --submodes= --more 0"
for CFG in "C3:seg:" "C2:rgb:" "C5:seg:" "C3:seg:--multiview=1"; do
  C=${CFG%%:*}; R=${CFG#*:}; S=${R%%:*}; X=${R#*:}
  N=${C}_${S}$( [ -n "$X" ] && echo _multiview )
  rm -rf /tmp/st && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/st -o st --output-format csv -- python $ROOT/bench.py --config $C --step $S --steps 20 --warmup 10 $ONE $X > $OUT/stats_run_$N.log 2>&1
  cp $(find /tmp/st -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$N.csv
  python $ROOT/tools/trace_summary.py /tmp/st 15 > $OUT/trace_per_step_$N.txt 2>&1
  python $ROOT/tools/trace_timeline.py /tmp/st > $OUT/timeline_$N.txt 2>&1
  [ -n "$X" ] && continue
  for K in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$K && timeout 500 rocprofv3 --pmc $K -d /tmp/pmc_$K -o pmc --output-format csv -- python $ROOT/bench.py --config $C --step $S --steps 3 --warmup 2 $ONE > $OUT/pmc_run.log 2>&1
    python $ROOT/tools/pmc_summary.py /tmp/pmc_$K > $OUT/pmc_${K}_${C}_$S.txt 2>&1
  done
done
bash $ROOT/tools/pmc_fwd.sh ${TAG}_fwd > /dev/null 2>&1; cp $ROOT/gpurun_out/pmc_${TAG}_fwd.txt $OUT/pmc_issue_k_render_fwd.txt
KERNEL=k_render_bwd_geo RASTER_ARGS="--config C2 --iters 3 --geom 1" bash $ROOT/tools/pmc_fwd.sh ${TAG}_geo > /dev/null 2>&1; cp $ROOT/gpurun_out/pmc_${TAG}_geo.txt $OUT/pmc_issue_k_render_bwd_geo.txt
for C in C3 C5 C2; do python $ROOT/tools/fwd_ab.py --config $C 2>/dev/null | tail -1; done > $OUT/kernels_alone.txt
python $ROOT/tools/host_overhead.py > $OUT/host_overhead.txt 2>&1
python $ROOT/tools/soak_train.py --config C2 --blocks 4 --block 500 > $OUT/soak_C2_rgb.txt 2>&1
python $ROOT/tools/soak_train.py --config C3 --step seg --blocks 4 --block 500 > $OUT/soak_C3_seg.txt 2>&1
ls -la $OUT
