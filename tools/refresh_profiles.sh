#!/bin/bash
# Collect the evidence kept under profiles/ (round 2): bench lines, rocprofv3 kernel stats + per-step trace of the default
# bench command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE, each alone; --pmc with --kernel-trace only), the blend
# kernel's issue counters.  Usage (on the GPU box, from the repo root):  bash tools/refresh_profiles.sh TAG  ->
# gpurun_out/prof_TAG/ ; then  python tools/make_traffic_json.py gpurun_out/prof_TAG  rebuilds profiles/roofline_traffic.json
set -u
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py 2>$OUT/bench_c3.err | tail -1 > $OUT/bench_c3.json
# (the 779x519 train.py step is host-bound - 1.05 ms of enqueue work for 0.95 ms of kernels - and each bench ends with a
# 128-thread CPU baseline: let the host cores cool down, or the next headline block runs at 1.25 ms per step)
sleep 45
python $ROOT/bench.py --config C2 2>$OUT/bench_c2.err | tail -1 > $OUT/bench_c2_rgb.json
sleep 45
python $ROOT/bench.py --config C3 --step rgb --no-cpu-baseline --submodes exact 2>/dev/null | tail -1 > $OUT/bench_c3_rgb.json
python $ROOT/bench.py --config C5 --no-cpu-baseline --submodes "" 2>/dev/null | tail -1 > $OUT/bench_c5.json
for CFG in "C3:seg" "C2:rgb"; do
  C=${CFG%%:*}; S=${CFG#*:}
  rm -rf /tmp/st && rocprofv3 --kernel-trace --stats -d /tmp/st -o st --output-format csv -- python $ROOT/bench.py --config $C --step $S --steps 20 --warmup 5 --no-cpu-baseline --submodes "" > $OUT/stats_run_$C.log 2>&1
  cp $(find /tmp/st -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_${C}_$S.csv
  python $ROOT/tools/trace_summary.py /tmp/st 15 > $OUT/trace_per_step_${C}_$S.txt 2>&1
  python $ROOT/tools/trace_timeline.py /tmp/st > $OUT/timeline_${C}_$S.txt 2>&1
  for K in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$K && rocprofv3 --pmc $K -d /tmp/pmc_$K -o pmc --output-format csv -- python $ROOT/bench.py --config $C --step $S --steps 3 --warmup 2 --no-cpu-baseline --submodes "" > $OUT/pmc_run.log 2>&1
    python $ROOT/tools/pmc_summary.py /tmp/pmc_$K > $OUT/pmc_${K}_${C}_$S.txt 2>&1
  done
done
bash $ROOT/tools/pmc_fwd.sh ${TAG}_fwd > /dev/null 2>&1; cp $ROOT/gpurun_out/pmc_${TAG}_fwd.txt $OUT/pmc_issue_k_render_fwd.txt
KERNEL=k_render_bwd_geo RASTER_ARGS="--config C2 --iters 3 --geom 1" bash $ROOT/tools/pmc_fwd.sh ${TAG}_geo > /dev/null 2>&1; cp $ROOT/gpurun_out/pmc_${TAG}_geo.txt $OUT/pmc_issue_k_render_bwd_geo.txt
python $ROOT/tools/host_overhead.py > $OUT/host_overhead.txt 2>&1
ls -la $OUT
