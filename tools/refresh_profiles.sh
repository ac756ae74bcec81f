#!/bin/bash
# Collect the evidence kept under profiles/: bench lines (fast / exact), rocprofv3 kernel stats and per-step trace of the
# default bench command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE, each alone).  Usage (on the GPU box, from the repo
# root):  bash tools/refresh_profiles.sh v7     -> gpurun_out/prof_v7/
set -u
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py 2>$OUT/bench_fast.err | tail -1 > $OUT/bench_c3_fast.json
python $ROOT/bench.py --mode exact --no-cpu-baseline 2>$OUT/bench_exact.err | tail -1 > $OUT/bench_c3_exact.json
rm -rf /tmp/st && rocprofv3 --kernel-trace --stats -d /tmp/st -o st --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats_run.log 2>&1
cp $(find /tmp/st -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python $ROOT/tools/trace_summary.py /tmp/st 20 > $OUT/trace_per_step.txt 2>&1
python $ROOT/tools/trace_timeline.py /tmp/st > $OUT/timeline.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C && rocprofv3 --pmc $C -d /tmp/pmc_$C -o pmc --output-format csv -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/pmc_$C > $OUT/pmc_$C.txt 2>&1
done
python $ROOT/tools/bench_rgb.py --config C3 2>/dev/null | tail -1 > $OUT/bench_rgb_c3.json
python $ROOT/tools/bench_rgb.py --config C2 2>/dev/null | tail -1 > $OUT/bench_rgb_c2.json
ls -la $OUT
