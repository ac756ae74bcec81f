#!/bin/bash
# Copy a tools/refresh_profiles.sh result (gpurun_out/prof_TAG) into profiles/TAG_* and rebuild profiles/roofline_traffic.json.
# Usage: bash tools/install_refresh.sh TAG [bench_dir]   (bench_dir: a directory with bench_c3.json / bench_c3.err from a bench run
# made AFTER the traffic file was rebuilt, so that the stored line carries traffic_stale = false; default: the refresh's own)
set -e
TAG=${1:-r06}; O=gpurun_out/prof_$TAG; B=${2:-$O}
for f in $O/*.txt $O/*.csv; do
  n=$(basename $f); case $n in pmc_run.log|stats_run_*) continue;; esac
  cp $f profiles/${TAG}_$n
done
cp $B/bench_c3.json profiles/${TAG}_bench_c3_line.json
grep "^\[bench details\]" $B/bench_c3.err | sed 's/^\[bench details\] //' > profiles/${TAG}_bench_c3_details.json
python tools/make_traffic_json.py $O $TAG > /dev/null
python - <<PY
import json,sys
sys.path.insert(0,'.')
import bench
print('traffic hash matches tree:', bench.csrc_tree_hash()==json.load(open('profiles/roofline_traffic.json'))['csrc_tree_hash'])
d=json.load(open('profiles/${TAG}_bench_c3_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'stale', d['roofline']['traffic_stale'])
for k,v in json.load(open('profiles/${TAG}_bench_c3_details.json'))['sub_records'].items(): print(' ', k, v.get('value'), v.get('ms_per_step'))
PY
