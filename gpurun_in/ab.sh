R=$GRAFT_REPO_ROOT
cd $R && python -m pytest tests/test_gpu_rasterizer.py tests/test_gpu_harness.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3
for L in v5 v7 v5 v7; do
  cp $R/gpurun_in/lib_$L.so $R/instascene_amd/libinstascene_hip.so
  echo "== $L: $(python $R/bench.py --no-cpu-baseline --steps 30 --view-cache-gb 8 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels_ms_per_launch"]["k_render_bwd"])')  headline $(python $R/bench.py --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
