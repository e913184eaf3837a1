#!/bin/bash
# stage shares of k_likelihood by ablation (timing only; results are wrong by construction).  The switch exists only in
# the diagnostic variant library: python tools/build_variant.py diag -DPFT_DIAG   (before gpurun; hipcc cross-compiles)
export PFT_LIB_PATH=$PWD/pcl_tracking_amd/_build/var_diag.so
[ -f "$PFT_LIB_PATH" ] || { echo "build the diagnostic variant first"; exit 1; }
for a in 0 1 2 4 3 7; do
  echo -n "PFT_ABLATE=$a  "
  PFT_ABLATE=$a python bench.py --steps 30 --warmup 10 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lik us', round(d['roofline']['avg_launch_us'],1), 'frame ms', round(d['ms_per_step'],3))"
done
