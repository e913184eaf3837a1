#!/bin/bash
# usage (on the GPU box): tools/ab_lib.sh NAME...  -- phase ticks + the headline bench, briefly, for each variant library
# pcl_tracking_amd/_build/var_NAME.so (tools/build_variant.py); NAME "product" = the product build
for v in "$@"; do
  lib=pcl_tracking_amd/_build/var_$v.so
  [ "$v" = product ] && lib=pcl_tracking_amd/_build/libpft_hip.so
  echo "== $v"
  PFT_LIB_PATH=$PWD/$lib python tools/phase_ticks.py 8192 50000 10 2>/dev/null | grep -v population
  PFT_LIB_PATH=$PWD/$lib python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-frontend > gpurun_out/abl_$v.json 2> gpurun_out/abl_$v.err || { echo "bench failed"; tail -5 gpurun_out/abl_$v.err; exit 1; }
  python - gpurun_out/abl_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("ms/step %.4f" % d["ms_per_step"], "running %.4f" % d.get("ms_per_step_running", 0), "lik us %.1f" % d["roofline"]["avg_launch_us"], d["kernel_ms_per_frame"])
PY
done
