#!/bin/bash
# usage (on the GPU box): tools/quick_bench.sh <tag> [extra hipcc flags]  -- rebuilds with the flags, runs the headline bench briefly
tag=$1; shift
PFT_EXTRA_HIPCC_FLAGS="$*" python -m pcl_tracking_amd.build --force > gpurun_out/build_$tag.log 2>&1 || { echo build failed; tail -5 gpurun_out/build_$tag.log; exit 1; }
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-frontend > gpurun_out/qb_$tag.json 2> gpurun_out/qb_$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/qb_$tag.json"))
print("$tag", "ms/step %.4f" % d["ms_per_step"], "lik us %.1f" % d["roofline"]["avg_launch_us"], d["kernel_ms_per_frame"])
PY
