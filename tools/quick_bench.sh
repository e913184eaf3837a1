#!/bin/bash
# usage: bash tools/quick_bench.sh <label> [bench args...]
L=$1; shift
python bench.py --no-cpu-baseline "$@" > gpurun_out/qb_$L.json 2> gpurun_out/qb_$L.err
python - "$L" <<'PY'
import json, sys
d = json.load(open("gpurun_out/qb_%s.json" % sys.argv[1]))
print("%-14s ms/frame %.3f  fps %.1f  lik %.1f us  frac %.3f  crop %d depth %d kbar %.2f  per-frame ms %s" % (
    sys.argv[1], d["ms_per_step"], d["frames_per_s"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"],
    d["cropped_points"], d["octree_depth"], d["mean_leaf_occupancy"], {k: round(v, 3) for k, v in d["kernel_ms_per_frame"].items()}))
PY
