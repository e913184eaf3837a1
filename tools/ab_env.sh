#!/bin/bash
# usage (on the GPU box): tools/ab_env.sh "VAR=value ..." ["VAR=value ..." ...]  -- the headline bench, briefly, once per environment
# (first argument "" = the defaults); prints ms per frame and the per-kernel split of each
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-frontend > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err || { echo "run $i failed"; tail -5 gpurun_out/ab_$i.err; exit 1; }
  python - "$e" gpurun_out/ab_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("[%s]" % sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "running %.4f" % d.get("ms_per_step_running", 0), "lik us %.1f" % d["roofline"]["avg_launch_us"], d["kernel_ms_per_frame"])
PY
done
