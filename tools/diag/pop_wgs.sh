#!/bin/bash
# population kernel time against the number of workgroups (particles per thread), at 8 192 and 65 536 particles
for P in 8192 65536; do
  for W in 256 128 64 32 16 8; do
    PFT_POP_MAX_WGS=$W python bench.py --particles-per-gpu $P --steps 30 --warmup 8 --no-cpu-baseline --no-frontend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('P=$P max_wgs=$W ms/step %.4f population %.4f' % (d['ms_per_step'], d['kernel_ms_per_frame']['population']))"
  done
done
