#!/bin/bash
# workgroup-shape sweep of k_likelihood (rebuilds the library on the GPU box for each shape)
for cfg in "1024 1" "1024 2" "768 2" "512 2" "512 3" "512 4" "256 4" "256 6"; do
  set -- $cfg
  PFT_EXTRA_HIPCC_FLAGS="-DPFT_LIK_THREADS=$1 -DPFT_LIK_WGS_PER_CU=$2" python -m pcl_tracking_amd.build --force > /dev/null 2>&1
  echo -n "threads=$1 wgs/cu=$2: "
  python tools/lik_microbench.py 2>&1 | head -1
done
python -m pcl_tracking_amd.build --force > /dev/null 2>&1
