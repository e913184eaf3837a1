#!/bin/bash
# rocprofv3 kernel stats of the front-end timing tool.  usage (GPU box, repo root): bash tools/profile_filters.sh <tag>
TAG=${1:-filt}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/filter_bench.py 100 > $OUT/filter_bench.log 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep -E "path|Grid|Pass" $OUT/filter_bench.log
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[:12]:
    print(r[0][:56].ljust(56), r[1], r[3])
PY
