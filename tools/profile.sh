#!/bin/bash
# Produces the artefacts kept under profiles/: rocprofv3 kernel-trace stats of the bench command, and the
# HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) for the dominant kernel.
# usage (GPU box, repo root): bash tools/profile.sh <tag> [bench.py arguments of another configuration]  -> gpurun_out/prof_<tag>/
TAG=${1:-r01}; shift; ARGS="$*"; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-frontend $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-frontend $ARGS > /dev/null 2> $OUT/pmc_$n.err
done
cd $ROOT
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
vals = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_likelihood<false" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
avg = {k: sum(v[len(v)//2:]) / len(v[len(v)//2:]) for k, v in vals.items()}
# MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB units; on gfx950 FETCH_SIZE reports half of the
# bytes of wide coalesced reads -> doubled (upper bound for this gather-heavy kernel: other widths are uncalibrated)
fetch = avg.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
write = avg.get("WRITE_SIZE", 0.0) * 1024.0
valu, thr = avg.get("SQ_INSTS_VALU"), avg.get("SQ_THREAD_CYCLES_VALU")
json.dump({"kernel": "k_likelihood<false>", "fetch_size_raw_kib": avg.get("FETCH_SIZE"), "write_size_raw_kib": avg.get("WRITE_SIZE"),
           "hbm_bytes_per_launch": fetch + write, "tcc_hit": avg.get("TCC_HIT_sum"), "tcc_miss": avg.get("TCC_MISS_sum"),
           # VALU issue: wave-instructions per launch, and the share of the 64 lanes that were active in them
           "valu_wave_insts_per_launch": valu, "salu_insts_per_launch": avg.get("SQ_INSTS_SALU"),
           "lds_insts_per_launch": avg.get("SQ_INSTS_LDS"), "waves_per_launch": avg.get("SQ_WAVES"),
           "valu_lane_utilisation": (thr / (64.0 * valu)) if valu and thr else None},
          open(os.path.join(out, "traffic_likelihood.json"), "w"), indent=1)
print(open(os.path.join(out, "traffic_likelihood.json")).read())
PY
head -12 $OUT/kernel_stats.csv | cut -c1-160
