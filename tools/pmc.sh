#!/bin/bash
# PMC passes over tools/run_frames.py; one rocprofv3 run per counter group (SQ: 8 slots, TCC: 4).
# usage (on the GPU box, from the repo root): bash tools/pmc.sh <outdir> [frames]
OUT=${1:-gpurun_out/pmc}; FR=${2:-6}
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
 "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_ANY" \
 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/$OUT/g$i -- python3 $ROOT/tools/run_frames.py $FR > $ROOT/$OUT/g$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT
