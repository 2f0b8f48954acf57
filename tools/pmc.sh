#!/bin/bash
# Collects PMC counters for the C3 workload in separate passes (one counter group per pass, as the guide asks).
# usage (on the GPU box, from the repo root): bash tools/pmc.sh <outdir-under-gpurun_out> ["sweep set"]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
SET=${2:-"sif=16"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE MeanOccupancyPerCU"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/tools/sweep.py --config C3 --sets "$SET" --rounds 1 --spp 16 > $OUT/pass$i.log 2>&1
  echo "pass $i ($grp): rc=$?"
done
