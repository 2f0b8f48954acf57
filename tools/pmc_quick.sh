#!/bin/bash
# Quick counter passes of bench.py for one parameter set (diagnostics; run on the GPU box from the repo root):
#   bash tools/pmc_quick.sh <tag> "<PRT_PARAMS>" [bench args]
TAG=$1; PARAMS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PRT_PARAMS="$PARAMS"
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary $*"
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py $ARGS > $OUT/pass$i.log 2>&1
  echo "pass $i ($grp): rc=$?"
done
python3 $ROOT/tools/pmc_report.py $OUT | grep -A14 "k_traverse8_persistent<[0-9]*, [0-9]*, false, false, true, false"
