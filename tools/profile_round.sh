#!/bin/bash
# One call = everything under profiles/ for one tag (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r2_a [bench.py args, e.g. --config C5 --steps 2 ...]
# pass stats : rocprofv3 --kernel-trace --stats of bench.py (per-kernel averages; also prints the bench line)
# pass fetch / write : --pmc FETCH_SIZE / --pmc WRITE_SIZE on their own (the guide's HBM recipe)
# pass sq / sq2 / tcc / occ : SQ cycle + instruction counters (8 SQ slots per pass), L2 hit rate, occupancy + GRBM clock
# Every pass is `rocprofv3 ... -- python3 bench.py` (the program directly after --), counters never mixed with sys/hip traces.
# Outputs land in gpurun_out/<tag>/ (scratch); the summaries to commit are copied into gpurun_out/<tag>/profiles/.
TAG=${1:-r2_x}
shift
ARGS="--no-cpu-baseline --no-secondary $*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
READ_FACTOR=${READ_FACTOR:-2.0}
mkdir -p $OUT/profiles
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
grep '^{"metric"' $OUT/stats.log > $OUT/profiles/${TAG}_bench_line.json
cp $OUT/stats/*/*kernel_stats.csv $OUT/profiles/${TAG}_kernel_stats.csv
echo "stats pass ok"
run_pass() {  # name, counters...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pass_$name -- python3 $ROOT/bench.py $ARGS > $OUT/pass_$name.log 2>&1 || { echo "$name pass failed"; tail -5 $OUT/pass_$name.log; return 1; }
  echo "$name pass ok"
}
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
run_pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES || exit 1
run_pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT || echo "(sq2 optional)"
run_pass tcc TCC_HIT_sum TCC_MISS_sum || echo "(tcc optional)"
run_pass occ GRBM_GUI_ACTIVE MeanOccupancyPerCU || echo "(occ optional)"
J() { python3 -c "import json,sys; d=json.loads(open('$OUT/profiles/${TAG}_bench_line.json').readline()); print($1)"; }
SIF=$(J "d['config']['samples_in_flight']"); SPP=$(J "d['config']['spp_per_step']"); KERNEL=$(J "d['roofline']['kernel']")
CFG=$(J "d['config']['workload'].split(':')[0]"); JIT=$(J "d['config'].get('jitter', 0)")
python3 $ROOT/tools/pmc_traffic.py $OUT/pass_fetch $OUT/pass_write $OUT/profiles/${TAG}_traffic.json --config $CFG --spp-per-step $SPP --sif $SIF --kernel $KERNEL --jitter $JIT --read-factor $READ_FACTOR
python3 $ROOT/tools/pmc_sq.py $OUT $OUT/profiles/${TAG}_sq.json --config $CFG --sif $SIF --kernel $KERNEL --jitter $JIT
# the fingerprint of the kernel sources these counters belong to, and the bench line again now that its own counter files exist
python3 $ROOT/tools/kernel_sha.py $OUT/profiles/${TAG}_traffic.json $OUT/profiles/${TAG}_sq.json > /dev/null
mkdir -p $ROOT/profiles && cp $OUT/profiles/${TAG}_traffic.json $OUT/profiles/${TAG}_sq.json $ROOT/profiles/
# (the loops' static instruction counts of exactly this kernel code as well: the line takes its per-trip VALU counts from them)
ROUND=${TAG%%_*}
python3 $ROOT/tools/isa_count.py --out $ROOT/profiles/${ROUND}_isa_counts.json > /dev/null 2>&1 && cp $ROOT/profiles/${ROUND}_isa_counts.json $OUT/profiles/
python3 $ROOT/bench.py $ARGS 2> /dev/null | grep '^{"metric"' > $OUT/profiles/${TAG}_bench_line.json || echo "(final bench line failed; the stats pass's line is kept)"
head -4 $OUT/profiles/${TAG}_kernel_stats.csv | cut -c1-60,300-
