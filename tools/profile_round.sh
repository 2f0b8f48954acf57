#!/bin/bash
# One call = everything under profiles/ for one tag (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r1_c [bench.py args...]
# pass 1: rocprofv3 --kernel-trace --stats of bench.py (per-kernel averages; also prints the bench line)
# pass 2/3: --pmc FETCH_SIZE and --pmc WRITE_SIZE of the same command, on their own (the guide's HBM recipe)
# Outputs land in gpurun_out/<tag>/ (scratch); the summaries to commit are copied into gpurun_out/<tag>/profiles/.
TAG=${1:-r1_x}
shift
ARGS="--no-cpu-baseline $*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT/profiles
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
grep '^{"metric"' $OUT/stats.log > $OUT/profiles/${TAG}_bench_line.json
cp $OUT/stats/*/*kernel_stats.csv $OUT/profiles/${TAG}_kernel_stats.csv
echo "stats pass ok"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/fetch.log; exit 1; }
echo "fetch pass ok"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/write.log; exit 1; }
echo "write pass ok"
SIF=$(python3 -c "import json,sys; print(json.loads(open('$OUT/profiles/${TAG}_bench_line.json').readline())['config']['samples_in_flight'])")
SPP=$(python3 -c "import json,sys; print(json.loads(open('$OUT/profiles/${TAG}_bench_line.json').readline())['config']['spp_per_step'])")
KERNEL=$(python3 -c "import json,sys; print(json.loads(open('$OUT/profiles/${TAG}_bench_line.json').readline())['roofline']['kernel'])")
python3 $ROOT/tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/profiles/${TAG}_traffic.json --config C3 --spp-per-step $SPP --sif $SIF --kernel $KERNEL
head -4 $OUT/profiles/${TAG}_kernel_stats.csv | cut -c1-60,300-
