#!/bin/bash
# A/B of compiler scheduling flags for the device code: one libprt.so per flag set, alternated process by process.
#   bash tools/flags_ab.sh build      (here, no GPU: compiles prt_kernels.hip once per flag set into csrc/alt/, ~1 min each, in parallel)
#   gpurun -- 'bash tools/flags_ab.sh'   (on the GPU box: tools/sweep.py on C3 at the headline batch size with each library, twice)
# Round 3 (gpurun_out/flags1.log, TUNING.md): every set within +-1 % of the default build.
cd "$(dirname "$0")/.." || exit 1
CS=parallelraytracing_amd/csrc
declare -A FL=( [base]="" [bias0]="-mllvm -amdgpu-schedule-metric-bias=0" [nopost]="-mllvm -enable-post-misched=false"
                [relaxed]="-mllvm -amdgpu-schedule-relaxed-occupancy" [trackers]="-mllvm -amdgpu-use-amdgpu-trackers"
                [maxilp]="-mllvm -amdgpu-sched-strategy=max-ilp" [maxclause]="-mllvm -amdgpu-sched-strategy=max-memory-clause" )
if [ "$1" = build ]; then
  make -C $CS -j8 > /dev/null || exit 1
  mkdir -p $CS/alt
  for t in "${!FL[@]}"; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 ${FL[$t]} \
        -c $CS/prt_kernels.hip -o $CS/alt/k_$t.o -Rpass-analysis=kernel-resource-usage > $CS/alt/$t.report 2>&1 &&
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $CS/alt/libprt_$t.so $CS/alt/k_$t.o $CS/bvh_gpu.o $CS/prt_api.o \
        $CS/prt_group.o $CS/prt_host.o $CS/bvh.o -pthread -ldl && echo "built $t" ) &
  done
  wait
  exit 0
fi
out=gpurun_out/flags_ab.log
mkdir -p gpurun_out
: > $out
for rep in 1 2; do
  for t in base bias0 nopost relaxed trackers maxilp maxclause; do
    echo "== $t rep $rep" >> $out
    PRT_LIB_PATH=$PWD/$CS/alt/libprt_$t.so python tools/sweep.py --config C3 --sets "variant=0" --spp 256 --sif 256 --rounds 5 2>&1 |
      grep "median\|!!\|rror" >> $out || exit 1
  done
done
cat $out
