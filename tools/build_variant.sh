#!/bin/bash
# Build an alternative libprt.so from the same sources with extra compiler flags / -D switches for A/B timing:
#   tools/build_variant.sh <tag> "<extra hipcc flags>"   ->  parallelraytracing_amd/csrc/ab/libprt_<tag>.so
# (run here, in the build container: the .so travels to the GPU box with the snapshot; tools/ab_libs.py times them)
set -e
cd "$(dirname "$0")/../parallelraytracing_amd/csrc"
tag=$1; shift
mkdir -p ab/$tag
make -s libprt.so >/dev/null
FL="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 $*"
/opt/rocm/bin/hipcc $FL -c prt_kernels.hip -o ab/$tag/prt_kernels.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ab/libprt_$tag.so ab/$tag/prt_kernels.o bvh_gpu.o prt_api.o prt_group.o prt_host.o bvh.o -pthread -ldl
echo "built ab/libprt_$tag.so ($*)"
