#!/bin/bash
# Build a variant of librustracer_hip.so with extra -D flags for A/B runs on the GPU box:
#   bash tools/build_variant.sh <tag> -DRT_SB_WAVES=5 ...   ->  rustracer_amd/variants/lib_<tag>.so   (RUSTRACER_HIP_LIB=... python bench.py)
set -e
TAG=$1; shift
cd "$(dirname "$0")/../rustracer_amd/csrc"
mkdir -p ../variants
make -s rt_build.o host_scene.o
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-parameter"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "$@" -c rt_api.hip -o /tmp/rt_api_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/lib_$TAG.so /tmp/rt_api_$TAG.o rt_build.o host_scene.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built rustracer_amd/variants/lib_$TAG.so
