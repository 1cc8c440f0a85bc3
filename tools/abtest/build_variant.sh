#!/bin/bash
# build_variant.sh <name> [-DKNOB ...]: another build of the product sources as tools/abtest/<name>.so (A/B with ab.sh)
cd "$(dirname "$0")/../../ray-marching-distance-fields_amd/csrc" || exit 1
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -fno-gpu-flush-denormals-to-zero -Wno-unused-function "$@" -shared -x hip rmdf_render.hip rmdf_env.hip rmdf_util.hip rmdf_api.cpp \
  -o ../../tools/abtest/$name.so -lz -ldl
