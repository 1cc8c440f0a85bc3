#!/bin/bash
# asan_host.sh: the library's host-only code under AddressSanitizer and UndefinedBehaviorSanitizer.  Builds the cross-check library with -fsanitize=address on the HOST
# side only (-fno-gpu-sanitize: GPU ASan needs XNACK, which this pool does not offer) into a temporary directory and runs
# tools/asan_host_workload.py against it with the ASan runtime preloaded.  No GPU needed: only host-only entry points are called.
# Output of the round-5 run: profiles/r05_asan_host.txt.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
tmp="$(mktemp -d)"; trap 'rm -rf "$tmp"' EXIT
cd "$here/../ray-marching-distance-fields_amd/csrc"
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -DRMDF_XCHECK -I. \
  -fsanitize=address,undefined -fno-sanitize=vptr -fno-gpu-sanitize -shared -x hip rmdf_api.cpp rmdf_render.hip rmdf_env.hip rmdf_util.hip xcheck/rmdf_march.hip xcheck/rmdf_stats.hip \
  -o "$tmp/librmdf_asan.so" -lz -ldl
rt="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
LD_PRELOAD="$rt" ASAN_OPTIONS=detect_leaks=0 python3 "$here/asan_host_workload.py" "$tmp/librmdf_asan.so"
