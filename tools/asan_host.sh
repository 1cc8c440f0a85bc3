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
# ... and everything behind a ctx -- staging, whole-frame bands (all four hand-overs), tile jobs, shards, the env pipeline and its cache
# files, injected allocation failures, the N-rank exchange against the RCCL double -- with the HIP runtime replaced by tests/fake_hip.cpp
# ("device memory" = malloc'd memory: every copy in or out of it is checked by the sanitizer's red zones)
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --cuda-host-only --offload-arch=gfx950 -x hip -I . -fsanitize=address,undefined -fno-sanitize=vptr \
  -shared "$here/../tests/fake_hip.cpp" -o "$tmp/libfake_hip.so"
gcc -O1 -g -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include "$here/../tests/fake_rccl.c" -o "$tmp/libfake_rccl.so" -L/opt/rocm/lib -lamdhip64
LD_PRELOAD="$rt $tmp/libfake_hip.so" ASAN_OPTIONS=detect_leaks=0 FAKE_HIP_WORKLOAD_XCHECK_LIB="$tmp/librmdf_asan.so" FAKE_HIP_LIB="$tmp/libfake_hip.so" \
  RMDF_RCCL_LIB="$tmp/libfake_rccl.so" FAKE_RCCL_TIMEOUT_S=120 python3 "$here/../tests/fake_hip_workload.py" xcheck
