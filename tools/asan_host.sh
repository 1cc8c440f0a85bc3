#!/bin/bash
# asan_host.sh: the library's host code under AddressSanitizer + UndefinedBehaviorSanitizer, then under ThreadSanitizer.  Builds the cross-check library with -fsanitize=address on the HOST
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
for async in 0 1; do
  echo "# AddressSanitizer + UBSan, HIP double with FAKE_HIP_ASYNC=$async"
  LD_PRELOAD="$rt $tmp/libfake_hip.so" ASAN_OPTIONS=detect_leaks=0 FAKE_HIP_WORKLOAD_XCHECK_LIB="$tmp/librmdf_asan.so" FAKE_HIP_LIB="$tmp/libfake_hip.so" \
    FAKE_HIP_ASYNC=$async FAKE_HIP_JITTER_US=30 FAKE_HIP_WORKLOAD_WATCHDOG_S=1500 \
    RMDF_RCCL_LIB="$tmp/libfake_rccl.so" FAKE_RCCL_TIMEOUT_S=120 python3 "$here/../tests/fake_hip_workload.py" xcheck
done
# the PRODUCT flavour of the sources (no -DRMDF_XCHECK: what librmdf.so is built from) under the same sanitizers, asynchronous double
echo "# AddressSanitizer + UBSan, product flavour (no RMDF_XCHECK), HIP double with FAKE_HIP_ASYNC=1"
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -I. \
  -fsanitize=address,undefined -fno-sanitize=vptr -fno-gpu-sanitize -shared -x hip rmdf_api.cpp rmdf_render.hip rmdf_env.hip rmdf_util.hip \
  -o "$tmp/librmdf_product_asan.so" -lz -ldl
LD_PRELOAD="$rt $tmp/libfake_hip.so" ASAN_OPTIONS=detect_leaks=0 RMDF_LIB="$tmp/librmdf_product_asan.so" FAKE_HIP_LIB="$tmp/libfake_hip.so" \
  FAKE_HIP_ASYNC=1 FAKE_HIP_JITTER_US=30 FAKE_HIP_WORKLOAD_WATCHDOG_S=1500 python3 "$here/../tests/fake_hip_workload.py"
# ThreadSanitizer: the same host code and the double in its asynchronous mode (streams are threads: a buffer touched by the host and by a
# queued operation without an event, a stream synchronisation or a flag between them is a reported race)
echo "# ThreadSanitizer, HIP double with FAKE_HIP_ASYNC=1"
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-option-ignored -DRMDF_XCHECK -I. \
  -fsanitize=thread -fno-gpu-sanitize -shared -x hip rmdf_api.cpp rmdf_render.hip rmdf_env.hip rmdf_util.hip xcheck/rmdf_march.hip xcheck/rmdf_stats.hip \
  -o "$tmp/librmdf_tsan.so" -lz -ldl
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --cuda-host-only --offload-arch=gfx950 -x hip -I . -fsanitize=thread -shared "$here/../tests/fake_hip.cpp" -o "$tmp/libfake_hip_tsan.so"
trt="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)"
LD_PRELOAD="$trt $tmp/libfake_hip_tsan.so" TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66" FAKE_HIP_ASYNC=1 FAKE_HIP_WORKLOAD_WATCHDOG_S=3000 \
  FAKE_HIP_WORKLOAD_XCHECK_LIB="$tmp/librmdf_tsan.so" FAKE_HIP_LIB="$tmp/libfake_hip_tsan.so" RMDF_RCCL_LIB="$tmp/libfake_rccl.so" FAKE_RCCL_TIMEOUT_S=300 \
  python3 "$here/../tests/fake_hip_workload.py" xcheck 2> "$tmp/tsan.err" || { tail -50 "$tmp/tsan.err"; exit 1; }
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' "$tmp/tsan.err" || true)"
