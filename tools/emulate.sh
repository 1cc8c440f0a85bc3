#!/bin/bash
# emulate.sh <command ...> -- run a program that uses librmdf (a test selection, examples/c_host, tools/render_png.py ...) on the EMULATED device: the HIP
# test double (tests/fake_hip.cpp) with FAKE_HIP_EMULATE=1 hands every kernel launch to the SIMT emulator's builds of the kernel source
# (tests/kernel_on_host.cpp).  No GPU needed; a 1080p frame takes ~6 s on eight cores; pixels are bit-exact.  The light probe is a private copy
# with the oracle's cache files beside it (the emulated 256 x 128 prefilter would take an hour).
#   tools/emulate.sh python -m pytest tests -m gpu -q -k "small_frames or golden"
#   tools/emulate.sh python tools/render_png.py /tmp/out
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import os, subprocess, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import rmdf_amd, test_kernel_source_on_host as t
from test_host_logic import _fake_hip_lib
rmdf_amd.build()
_fake_hip_lib()
t.Emulated.build([((), ""), (("-DRMDF_XCHECK",), "_xcheck")])
PY
d=$(mktemp -d /tmp/rmdf_emulated_probe_XXXX)
cp ray-marching-distance-fields_amd/data/latlong_envmaps/uffizi_512.hdr tests/golden/env_cache/*.hdr "$d"/
export RMDF_ENV_HDR=$d/uffizi_512.hdr FAKE_HIP_EMULATE=1 LD_PRELOAD=$PWD/tests/libfake_hip.so${LD_PRELOAD:+:$LD_PRELOAD}
"$@"; rc=$?
rm -rf "$d"
exit $rc
