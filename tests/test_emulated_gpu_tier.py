"""CPU tier (round 6): the GPU tier's own parity tests, through the C ABI, on an EMULATED device.

With FAKE_HIP_EMULATE=1 the HIP test double (tests/fake_hip.cpp) no longer writes stand-in pixels: every hipLaunchKernel of librmdf.so is looked up
by its mangled name in tests/libkernel_on_host.so -- the library's kernel SOURCE compiled for the CPU and executed by the SIMT emulator
(tests/kernel_on_host.cpp, tests/koh_shim/) -- and RUN.  The process then is: the product's shared library as shipped (host code: contexts,
staging, tile jobs, env pipeline, cache files), its kernels' source executed lane by lane, and the `-m gpu` tests as they are written, calling
through the C ABI and comparing with the oracle.  88 of the tier's tests fit the emulator's speed (frames up to 480 x 270); the others
need full-size frames, torch device buffers or RCCL and stay the GPU's.  The selection below runs in a child process, four workers wide."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_host_logic import _fake_hip_lib

SELECTION = ("test_small_frames_vs_oracle or test_vs_committed_golden or test_fixture_grid_256x144 or test_hip_planes_vs_reference_shader_fixtures "
             "or test_medium_and_ragged_frames or test_degenerate_frame_sizes or test_tiles_of_sizes_8_does_not_divide or test_tiled_frame_equals_full_frame "
             "or test_tile_jobs_issued_ahead or test_random_views_vs_oracle or test_max_steps_edge_cases or test_fresh_frame_is_cleared_to_opaque_black "
             "or test_error_convention or test_env_upload_matches_oracle_padding or test_determinism or test_argument_limits "
             "or test_resize_of_a_map_that_is_not_2_to_1 or test_malformed_hdr_files_fail_cleanly or test_latlong_to_cube_is_bit_exact "
             "or (test_lobe_prefilter_is_bit_exact and (32-16 or 8-3 or 4-2))")


def _emulator_builds(rmdf):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_kernel_source_on_host as t
    if not os.path.exists(t.CLANG):
        pytest.skip("no clang++")
    t.Emulated.build([((), ""), (("-DRMDF_XCHECK",), "_xcheck")])


def test_smoke_of_the_driver_entry_point_on_the_emulated_device(rmdf, tmp_path):
    """__graft_entry__.smoke() -- what the driver runs on the MI355X before the bench -- with the emulated device: the product's env pipeline
    (cache files found next to a private copy of the probe, lat/long -> cube and RGB16F upload on the emulator), one Mandelbulb and one Cornell
    frame through the C ABI; its own assertions: cube maps bit-equal, steps / iterations bit-exact, colour within 1e-4 (here: 0)."""
    import shutil
    from conftest import ENV_CACHE
    _emulator_builds(rmdf)
    probe = str(tmp_path / os.path.basename(rmdf.DEFAULT_ENV_HDR))
    shutil.copy(rmdf.DEFAULT_ENV_HDR, probe)
    for f in os.listdir(ENV_CACHE):
        shutil.copy(os.path.join(ENV_CACHE, f), str(tmp_path))
    code = ("import sys; sys.path.insert(0, %r); import rmdf_amd; rmdf_amd.DEFAULT_ENV_HDR = %r; import __graft_entry__ as g; g.smoke(); print('smoke ok')" % (ROOT, probe))
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), FAKE_HIP_EMULATE="1")
    env.pop("RMDF_LIB", None)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "smoke ok" in r.stdout and "stand-in used" not in r.stderr, (r.stdout[-1500:], r.stderr[-3000:])
    assert "max rel colour err 0," in r.stdout, r.stdout


def test_the_gpu_tiers_parity_tests_pass_on_the_emulated_device(rmdf):
    _emulator_builds(rmdf)
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), FAKE_HIP_EMULATE="1", FAKE_HIP_EMULATE_THREADS="2")
    for k in ("RMDF_LIB", "RMDF_TEST_UNVERIFIED"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_env.py"),
                        "-q", "-m", "gpu", "-p", "no:cacheprovider", "-n", "4", "--timeout=300", "-k", SELECTION],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    tail = [l for l in r.stdout.strip().splitlines() if " passed" in l or " failed" in l or " error" in l]
    assert r.returncode == 0 and tail and "failed" not in tail[-1] and "error" not in tail[-1], (r.stdout[-4000:], r.stderr[-1500:])
    n = int(tail[-1].split(" passed")[0].split()[-1])
    assert n >= 86, tail[-1]
    assert "stand-in used" not in r.stderr and "stand-in used" not in r.stdout, "a launch fell back to a stand-in: the pixels compared were not the kernels'"
