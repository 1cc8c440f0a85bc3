import os
import sys

import numpy as np
import pytest

try:
    # PyTorch bundles its own copy of the HIP runtime (same soname as the system's): it must be loaded BEFORE librmdf.so
    # pulls in the system one, or the process ends up with two runtimes and torch finds no GPU afterwards.  The tests use
    # torch only for device buffers and streams; the product never imports it.
    import torch  # noqa: F401
except Exception:                                       # noqa: BLE001
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
ENV_CACHE = os.path.join(GOLD, "env_cache")        # the ORACLE's pre-convolved maps of uffizi_512 (make_fixtures.py --caches)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Round 5: GPU access was closed from outside the build while part of the round's work was still unmeasured.  Tests of code that has
# never run on hardware are kept, but only run when asked for (RMDF_TEST_UNVERIFIED=1): a test that has never been seen green must not
# stand in the tier the driver runs -- it would stop the tier (-x) for a reason nobody has looked at.  DESIGN.md section 5 lists them.
ON_HIP_DOUBLE = "libfake_hip" in os.environ.get("LD_PRELOAD", "")       # the process runs against the HIP test double (no GPU)
# ... and with FAKE_HIP_EMULATE=1 the double RUNS the library's kernels (tests/kernel_on_host.cpp: their source under a SIMT emulator): the pixels
# are the real ones, so the GPU tier's tests hold as written -- as far as their frame sizes allow (the emulator is ~10^4 x slower than the GPU)
ON_HIP_EMULATOR = ON_HIP_DOUBLE and os.environ.get("FAKE_HIP_EMULATE", "0") not in ("", "0")
if ON_HIP_EMULATOR:
    # tests that hold torch device buffers and streams: the same stand-ins bench.py's dry run uses ("device" tensors are CPU tensors: the
    # double's device memory IS host memory, and the emulated kernels write where the pointer says)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_cuda_standins
    torch_cuda_standins.install()
unverified = pytest.mark.skipif(os.environ.get("RMDF_TEST_UNVERIFIED") != "1",
                                reason="code written after GPU access closed in round 5: never run on hardware (RMDF_TEST_UNVERIFIED=1 runs it)")


@pytest.fixture(scope="session", autouse=True)
def _no_stand_in_cache_files_in_the_tree():
    """Against the HIP double the env kernels are stand-ins: a test that loads the shipped light probe would leave stand-in cache files
    next to it, where the real library would find them later.  There the probe's path points at a private copy."""
    if not ON_HIP_DOUBLE:
        yield
        return
    import shutil
    import tempfile
    import rmdf_amd
    d = tempfile.mkdtemp(prefix="rmdf_double_probe_")
    real = rmdf_amd.DEFAULT_ENV_HDR
    rmdf_amd.DEFAULT_ENV_HDR = os.path.join(d, os.path.basename(real))
    shutil.copy(real, rmdf_amd.DEFAULT_ENV_HDR)
    if ON_HIP_EMULATOR:
        # the emulated prefilter of the 256 x 128 probe would take an hour: the private copy comes with the ORACLE's cache files (the
        # product's are byte-identical: tests/test_gpu_env.py), so rmdf_load_env_hdr finds its caches as it does on every run but the first
        for f in os.listdir(ENV_CACHE):
            shutil.copy(os.path.join(ENV_CACHE, f), d)
    os.environ["RMDF_ENV_HDR"] = rmdf_amd.DEFAULT_ENV_HDR        # child processes (tools/tile_mode_fuzz.py, the C hosts) as well
    yield
    os.environ.pop("RMDF_ENV_HDR", None)
    rmdf_amd.DEFAULT_ENV_HDR = real
    shutil.rmtree(d, ignore_errors=True)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def rmdf():
    import rmdf_amd
    return rmdf_amd




@pytest.fixture(scope="session")
def env_latlongs(orc, rmdf):
    """uffizi_512 + the oracle's pre-convolved caches (committed fixtures, tests/golden/env_cache), decoded by the oracle."""
    rd = lambda fn: orc.hdr_decode(open(fn, "rb").read())
    return {"refl": rd(rmdf.DEFAULT_ENV_HDR),
            "cos1": rd(os.path.join(ENV_CACHE, "uffizi_512_cache_pow_1.0.hdr")),
            "cos8": rd(os.path.join(ENV_CACHE, "uffizi_512_cache_pow_8.0.hdr")),
            "cos64": rd(os.path.join(ENV_CACHE, "uffizi_512_cache_pow_64.0.hdr")),
            "cos512": rd(os.path.join(ENV_CACHE, "uffizi_512_cache_pow_512.0.hdr"))}


@pytest.fixture(scope="session")
def env_faces(orc, env_latlongs):
    """Oracle-built float32 cube faces (the input both sides share in strict parity tests)."""
    return {k: orc.latlong_to_cube(env_latlongs[k]) for k in ("refl", "cos1", "cos8")}


@pytest.fixture(scope="session")
def env_oracle(orc, env_faces):
    return orc.EnvSet(*(orc.cube_pad_f16(env_faces[k]) for k in ("refl", "cos1", "cos8")))


def _renderer_with_product_env(rmdf, env_oracle, **kw):
    """A ShaderRenderer whose cube maps come from the PRODUCT's own env pipeline (rmdf_load_env_hdr on the shipped uffizi_512.hdr:
    GPU resize, GPU lobe prefilter, RGBE cache files, GPU cube conversion) -- checked here, once per renderer, to be bit-equal
    to the oracle-built maps every parity test compares against."""
    rmdf.build()
    r = rmdf.ShaderRenderer(0, **kw)
    if ON_HIP_DOUBLE and not ON_HIP_EMULATOR:
        # dry run of GPU-tier tests against tests/fake_hip.cpp (tests/test_host_logic.py: test_gpu_tier_tests_that_need_no_oracle_...): the
        # double's env kernels are stand-ins -- no cache files into the tree, no comparison with the oracle's maps
        for slot, ref in ((rmdf.ENV_REFLECTION, env_oracle.reflection), (rmdf.ENV_COS_1, env_oracle.cos_1), (rmdf.ENV_COS_8, env_oracle.cos_8)):
            r.set_env_cube(slot, np.ascontiguousarray(ref.view(np.float16)[:, 1:-1, 1:-1, :3].astype(np.float32)))
        return r
    r.load_env_hdr(rmdf.DEFAULT_ENV_HDR)
    for slot, ref in ((rmdf.ENV_REFLECTION, env_oracle.reflection), (rmdf.ENV_COS_1, env_oracle.cos_1), (rmdf.ENV_COS_8, env_oracle.cos_8)):
        assert np.array_equal(r.get_env_cube_padded(slot), ref), "product-built cube map %d differs from the oracle's" % slot
    return r


@pytest.fixture(scope="session")
def sr(rmdf, env_oracle):
    """The product renderer on cuda:0, environment built by its own pipeline."""
    r = _renderer_with_product_env(rmdf, env_oracle)
    yield r
    r.close()


@pytest.fixture(scope="session")
def sr_alt(rmdf, env_oracle):
    """librmdf_xcheck.so: the Mandelbulb runs on the flattened march + shade kernels."""
    r = _renderer_with_product_env(rmdf, env_oracle, flags=rmdf.FLAG_FLAT_MARCH)
    yield r
    r.close()


def rel_err(a, b, floor=1e-6):
    """max relative error with an absolute floor; NaN matches NaN, inf matches inf."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & (a == b)
    with np.errstate(invalid="ignore"):
        e = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), floor)
    e = np.where(both_nan | same_inf, 0.0, e)
    e = np.where(np.isnan(e), np.inf, e)
    return e
