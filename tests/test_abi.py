"""CPU tier: librmdf.so builds for gfx950, loads without a GPU, exports every symbol include/rmdf.h declares,
and fails LOUDLY (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib(rmdf):
    rmdf.build()
    return rmdf.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rmdf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rmdf_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(rmdf):
    assert declared_symbols() == sorted(rmdf.ABI_SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_code_object_is_gfx950_only(rmdf):
    out = os.popen("/opt/rocm/lib/llvm/bin/llvm-readelf --notes %s 2>/dev/null | head -0; "
                   "strings -a %s | grep -o 'amdgcn-amd-amdhsa--gfx[0-9a-z]*' | sort -u" % (rmdf.LIB_PATH, rmdf.LIB_PATH)).read().split()
    assert out == ["amdgcn-amd-amdhsa--gfx950"], out


def test_tile_index_predicates(lib, rmdf):
    # isTileIdxFirstTile / isTileIdxLastTile, ShaderRendering.hs:54-58
    for idx in (0, 1, 63, 64, 127, 128, 640):
        assert bool(lib.rmdf_is_tile_idx_first_tile(idx)) == (idx % 64 == 0) == rmdf.is_tile_idx_first_tile(idx)
        assert bool(lib.rmdf_is_tile_idx_last_tile(idx)) == (idx % 64 == 63) == rmdf.is_tile_idx_last_tile(idx)


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
def test_create_fails_loudly_without_a_device(lib, rmdf):
    ctx = C.c_void_p()
    rc = lib.rmdf_create(C.byref(ctx), None)
    assert rc == -2 and not ctx.value                       # RMDF_E_NO_DEVICE, nothing allocated
    assert b"HIP device" in lib.rmdf_last_error(None)
    with pytest.raises(rmdf.RmdfError) as e:
        rmdf.ShaderRenderer(0)
    assert e.value.code == -2
    with pytest.raises(rmdf.RmdfError):
        with rmdf.with_shader_renderer():
            pass


def test_null_ctx_is_an_error_not_a_crash(lib):
    assert lib.rmdf_render_tile(None, 2, -1, 16, 16, 0.0, 16, None) == -1
    assert lib.rmdf_set_env_cube(None, 0, None, 4) == -1
    assert lib.rmdf_synchronize(None, None) == -1
    lib.rmdf_destroy(None)


def test_product_never_touches_the_oracle(rmdf):
    """The shipped package and library must not import, link or call anything under oracle/."""
    pkg = os.path.dirname(rmdf.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".hs")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "rmdf_oracle" not in text and "from oracle" not in text, f
    deps = os.popen("ldd %s" % rmdf.LIB_PATH).read()
    assert "oracle" not in deps and "amdhip64" in deps
