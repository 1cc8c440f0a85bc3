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


def declared_xcheck_symbols():
    text = open(os.path.join(ROOT, "include", "rmdf_xcheck.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rmdf_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(rmdf):
    assert declared_symbols() == sorted(rmdf.ABI_SYMBOLS)
    assert declared_xcheck_symbols() == sorted(rmdf.XCHECK_SYMBOLS)


def test_every_declared_symbol_is_exported(lib, rmdf):
    for name in declared_symbols():
        assert hasattr(lib, name), name
    xlib = rmdf.load_library(xcheck=True)
    for name in declared_symbols() + declared_xcheck_symbols():
        assert hasattr(xlib, name), name


def test_product_library_holds_no_alternative_schedules_or_debug_knobs(rmdf):
    """librmdf.so is the product: the slower alternative schedules, the per-wave statistics and every measurement knob
    live in librmdf_xcheck.so (or nowhere)."""
    syms = os.popen("nm -D --defined-only %s; strings -a %s" % (rmdf.LIB_PATH, rmdf.LIB_PATH)).read()
    for word in ("rmdf_debug_march_stats", "k_march_mb8", "k_march_pool", "k_march_refill", "k_march_stats",
                 "RMDF_DBG_SKIP", "RMDF_WPB", "RMDF_OCC_LDS", "RMDF_MERGE", "RMDF_PRIO_STRIPS", "RMDF_NESTED_STATS"):
        assert word not in syms, word
    xsyms = os.popen("strings -a %s" % rmdf.XCHECK_LIB_PATH).read()
    assert "k_march_mb8" in xsyms and "k_march_stats" in xsyms


def test_product_library_reads_no_environment_knob(rmdf):
    """no getenv knob on any launch path of the product.  Every RMDF_* variable (prefilter forms, streaming copies, RCCL
    override, guard allocator, statistics) is compiled into librmdf_xcheck.so only; the one variable the product names is the HIP runtime's own
    GPU_MAX_HW_QUEUES, which rmdf_create sets when the host has not."""
    names = sorted(set(l for l in os.popen("strings -a %s" % rmdf.LIB_PATH).read().split("\n") if l.startswith("RMDF_")))
    assert names == [], names
    xnames = set(l for l in os.popen("strings -a %s" % rmdf.XCHECK_LIB_PATH).read().split("\n") if l.startswith("RMDF_"))
    assert {"RMDF_PREFILTER_RING", "RMDF_COPY_NT", "RMDF_RCCL_LIB"} <= xnames, xnames


def test_every_product_kernel_is_one_the_gpu_tier_has_run(rmdf):
    """tests/golden/gpu_tested_kernels.json: the instruction stream of every kernel of the last library build a GPU executed (commit and run
    record inside).  The shipped librmdf.so must consist of those kernels, instruction for instruction -- code that has never met hardware
    (round 5: the band epilogue, the band-aware strip order, the ring prefilter) belongs in librmdf_xcheck.so until it has -- or name the
    exceptions in the manifest's `not_yet_run` with the reason, so that the list of unverified product code is never implicit."""
    import json
    import sys
    from conftest import GOLD, ROOT
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang-offload-bundler"):
        pytest.skip("no clang-offload-bundler")
    sys.path.insert(0, os.path.join(ROOT, "tools", "isa"))
    import kernel_diff
    man = json.load(open(os.path.join(GOLD, "gpu_tested_kernels.json")))
    have = kernel_diff.kernel_hashes(rmdf.LIB_PATH)
    assert len([k for k in have if "k_render<" in k]) == 21
    new = {k: v for k, v in have.items() if man["kernels"].get(k) != v and k not in man["not_yet_run"]}
    assert not new, "product kernels no GPU has run (move them to librmdf_xcheck.so or list them under not_yet_run): %s" % sorted(new)
    stale = [k for k in man["not_yet_run"] if man["kernels"].get(k) == have.get(k)]
    assert not stale, "listed as not yet run but identical to the GPU-tested build: %s" % stale


def test_rccl_is_not_a_link_dependency(rmdf):
    """RCCL is dlopen()ed by rmdf_comm_init: a single-GPU C host must not need it at load time."""
    deps = os.popen("ldd %s" % rmdf.LIB_PATH).read()
    assert "rccl" not in deps and "nccl" not in deps


def test_last_error_without_ctx_is_process_wide(lib):
    """rmdf_last_error(NULL) must work from another OS thread than the one the call failed on (a Haskell runtime may
    migrate its threads): the message is process-wide, the returned pointer a per-thread copy."""
    import threading
    assert lib.rmdf_save_png(None, None, 0, 0) == -1
    here = lib.rmdf_last_error(None)
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.rmdf_last_error(None)))
    t.start()
    t.join()
    assert here == seen[0] and b"rmdf_save_png" in here
    err = C.create_string_buffer(256)
    ctx = C.c_void_p()

    class Cfg(C.Structure):
        _fields_ = [("device", C.c_int), ("reserved", C.c_int * 7)]
    cfg = Cfg(device=-5)
    rc = lib.rmdf_create_ex(C.byref(ctx), C.byref(cfg), err, 256)
    assert rc != 0 and not ctx.value and len(err.value) > 0


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
    assert lib.rmdf_comm_init(None, None, 0, 1) == -1 and lib.rmdf_gather_shards_device(None, 8, 8, None, None, None) == -1
    assert lib.rmdf_comm_get_unique_id(None) == -1
    assert lib.rmdf_prefilter_env_powers(None, None, 4, 4, None, 1, None) == -1
    lib.rmdf_destroy(None)


def test_product_never_touches_the_oracle(rmdf):
    """The shipped package and library must not import, link or call anything under oracle/."""
    pkg = os.path.dirname(rmdf.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".hs", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "rmdf_oracle" not in text and "from oracle" not in text, f
                # ... nor know of the test doubles: the HIP double reaches a process by LD_PRELOAD only, the RCCL double by RMDF_RCCL_LIB
                # (cross-check build only); nothing the product ships names either
                assert "fake_hip" not in text and "libfake" not in text and "fake_rccl" not in text.replace("tests/fake_rccl.c", ""), f
    for lib_path in (rmdf.LIB_PATH, rmdf.XCHECK_LIB_PATH):
        deps = os.popen("ldd %s" % lib_path).read()
        assert "oracle" not in deps and "amdhip64" in deps
    # the product data directory ships the reflection map alone: no oracle-generated cache files
    # (cache files the product built at a first load may sit next to it; they are git-ignored)
    tracked = os.popen("git -C %s ls-files %s" % (ROOT, os.path.join(rmdf.DATA_DIR, "latlong_envmaps"))).read().split()
    assert [os.path.basename(t) for t in tracked] == ["uffizi_512.hdr"], tracked
    # ... and if cache files do sit there, they are what the real kernels build -- byte for byte the oracle's committed ones -- never the
    # stand-in maps of a run against the HIP test double (those runs load a private copy of the probe: tests/conftest.py)
    from conftest import ENV_CACHE
    d = os.path.join(rmdf.DATA_DIR, "latlong_envmaps")
    for f in os.listdir(d):
        if "_cache_pow_" in f:
            assert open(os.path.join(d, f), "rb").read() == open(os.path.join(ENV_CACHE, f), "rb").read(), f


def _build_c_host(tmp_path, name="c_host"):
    import subprocess
    import rmdf_amd
    rmdf_amd.build()
    exe = str(tmp_path / name)
    libdir = os.path.dirname(rmdf_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-std=c99", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-o", exe, "-L", libdir, "-lrmdf", "-Wl,-rpath," + libdir])
    return exe


def test_c_host_compiles_against_the_header(tmp_path):
    """include/rmdf.h is plain C (C99, -Wall -Werror) and librmdf.so links into a C program: the example host that does what
    App.draw does (64 drawShaderTile calls into a Word32 buffer, then the screenshot) builds without a GPU."""
    exe = _build_c_host(tmp_path)
    assert os.path.exists(exe)
    # the multi-GPU host: one process per GPU, unique id through a shared page, rmdf_comm_init, one call per frame
    assert os.path.exists(_build_c_host(tmp_path, "c_host_multi"))


def test_c_host_runs_against_the_hip_double(tmp_path):
    """The plain-C host without a GPU: its HIP runtime is the test double (tests/fake_hip.cpp, LD_PRELOAD), so its pixels mean
    nothing -- but the C program's own check `64 tiles == untiled` holds, it exits 0, the env pipeline writes its four cache files next to
    the (copied) light probe, and the PNG it writes decodes to the frame the Python mirror gets from the same double and the same
    files: the ABI as a C compiler sees it, the tile loop, the screenshot path, end to end on the CPU tier."""
    import shutil
    import subprocess
    import sys
    import rmdf_amd
    from test_host_logic import _fake_hip_lib
    exe = _build_c_host(tmp_path)
    hdr = str(tmp_path / "probe.hdr")
    shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, hdr)                     # (the double's prefilter is a stand-in: its cache files must not land in the tree)
    png = str(tmp_path / "c_host.png")
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib())
    out = subprocess.run([exe, hdr, png, "2", "320", "184"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "64 tiles == untiled: yes" in out.stdout
    assert len([f for f in os.listdir(str(tmp_path)) if "_cache_pow_" in f]) == 4
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import rmdf_amd\n"
            "from PIL import Image\n"
            "with rmdf_amd.with_shader_renderer(%r) as sr:\n"
            "    fb = rmdf_amd.FrameBuffer(320, 184)\n"
            "    sr.draw_shader_tile(2, None, 320, 184, 1.5, fb.vec, max_steps=256)\n"
            "print('same' if np.array_equal(np.asarray(Image.open(%r)), fb.to_image_rows_top_down()) else 'differs')\n" % (ROOT, hdr, png))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip() == "same", (r.stdout, r.stderr[-2000:])


@pytest.mark.parametrize("nranks", [2, 8])
def test_c_host_multi_runs_against_the_doubles(tmp_path, nranks):
    """examples/c_host_multi.c -- one forked process per GPU, the unique id through a shared page, rmdf_comm_init, the cost-aware deal,
    one rmdf_render_frame_sharded_device per frame -- with N > 1 ranks and no GPU: N devices of the HIP double (FAKE_HIP_DEVICES), the
    RCCL double between the processes (files in /dev/shm), the program linked against the cross-check library (the only one that honours
    RMDF_RCCL_LIB).  All ranks load the same light probe at once and build its four cache files concurrently (private name + rename:
    exactly four files afterwards, none temporary); the program's own check `sharded == single launch` holds and its PNG decodes to
    the frame a single renderer gets from the same double and the same files."""
    import shutil
    import subprocess
    import sys
    import rmdf_amd
    from test_host_logic import _fake_hip_lib
    rmdf_amd.build()
    libdir = os.path.dirname(rmdf_amd.XCHECK_LIB_PATH)
    exe = str(tmp_path / "c_host_multi")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_host_multi.c"),
                           "-o", exe, "-L", libdir, "-lrmdf_xcheck", "-Wl,-rpath," + libdir])
    fake_rccl = os.path.join(ROOT, "tests", "libfake_rccl.so")
    if not os.path.exists(fake_rccl) or os.path.getmtime(fake_rccl) < os.path.getmtime(os.path.join(ROOT, "tests", "fake_rccl.c")):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "fake_rccl.c"),
                               "-o", fake_rccl, "-L/opt/rocm/lib", "-lamdhip64"])
    hdr = str(tmp_path / "probe.hdr")
    shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, hdr)
    png = str(tmp_path / "multi.png")
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), FAKE_HIP_DEVICES=str(nranks), RMDF_RCCL_LIB=fake_rccl, FAKE_RCCL_TIMEOUT_S="120")
    out = subprocess.run([exe, hdr, png, str(nranks), "640", "360", "5"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sharded == single launch: yes" in out.stdout and ("rank 0 of %d" % nranks) in out.stdout, out.stdout
    files = sorted(os.listdir(str(tmp_path)))
    assert len([f for f in files if "_cache_pow_" in f]) == 4 and not [f for f in files if ".tmp" in f], files
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import rmdf_amd\n"
            "from PIL import Image\n"
            "with rmdf_amd.with_shader_renderer(%r) as sr:\n"
            "    fb = rmdf_amd.FrameBuffer(640, 360)\n"
            "    sr.draw_shader_tile(2, None, 640, 360, 0.0, fb.vec, max_steps=256)\n"
            "print('same' if np.array_equal(np.asarray(Image.open(%r)), fb.to_image_rows_top_down()) else 'differs')\n" % (ROOT, hdr, png))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, LD_PRELOAD=_fake_hip_lib()))
    assert r.returncode == 0 and r.stdout.strip() == "same", (r.stdout, r.stderr[-2000:])


@pytest.mark.gpu
def test_c_host_runs(tmp_path):
    """The C host on the GPU: 64 tiled calls accumulate the untiled frame (it checks that itself) and the PNG it writes
    decodes to the frame the Python mirror renders."""
    import subprocess
    import numpy as np
    from PIL import Image
    import rmdf_amd
    exe = _build_c_host(tmp_path)
    png = str(tmp_path / "c_host.png")
    out = subprocess.run([exe, rmdf_amd.DEFAULT_ENV_HDR, png, "2", "320", "184"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "64 tiles == untiled: yes" in out.stdout
    with rmdf_amd.with_shader_renderer() as sr:
        fb = rmdf_amd.FrameBuffer(320, 184)
        sr.draw_shader_tile(2, None, 320, 184, 1.5, fb.vec, max_steps=256)
    assert np.array_equal(np.asarray(Image.open(png)), fb.to_image_rows_top_down())


@pytest.mark.gpu
def test_c_host_multi_runs_with_one_rank(tmp_path):
    """examples/c_host_multi.c -- the N > 1 path (cost-aware deal, RCCL communicator, rmdf_render_frame_sharded_device) from
    plain C with no Python in the process -- with the one rank a 1-GPU box can hold: the sharded frame equals the
    single-launch frame (it checks that itself) and the PNG decodes to the frame the Python mirror renders."""
    import subprocess
    from PIL import Image
    import rmdf_amd
    exe = _build_c_host(tmp_path, "c_host_multi")
    png = str(tmp_path / "multi.png")
    out = subprocess.run([exe, rmdf_amd.DEFAULT_ENV_HDR, png, "1", "640", "360", "5"], capture_output=True, text=True, timeout=900)   # a fresh box pages librccl in (minutes, once)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sharded == single launch: yes" in out.stdout and "rank 0 of 1" in out.stdout
    with rmdf_amd.with_shader_renderer() as sr:
        fb = rmdf_amd.FrameBuffer(640, 360)
        sr.draw_shader_tile(2, None, 640, 360, 0.0, fb.vec, max_steps=256)
    assert np.array_equal(np.asarray(Image.open(png)), fb.to_image_rows_top_down())


def test_haskell_binding_matches_the_header():
    """hs/RmdfFFI.hs cannot be compiled here (no GHC), so at least its `foreign import ccall` declarations are held to include/rmdf.h: every
    imported symbol is declared there, with the same NUMBER of parameters and compatible types in every position -- CInt <-> int, CDouble
    <-> double, CSize <-> size_t, CString <-> (const) char *, Ptr x <-> any pointer, IO CInt / IO () / IO CString <-> int / void / const char *.
    A binding that drifts from the C ABI (a parameter added to rmdf_render_tile, say) would otherwise be found by the first maintainer with a GHC."""
    hs = open(os.path.join(ROOT, "ray-marching-distance-fields_amd", "hs", "RmdfFFI.hs")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "rmdf.h")).read(), flags=re.S)
    hdr = re.sub(r"//[^\n]*", " ", hdr)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(rmdf_\w+)\s*\(([^;{}]*?)\)\s*;", hdr):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, params)

    def c_kind(t):
        t = re.sub(r"\b(const|volatile|struct|restrict)\b", " ", t)
        if "*" in t or "[" in t:
            return "cstring" if re.search(r"\bchar\b", t) and t.count("*") == 1 and "[" not in t else "ptr"
        t = re.sub(r"\b\w+$", "", t.strip()).strip() or t.strip()               # drop the parameter name
        return {"int": "int", "double": "double", "float": "float", "size_t": "size", "void": "void", "unsigned": "uint", "uint32_t": "u32",
                "unsigned int": "uint", "uint64_t": "u64"}.get(t.strip(), t.strip())

    def hs_kind(t):
        t = t.strip()
        if t.startswith("Ptr") or t.startswith("FunPtr") or t.startswith("(Ptr"):
            return "ptr"
        return {"CInt": "int", "CDouble": "double", "CFloat": "float", "CSize": "size", "CString": "cstring", "CUInt": "uint", "Word32": "u32", "Word64": "u64",
                "()": "void"}.get(t, t)

    decls = re.findall(r'foreign import ccall (?:safe|unsafe)\s+"(\w+)"\s+\w+\s*::\s*((?:[^\n]|\n\s+(?=->|[A-Z(]))+)', hs)
    assert len(decls) >= 16, len(decls)
    for name, sig in decls:
        assert name in protos, "%s is not declared in include/rmdf.h" % name
        parts, depth, cur = [], 0, ""
        for tok in re.split(r"(\(|\)|->)", sig.replace("\n", " ")):
            if tok == "(":
                depth += 1
            elif tok == ")":
                depth -= 1
            if tok == "->" and depth == 0:
                parts.append(cur.strip()); cur = ""
            else:
                cur += tok
        parts.append(cur.strip())
        hs_args, hs_ret = parts[:-1], parts[-1]
        ret, params = protos[name]
        assert len(hs_args) == len(params), "%s: %d Haskell arguments, %d C parameters (%s)" % (name, len(hs_args), len(params), params)
        for i, (h, c) in enumerate(zip(hs_args, params)):
            hk, ck = hs_kind(h), c_kind(c)
            assert hk == ck or (hk == "ptr" and ck in ("ptr", "cstring")) or (hk == "cstring" and ck in ("ptr", "cstring")), "%s, parameter %d: %s vs %s" % (name, i, h, c)
        assert hs_ret.startswith("IO "), (name, hs_ret)
        hr, cr = hs_kind(hs_ret[3:].strip()), c_kind(ret + " x")
        assert hr == cr or (hr in ("ptr", "cstring") and cr in ("ptr", "cstring")), "%s: returns %s vs %s" % (name, hs_ret, ret)
