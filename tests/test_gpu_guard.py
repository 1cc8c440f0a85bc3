"""Round 5: the library's kernels under an electric-fence device allocator (rmdf_host.hpp: GuardAlloc, cross-check build only).  GPU
AddressSanitizer is not available on this pool; this is the next best thing and it tests the SHIPPED code paths: every device allocation
ends (RMDF_GUARD_ALLOC=end) or starts (=start) at an unmapped page, so one element read or written outside a buffer is a GPU memory
fault -- the process aborts -- instead of a silent access to whatever lies next to it."""
import os
import subprocess
import sys

import pytest

from conftest import unverified

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@unverified
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["end", "start"])
def test_no_kernel_touches_memory_outside_its_buffers(mode):
    env = dict(os.environ, RMDF_GUARD_ALLOC=mode)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "guard_workload.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "guard workload ok" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-3000:])


@unverified
@pytest.mark.gpu
def test_the_fence_does_catch_an_overrun():
    """The allocator is only evidence if it faults when it should: a kernel launched on a guarded buffer with one row too many must kill the
    child (the library's own resolve kernel, told that the source frame is two rows taller than the buffer it is given)."""
    code = ("import os, sys; sys.path.insert(0, %r); import torch, rmdf_amd, ctypes as C\n"
            "sr = rmdf_amd.ShaderRenderer(0, xcheck=True)\n"
            "L = rmdf_amd.load_library(True); p = C.c_void_p(); q = C.c_void_p()\n"
            "assert L.rmdf_device_malloc(sr.handle, 256 * 64 * 4, C.byref(p)) == 0 and L.rmdf_device_malloc(sr.handle, 128 * 33 * 4, C.byref(q)) == 0\n"
            "print('before', flush=True)\n"
            "sr.resolve_box2_device(p.value, 256, 66, q.value)\n"      # reads rows 64, 65 of a 64-row source
            "sr.synchronize(); print('survived', flush=True)\n" % ROOT)
    env = dict(os.environ, RMDF_GUARD_ALLOC="end")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert "before" in r.stdout and "survived" not in r.stdout and r.returncode != 0, (r.returncode, r.stdout, r.stderr[-1500:])
    assert "Memory access fault" in r.stderr or r.returncode < 0, r.stderr[-1500:]
