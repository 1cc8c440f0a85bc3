"""CPU tier: a lint over the GPU tier's own sources.  torch fills / copies run on torch's current stream; the library renders on its
own NON-blocking streams (or on a torch.cuda.Stream the test hands it), which do not wait for torch's.  A test that writes a device
buffer through torch and then hands it to the library without torch.cuda.synchronize() in between races the fill against the
kernel (VERDICT r05 "What's weak" 2: five all-zero frames in 91 tier runs under load came from exactly that).  This test fails on
the pattern: a torch-side device write followed, in the same function, by a library call taking a device pointer or a stream before
any synchronize()."""
import ast
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_SOURCES = sorted(glob.glob(os.path.join(ROOT, "tests", "test_gpu_*.py"))) + [
    os.path.join(ROOT, "tests", "fake_rccl_worker.py"), os.path.join(ROOT, "tests", "guard_workload.py"),
    os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]

WRITE = re.compile(r"\.zero_\(|\.fill_\(|\.copy_\(|torch\.(zeros|ones|full|tensor|arange|rand\w*|zeros_like|ones_like|full_like)\(|\.cuda\(\)|\.to\(\s*['\"]?cuda|\.to\(dev")
SYNC = re.compile(r"synchronize\(|dev_zeros\(|\.item\(\)|\.cpu\(\)")    # .item() / .cpu() wait for torch's stream
HANDOVER = re.compile(r"data_ptr\(\)|cuda_stream|stream=")


def _offences(path):
    src = open(path).read()
    lines = src.split("\n")
    out = []
    for fn in ast.walk(ast.parse(src)):
        if not isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef)) or fn.name == "dev_zeros":
            continue
        dirty = None                                   # line number of the torch-side write nothing has waited for yet
        for no in range(fn.lineno, fn.end_lineno + 1):
            ln = lines[no - 1].split("#")[0]
            if dirty is not None and HANDOVER.search(ln) and not SYNC.search(ln):
                # a line may both write and hand over ("x = torch.zeros(...); f(x.data_ptr())" on one line is still an offence)
                out.append("%s:%d hands a buffer to the library; torch wrote at line %d and nothing synchronised since"
                           % (os.path.relpath(path, ROOT), no, dirty))
                dirty = None
            if SYNC.search(ln):
                dirty = None
            elif WRITE.search(ln) and "device=" in ln or re.search(r"\.zero_\(|\.fill_\(|\.copy_\(|\.cuda\(\)", ln):
                if not SYNC.search(ln):
                    dirty = no
    return out


def test_no_gpu_test_hands_over_a_buffer_torch_may_still_be_writing():
    bad = []
    for p in GPU_SOURCES:
        if os.path.exists(p):
            bad += _offences(p)
    assert not bad, "\n".join(bad)


def test_the_lint_sees_the_pattern(tmp_path):
    p = tmp_path / "t.py"
    p.write_text("def f(r, torch, st):\n    frame = torch.zeros(4, device='cuda')\n    frame.zero_()\n"
                 "    r.render(frame.data_ptr(), stream=st.cuda_stream)\n")
    assert len(_offences(str(p))) == 1
    p.write_text("def f(r, torch, st):\n    frame = torch.zeros(4, device='cuda')\n    frame.zero_()\n    torch.cuda.synchronize()\n"
                 "    r.render(frame.data_ptr(), stream=st.cuda_stream)\n")
    assert _offences(str(p)) == []
