"""torch.cuda stand-ins over the HIP test double (tests/fake_hip.cpp): TEST INFRASTRUCTURE for processes that run without a GPU but want the few
torch.cuda entry points bench.py, the tools and the GPU tier's tests use -- "device" tensors are CPU tensors (the double's device memory IS host
memory), streams are the double's streams, events read the wall clock.  install() patches torch in place.  Used by tests/bench_dry_run.py and,
when the double runs the kernels (FAKE_HIP_EMULATE=1), by tests/conftest.py so that tests holding torch device buffers run on the emulated device."""
import ctypes as C
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def install():
    import torch
    if getattr(torch.cuda, "_rmdf_standins", False):
        return
    FAKE = C.CDLL(os.environ.get("FAKE_HIP_LIB", os.path.join(ROOT, "tests", "libfake_hip.so")))
    FAKE.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    FAKE.hipStreamSynchronize.argtypes = [C.c_void_p]


    class Stream:
        def __init__(self, device=None, priority=0):
            h = C.c_void_p()
            assert FAKE.hipStreamCreateWithFlags(C.byref(h), 1) == 0
            self.cuda_stream = h.value

        def synchronize(self):
            FAKE.hipStreamSynchronize(self.cuda_stream)

        def wait_stream(self, other): pass
        def wait_event(self, ev): pass
        def query(self): return True


    class Event:
        def __init__(self, enable_timing=False): self.t = None
        def record(self, stream=None): self.t = time.perf_counter()
        def synchronize(self): pass
        def query(self): return True
        def elapsed_time(self, other): return max((other.t - self.t) * 1e3, 1e-3)


    class _StreamCtx:
        def __init__(self, s): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False


    def _cpu(fn):
        def f(*a, **k):
            if k.get("device") is not None and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return f


    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 1
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda d=None: FAKE.hipDeviceSynchronize()
    torch.cuda.Stream = Stream
    torch.cuda.Event = Event
    torch.cuda.stream = _StreamCtx
    torch.cuda.set_stream = lambda s: None
    for name in ("empty", "zeros", "tensor", "ones", "full"):
        setattr(torch, name, _cpu(getattr(torch, name)))
    _real_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and str(a[0]).startswith("cuda")) or str(k.get("device", "")).startswith("cuda") else _real_to(self, *a, **k)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda._rmdf_standins = True
