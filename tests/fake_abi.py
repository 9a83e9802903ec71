"""Recording stand-in for libemrt_hip.so used by the CPU tests of the HOST logic only (shapes, strides, tape order,
argument marshalling).  It computes nothing: every entry point validates its argument count/types against
include/emrt_hip.h and logs the call.  The product never uses it."""
import ctypes

import torch

from emrt_amd import _lib
from emrt_amd import runtime


class FakeLib:
    def __init__(self):
        self.protos = _lib.parse_header()
        self.calls = []

    def _check(self, name, args):
        ret, spec = self.protos[name]
        assert len(args) == len(spec), "%s: %d args given, header declares %d" % (name, len(args), len(spec))
        for a, (t, an) in zip(args, spec):
            if t.endswith("*"):
                assert a is None or isinstance(a, (ctypes.c_void_p, int)) or hasattr(a, "_type_"), "%s.%s: bad pointer %r" % (name, an, a)
            elif t in ("float", "double"):
                assert isinstance(a, (int, float)), "%s.%s: expected number, got %r" % (name, an, type(a))
            else:
                assert isinstance(a, int) and not isinstance(a, bool) or isinstance(a, bool), "%s.%s (%s): expected int, got %r" % (name, an, t, type(a))

    def call(self, name, *args):
        self._check(name, args)
        self.calls.append((name, args))

    def query(self, name, *args):
        self._check(name, args)
        return 1 << 16

    def last_error(self):
        return ""


def install():
    """Point the runtime at CPU tensors and the fake ABI.  Returns the FakeLib."""
    fake = FakeLib()
    _lib._LIB = fake
    c = runtime.ctx()
    c.device = torch.device("cpu")
    c.dtype = runtime.F32
    c._ws = torch.empty(1 << 20, dtype=torch.uint8)
    c._seed = torch.zeros(1, dtype=torch.int64)
    c.step_counter = torch.zeros(1, dtype=torch.int64)
    c.tape = None
    c.training = False
    c.world_size = 1
    type(c).stream = property(lambda self: ctypes.c_void_p(0))
    return fake


def uninstall():
    _lib._LIB = None
    c = runtime.ctx()
    type(c).stream = property(lambda self: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
