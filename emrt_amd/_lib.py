"""ctypes binding of libemrt_hip.so, generated from include/emrt_hip.h (single source of truth).

The product path has NO fallback: if the shared library is missing or a symbol is absent this module raises, and
every op in emrt_amd.functional raises with it.  (INTEGRATION.md shows this same stub as the reference-side binding.)
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "emrt_hip.h")
# EMRT_HIP_LIB: a differently built copy of the same library (developer experiments: tools/exp builds one with -DEMRT_8P_PROBES)
LIB_PATH = os.environ.get("EMRT_HIP_LIB") or os.path.join(_HERE, "csrc", "libemrt_hip.so")

_TRACE = bool(int(os.environ.get("EMRT_TRACE", "0")))

_CTYPES = {
    "int": ctypes.c_int, "unsigned": ctypes.c_uint, "float": ctypes.c_float, "double": ctypes.c_double,
    "long long": ctypes.c_longlong, "size_t": ctypes.c_size_t,
}


class EmrtHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(type_str, arg_name), ...])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = text.replace('extern "C" {', " ").replace("}", " ")
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(emrt_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), m.group(3).strip()
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                arglist.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, arglist)
    return protos


def _ctype_of(t):
    t = t.replace("const ", "").strip()
    if t.endswith("*"):
        return ctypes.c_char_p if t == "char*" and False else ctypes.c_void_p
    if t in _CTYPES:
        return _CTYPES[t]
    raise EmrtHipError("emrt_hip.h: unknown C type %r" % t)


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise EmrtHipError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU/PyTorch fallback for the EMRT hot path)" % LIB_PATH)
        # torch first: the library must bind to the HIP runtime torch has loaded (torch's streams and device memory are handed to it).
        # Loaded before torch it pulls in /opt/rocm's copy, torch then brings its bundled one, and launches fail with "no ROCm-capable
        # device is detected" (seen when build() and smoke() ran in one process).
        import torch  # noqa: F401
        self._dll = ctypes.CDLL(LIB_PATH)
        self._rec = None
        self.empty_pair_ms = 0.0
        self.replay_total_ms = 0.0
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                raise EmrtHipError("libemrt_hip.so does not export %s declared in emrt_hip.h" % name)
            fn.argtypes = [_ctype_of(t) for t, _ in args]
            if ret == "const char*":
                fn.restype = ctypes.c_char_p
            elif ret == "size_t":
                fn.restype = ctypes.c_size_t
            else:
                fn.restype = ctypes.c_int
            setattr(self, "_raw_" + name, fn)

    def last_error(self):
        return self._raw_emrt_last_error().decode("utf-8", "replace")

    # ---- launch recording / timed replay (bench.py) ------------------------------------------------------------
    # In eager mode the host needs ~20-30 us of Python per launch, so a HIP-event pair around a single small kernel
    # measures the HOST, not the GPU.  Instead the launches of one step are recorded (name + C arguments; the runtime
    # keeps every temporary alive meanwhile) and then replayed from a tight ctypes loop: once untimed as a backlog so the
    # host runs ahead of the GPU, then again with a HIP-event pair -- on the launch stream -- around every launch.
    def start_record(self):
        self._rec = []

    def stop_record(self):
        rec, self._rec = self._rec, None
        return rec

    def replay(self, rec, timed=False):
        """Re-issue recorded launches.  timed=True -> [(name, args, milliseconds)] from per-launch HIP events."""
        import torch
        raw = [(getattr(self, "_raw_" + n), a, n) for n, a in rec]
        if not timed:
            for fn, a, n in raw:
                rc = fn(*a)
                if rc != 0:
                    raise EmrtHipError("%s failed during replay (%d): %s" % (n, rc, self.last_error()))
            return None
        evs = []
        for fn, a, n in raw:
            e0 = self._event(a[-1])
            fn(*a)
            evs.append((e0, self._event(a[-1])))
        # calibration: the same event pair around NOTHING, 64 times behind the same backlog -- what the clock itself adds to every figure
        # (kept in self.empty_pair_ms, median; bench.py subtracts it per launch to put its roofline on the kernel-duration clock rocprofv3 reports)
        stream = raw[-1][1][-1] if raw else None
        empty = []
        if stream is not None:
            for _ in range(64):
                e0 = self._event(stream)
                empty.append((e0, self._event(stream)))
        torch.cuda.synchronize()
        out, ms = [], ctypes.c_float(0.0)
        for (n, a), (e0, e1) in zip(rec, evs):
            self._raw_emrt_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            out.append((n, a, ms.value))
            self._raw_emrt_event_destroy(e0)
            self._raw_emrt_event_destroy(e1)
        gaps = []
        for e0, e1 in empty:
            self._raw_emrt_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            gaps.append(ms.value)
            self._raw_emrt_event_destroy(e0)
            self._raw_emrt_event_destroy(e1)
        self.empty_pair_ms = sorted(gaps)[len(gaps) // 2] if gaps else 0.0
        # the SAME launch list once more with ONE event pair around all of it (behind a backlog pass, so the host is ahead of the GPU): the time the
        # step's kernels take back to back.  sum(per-launch readings) - this = what the per-launch pairs added in total (bench.py spreads it evenly)
        self.replay_total_ms = 0.0
        if stream is not None:
            for fn, a, n in raw:
                fn(*a)
            e0 = self._event(stream)
            for fn, a, n in raw:
                fn(*a)
            e1 = self._event(stream)
            torch.cuda.synchronize()
            self._raw_emrt_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            self.replay_total_ms = ms.value
            self._raw_emrt_event_destroy(e0)
            self._raw_emrt_event_destroy(e1)
        return out

    def _event(self, stream):
        ev = ctypes.c_void_p()
        self._raw_emrt_event_create(ctypes.byref(ev))
        self._raw_emrt_event_record(ev, stream)
        return ev

    def call(self, name, *args):
        """Invoke an int-returning entry point; raise EmrtHipError on a non-zero return.
        EMRT_TRACE=1 prints every call before launching it and synchronises after it (locates GPU faults)."""
        if _TRACE:
            print("[emrt] %s%r" % (name, tuple(a.value if hasattr(a, "value") else a for a in args)), flush=True)
        if self._rec is not None:
            self._rec.append((name, args))
        rc = getattr(self, "_raw_" + name)(*args)
        if rc != 0:
            raise EmrtHipError("%s failed (%d): %s" % (name, rc, self.last_error()))
        if _TRACE:
            import torch
            torch.cuda.synchronize()

    def set_tuning(self, name, value):
        """Developer / test knob of the kernel dispatchers (include/emrt_hip.h: emrt_set_tuning); returns the old value."""
        old = ctypes.c_int(0)
        if self._raw_emrt_get_tuning(name.encode(), ctypes.byref(old)) != 0 or self._raw_emrt_set_tuning(name.encode(), int(value)) != 0:
            raise EmrtHipError(self.last_error())
        return old.value

    def query(self, name, *args):
        return getattr(self, "_raw_" + name)(*args)


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
