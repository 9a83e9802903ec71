"""Builds emrt_amd/csrc/libemrt_hip.so with hipcc for gfx950 (in-tree; the .so travels with gpurun snapshots).

What is rebuilt is decided by CONTENT: every object and the library carry a `<file>.inputs` stamp with the sha256 of the sources, headers,
compiler path and flags they were built from; "reused" therefore means "built from exactly these inputs", on whatever machine.

hipcc cross-compiles without a GPU, so this runs in the build container as the "does it build" check."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libemrt_hip.so")
SOURCES = ["conv.hip", "norm.hip", "msda.hip", "attn.hip", "spatial.hip", "loss_optim.hip", "elementwise.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(paths, extra=()):
    """sha256 over the CONTENT of the given files (in the given order) and the extra strings (compiler, flags)."""
    import hashlib
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode() + b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _stamp(target):
    return target + ".inputs"


def _fresh(target, digest):
    """The object / library exists and was built from exactly these inputs (content hash recorded beside it): modification times are not
    trusted -- the built files travel between machines (gpurun snapshots, the driver's boxes) where they mean nothing."""
    try:
        with open(_stamp(target)) as f:
            return os.path.exists(target) and f.read().strip() == digest
    except OSError:
        return False


def source_digest():
    """Content hash of everything libemrt_hip.so is built from: csrc/*.hip, csrc/*.hpp, the compiler path and the flags."""
    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".hpp")]
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + hdrs, [_hipcc()] + FLAGS)


LAST_BUILD = {"mode": None, "digest": None, "compiled": []}      # what the last build() call did: "compiled" | "linked" | "reused"


def build(force=False, verbose=True):
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".hpp")]      # common.hpp, igemm8p.hpp (included by conv.hip)
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        d = _digest([s] + hdrs, [hipcc] + FLAGS)
        if force or not _fresh(o, d):
            jobs.append(([hipcc] + FLAGS + ["-c", s, "-o", o], o, d))

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))

    def compile_one(job):
        cmd, o, d = job
        if os.path.exists(_stamp(o)):
            os.remove(_stamp(o))
        run(cmd)
        with open(_stamp(o), "w") as f:
            f.write(d + "\n")
        return cmd[-3]

    LAST_BUILD["compiled"] = []
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for done in ex.map(compile_one, jobs):
                LAST_BUILD["compiled"].append(os.path.basename(done))
                if verbose:
                    print("[emrt_amd.build] compiled", os.path.basename(done), flush=True)
    total = source_digest()
    if force or jobs or not _fresh(LIB, total):
        if os.path.exists(_stamp(LIB)):
            os.remove(_stamp(LIB))
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
        with open(_stamp(LIB), "w") as f:
            f.write(total + "\n")
        LAST_BUILD["mode"] = "compiled" if jobs else "linked"
        if verbose:
            print("[emrt_amd.build] linked", LIB, flush=True)
    else:
        LAST_BUILD["mode"] = "reused"
    LAST_BUILD["digest"] = total
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
