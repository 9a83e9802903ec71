"""CPU: the C-ABI shared library loads and exports exactly what include/emrt_hip.h declares; the header's prototypes are
the definitions' prototypes (no compute calls -- there is no GPU here)."""
import ctypes
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from emrt_amd import build_ext
    return build_ext.build(verbose=False)


def test_library_exports_every_declared_symbol(built):
    from emrt_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 50
    out = subprocess.run(["nm", "-D", "--defined-only", built], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [n for n in protos if n not in exported]
    assert not missing, missing
    undeclared = [n for n in exported if n.startswith("emrt_") and n not in protos]
    assert not undeclared, "exported but not declared in emrt_hip.h: %s" % undeclared


def test_header_matches_definitions():
    from emrt_amd import _lib
    protos = _lib.parse_header()
    defs = {}
    for f in glob.glob(os.path.join(ROOT, "emrt_amd/csrc/*.hip")):
        s = open(f).read()
        for m in re.finditer(r'extern "C" ([^{;]+?)\{', s, re.S):
            p = re.sub(r"/\*.*?\*/", "", " ".join(m.group(1).split()))
            name = re.search(r"(\w+)\s*\(", p).group(1)
            args = p[p.index("(") + 1:p.rindex(")")]
            types = [] if args.strip() in ("", "void") else [re.match(r"(.*?)(\w+)$", " ".join(a.split())).group(1).strip() for a in args.split(",")]
            defs[name] = types
    assert set(defs) == set(protos)
    for name, (ret, args) in protos.items():
        assert [t for t, _ in args] == defs[name], name


def test_loader_binds_all_entry_points(built):
    from emrt_amd import _lib
    _lib._LIB = None
    L = _lib.lib()
    assert L.query("emrt_abi_version") == 9
    assert L.query("emrt_colreduce_workspace_bytes", 1000, 256) > 0
    import ctypes
    small = (ctypes.c_int * 6)(2, 2, 2, 2, 1, 2)                   # Lv = 10
    base = lambda B, Lq, M: ((B * Lq * M * 18 + 512 + 2 * B * M + B * M * max((Lq + 63) // 64, 64) + 3) // 4 * 4) * 4     # probabilities + max |dout| partials (>= 64 per (batch, head): row bands)
    assert L.query("emrt_msda_bwd_workspace_bytes", 2, 10, 8, 3, 6, ctypes.cast(small, ctypes.c_void_p), 1) == base(2, 10, 8)
    # batch 8 at 256x256 (Lv = 1344): no range of the value-gradient scatter is split by queries -> no partial slabs
    cfg2 = (ctypes.c_int * 6)(32, 32, 16, 16, 8, 8)
    assert L.query("emrt_msda_bwd_workspace_bytes", 8, 1344, 8, 3, 6, ctypes.cast(cfg2, ctypes.c_void_p), 1) == base(8, 1344, 8)
    # batch 4 at 512x512 (Lv = 5376), bf16: level 1 (2 ranges x 3 query splits) and level 2 (5 splits) leave int32 partial slabs;
    # the fp32 path never splits
    cfg3 = (ctypes.c_int * 6)(64, 64, 32, 32, 16, 16)
    part = 4 * 8 * (2 * 3 * 512 * 32 + 5 * 256 * 32) * 4
    assert L.query("emrt_msda_bwd_workspace_bytes", 4, 5376, 8, 3, 6, ctypes.cast(cfg3, ctypes.c_void_p), 1) == base(4, 5376, 8) + part
    assert L.query("emrt_msda_bwd_workspace_bytes", 4, 5376, 8, 3, 6, ctypes.cast(cfg3, ctypes.c_void_p), 0) == base(4, 5376, 8)
    # without the level shapes the size would leave the partial slabs out -> an error, not a too-small answer
    assert L.query("emrt_msda_bwd_workspace_bytes", 4, 5376, 8, 3, 6, None, 1) == 0 and "level shapes" in L.last_error()
    assert L.query("emrt_msda_bwd_workspace_bytes", 4, 5376, 8, 5, 6, ctypes.cast(cfg3, ctypes.c_void_p), 1) == 0
    # adaptive pooling: no workspace for small maps / odd channel counts; 256x256 tiles (32x32 map, scales 1 2 3 6): 8 + 4*2 + 9 + 36 slots
    sc = (ctypes.c_int * 4)(1, 2, 3, 6)
    scp = ctypes.cast(sc, ctypes.c_void_p)
    assert L.query("emrt_adaptive_avgpool_workspace_bytes", 16, 16, 8, 256, scp, 4) == 0
    assert L.query("emrt_adaptive_avgpool_workspace_bytes", 32, 32, 8, 250, scp, 4) == 0
    assert L.query("emrt_adaptive_avgpool_workspace_bytes", 32, 32, 8, 256, scp, 4) == 8 * (9 + 4 * 3 + 9 * 2 + 36) * 256 * 4
    assert L.query("emrt_adaptive_avgpool_workspace_bytes", 32, 32, 8, 256, None, 4) == 0


def test_abi6_entry_points_refuse_bad_arguments_before_any_launch(built):
    """Argument checks of the round-4 entry points run on the host before anything touches a device: a bad call returns non-zero with a
    message (no GPU here; the pointers are never dereferenced)."""
    from emrt_amd import _lib
    _lib._LIB = None
    L = _lib.lib()
    p = ctypes.c_void_p(0x10000)          # "a device pointer": aligned, never read
    bn = (p, 64.0, 1e-5, 0.9, p, p, p, p, p, p, 1)      # sums, count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, relu

    def refused(name, *args, match):
        with pytest.raises(_lib.EmrtHipError, match=match):
            L.call(name, *args)

    # classifier: C outside {64, 128, 256}; more than 8 outputs
    refused("emrt_bn_pointwise_fwd", p, 96, 96 * 64, p, None, p, 6, 6 * 64, 1, 64, 96, 6, *bn, 1, None, match="C in")
    refused("emrt_bn_pointwise_fwd", p, 256, 256 * 64, p, None, p, 9, 9 * 64, 1, 64, 256, 9, *bn, 1, None, match="OC <= 8")
    refused("emrt_bn_pointwise_bwd", p, 256, 256 * 64, p, 9, 9 * 64, p, p, 256, 256 * 64, p, None, None, 1, 64, 256, 9, p, p, p, p, 1, None, match="OC <= 8")
    # resize / max-pool with a BatchNorm operand: off the vector path, missing sums, non-positive count
    refused("emrt_bn_resize_bilinear_fwd", p, 6 * 16, 6, 4, 4, p, 6 * 64, 6, 8, 8, 1, 6, 0, *bn, 1, None, match="vector path")
    nosums = (None,) + bn[1:]
    refused("emrt_bn_resize_bilinear_fwd", p, 64 * 16, 64, 4, 4, p, 64 * 64, 64, 8, 8, 1, 64, 0, *nosums, 1, None, match="BatchNorm operand")
    zerocount = (p, 0.0) + bn[2:]
    refused("emrt_bn_maxpool_fwd", p, p, p, 1, 8, 8, 64, 3, 2, 1, *zerocount, 1, None, match="BatchNorm operand")
    refused("emrt_bn_maxpool_fwd", p, p, p, 1, 8, 8, 60, 3, 2, 1, *bn, 1, None, match="vector path")
    # the join: the shortcut's BatchNorm is not optional
    refused("emrt_bn_apply_join", p, 64, p, 64, p, 64, p, 64.0, 1e-5, 0.9, p, p, p, p, p, p, None, 64.0, 1e-5, 0.9, p, p, p, p, p, p, 64, 64, 1, 1, None,
            match="shortcut")
    # mask arguments come as a pair and replace y
    refused("emrt_bn_bwd_reduce", p, 64, p, 64, None, 0, p, p, 64, 64, p, p, None, 1, None, match="come together")
    refused("emrt_bn_bwd_reduce", p, 64, p, 64, p, 64, p, p, 64, 64, p, p, p, 1, None, match="replace y")
    refused("emrt_bn_bwd_dx", p, 64, p, 64, p, 64, p, 64, None, 64, p, p, p, p, None, 64.0, None, None, 64, 64, None, 0, p, 1, None, match="mask_beta")
    # pooling workspace smaller than the query's answer
    sc = (ctypes.c_int * 4)(1, 2, 3, 6)
    scp = ctypes.cast(sc, ctypes.c_void_p)
    need = L.query("emrt_adaptive_avgpool_workspace_bytes", 32, 32, 8, 256, scp, 4)
    refused("emrt_adaptive_avgpool_fwd", p, 32 * 32 * 256, 256, 32, 32, p, 50 * 256, 256, 8, 256, scp, 4, p, need - 4, 1, None, match="workspace smaller")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from emrt_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(_lib.EmrtHipError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    for f in glob.glob(os.path.join(ROOT, "emrt_amd/**/*.py"), recursive=True):
        src = open(f).read()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_build_is_keyed_on_the_content_of_its_inputs(tmp_path):
    """build_ext decides what to rebuild from a sha256 of the sources, headers, compiler path and flags recorded beside every object and the
    library (not from modification times, which mean nothing once the built files have travelled to another machine): an untouched tree is
    "reused" whatever the timestamps say, and the recorded digest is the digest of the tree."""
    import os
    from emrt_amd import build_ext
    build_ext.build(verbose=False)
    assert build_ext.LAST_BUILD["mode"] in ("reused", "linked", "compiled")
    # timestamps shuffled: the sources now look newer than the objects -- still nothing to do
    for src in build_ext.SOURCES:
        os.utime(os.path.join(build_ext.CSRC, src), None)
    build_ext.build(verbose=False)
    assert build_ext.LAST_BUILD["mode"] == "reused" and build_ext.LAST_BUILD["compiled"] == []
    assert build_ext.LAST_BUILD["digest"] == build_ext.source_digest()
    with open(build_ext.LIB + ".inputs") as f:
        assert f.read().strip() == build_ext.source_digest()
    # a different flag set (or source byte) is a different digest
    assert build_ext._digest([os.path.join(build_ext.CSRC, build_ext.SOURCES[0])], ["x"]) != build_ext._digest([os.path.join(build_ext.CSRC, build_ext.SOURCES[0])], ["y"])
