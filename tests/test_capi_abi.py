"""The drop-in boundary: libvaspfsi.so loads and exports every entry point include/vaspfsi.h declares; the ctypes
structs match the header; without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import re

import numpy as np
import pytest

from conftest import ROOT
from vasp_amd import capi

HEADER = (ROOT / "include" / "vaspfsi.h").read_text()


def declared_functions():
    return sorted(set(re.findall(r"^\s*(?:int|int64_t|void|const char\*)\s+(fsi_\w+)\s*\(", HEADER, flags=re.M)))


def test_every_declared_symbol_is_exported():
    lib = capi.load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vaspfsi.h but not exported"
    assert set(names) == set(capi.EXPORTED_SYMBOLS)


def test_struct_layouts_match_header():
    # field counts and sizes (all members are 8-byte aligned or explicitly int32 pairs)
    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), HEADER, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [part.strip().split()[-1].lstrip("*") for f in body.split(";") if f.strip() for part in f.split(",")]
    for name, cls in (("FsiMeshDesc", capi.FsiMeshDesc), ("FsiParams", capi.FsiParams),
                      ("FsiNewtonOpts", capi.FsiNewtonOpts), ("FsiNewtonIter", capi.FsiNewtonIter), ("FsiTimers", capi.FsiTimers),
                      ("FsiTuning", capi.FsiTuning),
                      ("FsiStepStats", getattr(capi, "FsiStepStats", None))):
        if cls is None:
            continue
        assert [f for f, _ in cls._fields_] == fields(name), name
    assert ctypes.sizeof(capi.FsiNewtonIter) == 32


def test_tuning_defaults_and_environment_overrides(monkeypatch):
    """FsiTuning (round 4): one documented struct instead of ~80 getenv switches.  The library fills the
    defaults, FSI_<NAME> overrides a field in ONE place (csrc/fsi_tuning.hip), the ctypes struct has the library's size, and
    no product option is read from the environment anywhere else in the library's sources."""
    lib = capi.load_library()
    t = capi.FsiTuning()
    lib.fsi_tuning_defaults(ctypes.byref(t))
    assert t.struct_size == ctypes.sizeof(capi.FsiTuning)
    d = t.as_dict()
    assert (d["krylov_fp32"], d["krylov_capacity"], d["prec_streams"], d["its_schur"], d["newton_forcing"]) == (2, 600, 1, 30, 1e-2)
    assert d["newton_forcing_late"] == 3e-3 and d["assembly_atomic"] == 0 and d["node_order"] == 0 and d["jacobian_mfma"] == 0
    monkeypatch.setenv("FSI_KRYLOV_FP32", "0")
    monkeypatch.setenv("FSI_ORDER", "colour")
    monkeypatch.setenv("FSI_ASSEMBLY", "atomic")
    monkeypatch.setenv("FSI_CHEB_P", "24")
    monkeypatch.setenv("FSI_NEWTON_FORCING_LATE", "1e-3")
    e = capi.FsiTuning()
    lib.fsi_tuning_from_env(ctypes.byref(e))
    assert (e.krylov_fp32, e.node_order, e.assembly_atomic, e.its_schur, e.newton_forcing_late) == (0, 2, 1, 24, 1e-3)
    assert e.krylov_capacity == 600                                        # untouched fields keep their defaults
    # the only getenv calls outside fsi_tuning.hip are debugging aids
    for src in (ROOT / "vasp_amd" / "csrc").glob("*.h*"):
        if src.name == "fsi_tuning.hip":
            continue
        for m in re.finditer(r'getenv\("(\w+)"\)', src.read_text()):
            assert m.group(1).startswith("FSI_DEBUG") or m.group(1) == "FSI_RCCL_HOST_REDUCE", (src.name, m.group(1))


def test_tuning_struct_size_is_honoured_in_both_directions():
    """a caller compiled against a shorter (older) FsiTuning announces its sizeof in struct_size; the three functions
    that fill a caller's buffer write at most that many bytes and report how many they wrote."""
    lib = capi.load_library()
    full = ctypes.sizeof(capi.FsiTuning)
    short = capi.FsiTuning.krylov_capacity.offset + 4          # an "old" struct that ends behind krylov_capacity
    for fill in (lib.fsi_tuning_defaults, lib.fsi_tuning_from_env,
                 lambda p: lib.fsi_tuning_copy_out(ctypes.byref(ref), p)):
        ref = capi.FsiTuning()
        lib.fsi_tuning_defaults(ctypes.byref(ref))
        buf = (ctypes.c_ubyte * full)(*([0xAB] * full))
        t = ctypes.cast(buf, ctypes.POINTER(capi.FsiTuning))
        t.contents.struct_size = short
        fill(t)
        assert t.contents.struct_size == short and t.contents.krylov_fp32 == 2 and t.contents.krylov_capacity == 600
        assert bytes(buf[short:]) == bytes([0xAB] * (full - short))          # nothing behind the caller's struct was touched
        for bad in (0, -5, full + 64):                                        # "this header's size"
            t.contents.struct_size = bad
            fill(t)
            assert t.contents.struct_size == full and t.contents.mg_cits == 16


def test_error_codes_match_header():
    for code, name in capi.ERROR_NAMES.items():
        assert re.search(r"#define %s %d\b" % (name, code), HEADER)


def test_no_gpu_means_loud_failure(cylinder_case):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.FsiError) as e:
        capi.HipBackend(cylinder_case[1])
    assert e.value.code == 2           # FSI_ERR_DEVICE


def test_create_rejects_bad_input():
    lib = capi.load_library()
    ctx = ctypes.c_void_p()
    coords = np.zeros((4, 3))
    tn = np.arange(10, dtype=np.int32)[None]
    kind = np.array([5], dtype=np.int32)           # neither fluid (0) nor solid (1)
    reg = np.zeros(1, dtype=np.int32)
    md = capi.FsiMeshDesc(4, 10, 1, capi._ptr(coords), capi._ptr(tn), capi._ptr(kind), capi._ptr(reg))
    fp = np.array([[1.0, 1.0]])
    sp = np.array([[1.0, 1.0, 1.0]])
    sm = np.zeros(1, dtype=np.int32)
    pr = capi.FsiParams(1e-3, 0.5, 1, capi._ptr(fp), 1, capi._ptr(sp), capi._ptr(sm), 1e7, 1.0)
    rc = lib.fsi_create(ctypes.byref(md), ctypes.byref(pr), 0, ctypes.byref(ctx))
    assert rc == 1 and b"kind" in lib.fsi_last_error(ctx)
    lib.fsi_destroy(ctx)
    assert lib.fsi_create(None, None, 0, ctypes.byref(ctx)) == 1
