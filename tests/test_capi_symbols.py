"""CPU checks of the drop-in boundary: the shared library loads without a GPU, exports every symbol that
include/oq_hip.h declares, and the ctypes signatures agree with the header (argument counts and kinds)."""
import ctypes
import os
import re

from conftest import ROOT
from omniquant_amd import _capi

HEADER = os.path.join(ROOT, "include", "oq_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(oq_\w+)\s*\(([^)]*)\)\s*;", src):
        args = [a.strip() for a in m.group(3).split(",") if a.strip() and a.strip() != "void"]
        decls[m.group(2)] = args
    return decls


def test_library_loads_and_exports_every_declared_symbol():
    lib = _capi.load()
    decls = _declared()
    assert len(decls) >= 20
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in include/oq_hip.h but not exported"
    assert lib.oq_version() >= 100
    assert isinstance(lib.oq_last_error(), bytes)


def test_ctypes_signatures_match_header():
    decls = _declared()
    kinds = {ctypes.c_void_p: "ptr", ctypes.c_int64: "int64_t", ctypes.c_int: "int", ctypes.c_float: "float"}
    for name, argtypes in list(_capi.SIGNATURES.items()) + list(_capi.SIZE_FUNCS.items()):
        assert name in decls, f"{name} bound in _capi.py but not declared in the header"
        hargs = decls[name]
        assert len(hargs) == len(argtypes), f"{name}: header has {len(hargs)} args, binding {len(argtypes)}"
        for h, a in zip(hargs, argtypes):
            k = kinds[a]
            if k == "ptr":
                assert "*" in h, f"{name}: '{h}' bound as pointer"
            else:
                assert "*" not in h and h.startswith(k + " "), f"{name}: '{h}' bound as {k}"
    for name in decls:
        assert name in _capi.SIGNATURES or name in _capi.SIZE_FUNCS or name in ("oq_version", "oq_last_error"), \
            f"{name} not bound"


def test_missing_gpu_is_loud():
    """No silent CPU fallback: CPU tensors are refused by the product path."""
    import pytest
    import torch
    from omniquant_amd import ops, OQError
    with pytest.raises(OQError):
        ops.fake_quant(torch.randn(4, 64), 4)
    with pytest.raises(OQError):
        ops.LinearFn.apply(torch.randn(4, 64), torch.randn(8, 64), None)
