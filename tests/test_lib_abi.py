"""The C-ABI shared library must load and export every symbol include/espresso_amd.h declares
(no compute calls here — there is no GPU in this container)."""
import ctypes
import os

import pytest

from espresso_amd import _lib


def test_header_parses_and_library_exports_everything():
    protos = _lib.parse_header()
    assert len(protos) >= 30
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(handle, name), name
    assert _lib.lib().ea_version() >= 1


def test_kernels_refuse_cpu_tensors():
    import torch

    from espresso_amd import kernels as K

    with pytest.raises(RuntimeError):
        K.cast_f32_to_bf16(torch.zeros(8))


def test_gemm_params_struct_matches_header_layout():
    """Field order of the ctypes struct mirrors the C struct (checked by name against the header text)."""
    import re

    src = open(_lib.HEADER_PATH).read()
    body = re.search(r"typedef struct EaGemmParams \{(.*?)\} EaGemmParams;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        parts = stmt.replace("*", " ").split(",")
        first = parts[0].split()[-1]
        names.append(first)
        names += [p.strip() for p in parts[1:]]
    assert names == [f[0] for f in _lib.EaGemmParams._fields_]
