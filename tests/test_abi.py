"""The C-ABI library: builds for gfx950, loads, and exports every symbol ``include/hip_nmf.h`` declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "hip_nmf.h")


@pytest.fixture(scope="module")
def lib():
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.build import build

    build()  # no-op when up to date
    return _lib.load()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hipnmf_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from muscle_synergies_amd import _lib

    assert declared_symbols() == sorted(_lib.EXPORTS)


def test_every_declared_symbol_is_exported(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_version_and_struct_layout(lib):
    from muscle_synergies_amd import _lib

    assert lib.hipnmf_version() == 212
    # struct hipnmf_problem: 4+4+8+4*6+8+8+4+4+8*5 bytes with natural alignment
    assert ctypes.sizeof(_lib.Problem) == 104
    text = open(HEADER).read()
    def struct_fields(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        groups = re.findall(r"^\s+(?:int32_t|int64_t|double)\s+([a-zA-Z0-9_, ]+);", body, flags=re.M)
        return [n.strip() for g in groups for n in g.split(",")]

    assert struct_fields("hipnmf_problem") == [f[0] for f in _lib.Problem._fields_]
    from muscle_synergies_amd.preprocess import EnvelopeParams

    assert struct_fields("hipnmf_envelope_params") == [f[0] for f in EnvelopeParams._fields_]
    assert ctypes.sizeof(EnvelopeParams) == 64
    from muscle_synergies_amd.preprocess import SosfiltParams

    assert struct_fields("hipnmf_sosfilt_params") == [f[0] for f in SosfiltParams._fields_]
    assert ctypes.sizeof(SosfiltParams) == 64


def test_no_device_is_a_loud_error(lib):
    import torch

    from muscle_synergies_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    rc = lib.hipnmf_create(0, ctypes.byref(h))
    assert rc == _lib.HIPNMF_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.hipnmf_last_error() or b"device" in lib.hipnmf_last_error()
    with pytest.raises(_lib.HipNmfError):
        _lib.Handle(0)


def test_workspace_query_is_pure(lib):
    from muscle_synergies_amd.engine import make_problem
    from muscle_synergies_amd import _lib

    p = make_problem(4096, 10000, 16, 5, x_layout=_lib.X_ROW_MAJOR, ldx=16, x_batch_stride=160000)
    n = lib.hipnmf_workspace_bytes(ctypes.byref(p), 4)
    assert n >= 4 * 4096 * 21 * 10000
    assert lib.hipnmf_workspace_bytes(ctypes.byref(p), 3) == 0
