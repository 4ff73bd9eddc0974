#!/usr/bin/env python3
"""Drives the host side of the C ABI (argument validation, error strings, handle life cycle, workspace query)
without a GPU.  tests/test_abi_sanitizers.py runs it against the ASan + UBSan build of the library
(``python -m muscle_synergies_amd.build --variant asan ...``) with the sanitizer runtime preloaded."""
import ctypes
import sys

from muscle_synergies_amd import _lib
from muscle_synergies_amd.engine import make_problem
from muscle_synergies_amd.preprocess import EnvelopeParams, SosfiltParams

lib = _lib.load()
assert lib.hipnmf_version() == 212
h = ctypes.c_void_p()
rc = lib.hipnmf_create(0, ctypes.byref(h))
have_gpu = rc == 0
assert rc in (0, _lib.HIPNMF_ERR_NO_DEVICE), (rc, lib.hipnmf_last_error())
assert lib.hipnmf_create(0, None) < 0
p = make_problem(4, 1000, 16, 5, x_layout=_lib.X_ROW_MAJOR, ldx=16, x_batch_stride=16000)
assert lib.hipnmf_workspace_bytes(ctypes.byref(p), 4) > 0
assert lib.hipnmf_workspace_bytes(None, 4) == 0
bad = make_problem(4, 1000, 16, 5, x_layout=_lib.X_ROW_MAJOR, ldx=16, x_batch_stride=16000)
bad.struct_size = 7
nul = [None] * 9
for sfx in ("f32", "f64"):
    for fn, nargs in ((f"hipnmf_fit_batched_{sfx}", 7), (f"hipnmf_shard_pass_{sfx}", 4), (f"hipnmf_shard_hupdate_{sfx}", 2),
                      (f"hipnmf_shard_residual_{sfx}", 5), (f"hipnmf_gram_{sfx}", 3), (f"hipnmf_nndsvd_stats_{sfx}", 4),
                      (f"hipnmf_fit_tsharded_{sfx}", 9)):
        f = getattr(lib, fn)
        f.restype = ctypes.c_int
        assert f(None, ctypes.byref(p), *nul[:nargs]) < 0, fn          # NULL handle
        assert len(lib.hipnmf_last_error()) > 0
        if have_gpu:
            assert f(h, ctypes.byref(bad), *nul[:nargs]) < 0, fn        # ABI guard
            assert f(h, ctypes.byref(p), *nul[:nargs]) < 0, fn          # NULL arrays
    f = getattr(lib, f"hipnmf_random_init_{sfx}")
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32] + [ctypes.c_void_p] * 3
    assert f(None, ctypes.byref(p), 1, 0, None, None, None) < 0
    f = getattr(lib, f"hipnmf_rank_sweep_{sfx}")
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_uint64, ctypes.c_int32] + \
                 [ctypes.c_void_p] * 7
    assert f(None, ctypes.byref(p), 2, 4, 0.9, 1, 0, *([None] * 7)) < 0
    if have_gpu:
        assert f(h, ctypes.byref(p), 4, 2, 0.9, 1, 0, *([None] * 7)) < 0
    f = getattr(lib, f"hipnmf_fit_ragged_{sfx}")
    f.restype = ctypes.c_int
    assert f(None, ctypes.byref(p), None, *nul[:7]) < 0
    e = EnvelopeParams(ctypes.sizeof(EnvelopeParams), 1, 100, 4, 0, 4, 400, 10, 1, 0, 1)
    f = getattr(lib, f"hipnmf_emg_envelope_{sfx}")
    f.restype = ctypes.c_int
    assert f(None, ctypes.byref(e), None, None) < 0
    f = getattr(lib, f"hipnmf_resample_weights_{sfx}")
    f.restype = ctypes.c_int
    assert f(None, ctypes.byref(e), None, None, None, 4, None) < 0
    s = SosfiltParams(ctypes.sizeof(SosfiltParams), 1, 100, 4, 0, 4, 400, 2, 1, -1, 0, 0, 0)
    f = getattr(lib, f"hipnmf_sosfilt_{sfx}")
    f.restype = ctypes.c_int
    assert f(None, ctypes.byref(s), None, None, None, None) < 0
for args in ((None, 0, 0, 0), (None, 300, 0, 0)):
    assert lib.hipnmf_set_tuning(*args) < 0
assert lib.hipnmf_last_kernel(None) == b""
ms = ctypes.c_float()
assert lib.hipnmf_last_kernel_ms(None, ctypes.byref(ms)) < 0
assert lib.hipnmf_set_stream(None, None) < 0 and lib.hipnmf_set_async(None, 1) < 0
assert lib.hipnmf_set_batch_hint(None, 4) < 0
if have_gpu:
    assert lib.hipnmf_set_tuning(h, 300, 0, 0) < 0 and lib.hipnmf_set_tuning(h, 512, 0, 9) < 0
    assert lib.hipnmf_set_tuning(h, 512, 0, 5) == 0
    assert lib.hipnmf_set_batch_hint(h, -1) < 0 and lib.hipnmf_set_batch_hint(h, 4096) == 0 and lib.hipnmf_set_batch_hint(h, 0) == 0
    assert lib.hipnmf_destroy(h) == 0
assert lib.hipnmf_destroy(None) <= 0
print("abi-host-drive: ok", "(gpu present)" if have_gpu else "(no gpu)")
sys.exit(0)
