"""RCCL straight through its C API (ctypes), for the collective callback of ``hipnmf_fit_tsharded_*``.

``include/hip_nmf.h`` (hipnmf_allreduce_fn) leaves the per-iteration all-reduce of the time-sharded loop to the host as a
callback ``int f(void* device_buf, size_t count, int elem_size, void* hip_stream, void* user)``.  A C / C++ / Go host
implements it with one ``ncclAllReduce`` on the stream it is handed; this module is that implementation, callable from
Python without ``torch.distributed`` in the path -- the library's kernels and RCCL's are then ordered by the library's own
stream, no host synchronisation inside an iteration.  (``torch.distributed`` with backend ``"nccl"`` is the other route:
``HipShardOps.fit_native`` wraps it into the same callback.)

The loop this serves replaces the reference's single-process call ``src/muscle_synergies/analysis.py:862-863`` for a recording
that does not fit one GPU (BASELINE.json config #5)."""
from __future__ import annotations

import ctypes
import os

from . import _lib

NCCL_UNIQUE_ID_BYTES = 128
NCCL_SUM = 0
NCCL_FLOAT32, NCCL_FLOAT64 = 7, 8


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


def _load_rccl():
    """The librccl the process already has (PyTorch-ROCm ships one next to libtorch), else ROCm's."""
    cands = []
    try:
        import torch

        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    except Exception:  # noqa: BLE001
        pass
    cands += ["/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"]
    err = None
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return ctypes.CDLL(c)
        except OSError as e:
            err = e
    raise _lib.HipNmfError(_lib.HIPNMF_ERR_UNSUPPORTED, f"librccl.so not found ({err})")


class RcclComm:
    """One rank's RCCL communicator.  ``unique_id``: the 128 bytes rank 0 got from :meth:`new_unique_id` and handed to the
    other ranks by any means (a file, MPI, a socket); a world of one needs none."""

    def __init__(self, rank: int = 0, world_size: int = 1, unique_id: bytes | None = None, device: int = 0):
        self.lib = _load_rccl()
        vp = ctypes.c_void_p
        self.lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        self.lib.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), ctypes.c_int, _UniqueId, ctypes.c_int]
        self.lib.ncclAllReduce.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp, vp]
        self.lib.ncclCommDestroy.argtypes = [vp]
        self.lib.ncclGetErrorString.restype = ctypes.c_char_p
        self.lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipSetDevice(int(device))
        uid = _UniqueId()
        if unique_id is None:
            if world_size != 1:
                raise ValueError("ranks of a world larger than one need rank 0's unique id")
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        else:
            if len(unique_id) != NCCL_UNIQUE_ID_BYTES:
                raise ValueError("unique_id must be %d bytes" % NCCL_UNIQUE_ID_BYTES)
            ctypes.memmove(ctypes.byref(uid), unique_id, NCCL_UNIQUE_ID_BYTES)
        self.comm = vp()
        self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), int(world_size), uid, int(rank)), "ncclCommInitRank")
        self.rank, self.world_size = int(rank), int(world_size)
        self.calls = 0
        self.elements = 0
        self._cb = None

    def new_unique_id(self) -> bytes:
        uid = _UniqueId()
        self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        return bytes(uid.internal)

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.ncclGetErrorString(rc)
            raise _lib.HipNmfError(_lib.HIPNMF_ERR_HIP, f"{what} failed: {msg.decode() if msg else rc}")

    def all_reduce_ptr(self, device_ptr: int, count: int, elem_size: int, hip_stream: int) -> int:
        """In-place SUM all-reduce of ``count`` floats (4-byte) or doubles (8-byte) at ``device_ptr`` on ``hip_stream``."""
        dt = NCCL_FLOAT32 if elem_size == 4 else NCCL_FLOAT64
        self.calls += 1
        self.elements += int(count)
        return self.lib.ncclAllReduce(ctypes.c_void_p(device_ptr), ctypes.c_void_p(device_ptr), int(count), dt, NCCL_SUM,
                                      self.comm, ctypes.c_void_p(hip_stream))

    def callback(self):
        """The ``hipnmf_allreduce_fn`` for ``hipnmf_fit_tsharded_*``: enqueues ``ncclAllReduce`` on the stream the library
        hands over (its own, or the one set with ``hipnmf_set_stream``) and returns without waiting."""
        if self._cb is None:
            def _cb(buf, count, elem_size, stream, user):
                try:
                    return 0 if self.all_reduce_ptr(buf or 0, count, elem_size, stream or 0) == 0 else 1
                except Exception:  # noqa: BLE001 -- must not propagate through the C frames
                    return 1

            self._cb = _lib.ALLREDUCE_FN(_cb)
        return self._cb

    def close(self):
        if getattr(self, "comm", None):
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
