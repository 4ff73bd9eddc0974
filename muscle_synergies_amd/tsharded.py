"""Time-sharded factorisation of one (or a few) very long recording(s) across GPUs.

BASELINE.json config #5: a single concatenated ``V`` (16 x 2e8) does not shard by matrix, it shards by
time: rank ``r`` holds rows ``[t0_r, t1_r)`` of ``X`` (sklearn orientation, ``T x m``) and of ``W``;
``H`` (``k x m``) is replicated.  One multiplicative-update iteration (``sklearn/decomposition/_nmf.py:831-862``)
then has exactly one exchange step:

    local   W_r <- W_r * (X_r H^T) / (W_r H H^T)                     (row-local, no communication)
    local   sums_r = [ W_r^T X_r (k x m) | W_r^T W_r (k x k) ]       (the T-long reductions, per shard)
    global  sums   = all_reduce(sum, sums_r)                         (k*m + k*k floats: RCCL over xGMI)
    local   H <- H * (W^T X) / ((W^T W) H)                           (replicated, identical on all ranks)

and, when the stop rule is live (``tol > 0``, every 10th iteration, ``_nmf.py:872-884``), one more
all-reduce of the ``m`` per-column squared errors.  The payload is 105 floats for m=16, k=5: the step is
latency-bound, so a single packed all-reduce per iteration is used and ring/tree choice is irrelevant.

The per-shard compute is delegated to a *shard-ops* object.  The product implementation is
:class:`HipShardOps` (the ``hipnmf_shard_*`` entry points of ``libhip_nmf.so``); the world_size-2
``gloo`` test injects an oracle-backed ops object to check this orchestration without a GPU.
"""

from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import Callable, Optional

from . import _lib
from .engine import make_problem, partition, resolve_device


@dataclass
class ShardedResult:
    W_local: "object"  # this rank's rows of W, [B, T_local, k]
    H: "object"  # replicated components, [B, k, m]
    n_iter: int
    reconstruction_err: "object"  # [B] global ||X - W H||_F
    vaf: "object"  # [B, 1 + m]


def shard_bounds(n_samples: int, world_size: int, align: int = 4):
    """Contiguous row ranges, one per rank, each a multiple of ``align`` rows except the last."""
    blocks = partition((n_samples + align - 1) // align, world_size)
    return [(min(lo * align, n_samples), min(hi * align, n_samples)) for lo, hi in blocks]


def plan_subshards(n_samples: int, world_size: int, rank: int, subshard: int):
    """Sub-shards of rank ``rank`` for one recording of ``n_samples`` rows time-sharded over ``world_size`` ranks
    (``bench.py --config 5``): the rank's rows (:func:`shard_bounds`) cut into pieces of at most ``subshard`` rows
    (the engine addresses < 2 GiB of X per shard).  Returns ``[(first_row, n_rows, shard_id), ...]``; ``shard_id``
    names the piece of the synthetic recording to generate: the global sub-shard index when the piece starts on a
    sub-shard boundary -- with ``n_samples / world_size`` a multiple of ``subshard`` (N = 1, 2, 4, 8 at the defaults)
    every N therefore generates the very same recording -- and a per-rank id otherwise."""
    lo, hi = shard_bounds(n_samples, world_size)[rank]
    out, t = [], lo
    while t < hi:
        n = min(subshard, hi - t)
        sid = t // subshard if t % subshard == 0 else 10_000 + 100 * rank + len(out)
        out.append((t, n, sid))
        t += n
    return out


def _is_wide(m: int, k: int, kl: bool = False) -> bool:
    """Beyond the narrow lane mappings -- and for the Kullback-Leibler loss whatever the shape -- the shard entry points run on
    the general-shape kernels, which take row-major X and W."""
    return m > 32 or k > 8 or kl


class HipShardOps:
    """Shard-local compute on one GPU through the C ABI.  Native layouts: up to 32 channels and 8 components X ``[B, m, T_local]``
    channel-major and W ``[B, k, T_local]`` component-major; beyond (up to 512 x 64, round 4) X ``[B, T_local, ldx]`` row-major
    with 16-byte aligned rows and W ``[B, T_local, KP]`` row-major, ``KP = round_up(k, 16)``, padding columns zero."""

    def __init__(self, X_local, W_local, H, *, l1_reg_W=0.0, l1_reg_H=0.0, l2_reg_W=0.0, l2_reg_H=0.0,
                 update_H=True, device=None, beta_loss="frobenius"):
        import torch

        from .engine import beta_loss_code

        self.torch = torch
        self.kl = beta_loss_code(beta_loss) == _lib.LOSS_KL  # residual() then returns the divergence per column
        self.dev = resolve_device(device)
        X = torch.as_tensor(X_local).to(self.dev)
        if X.dim() == 2:
            X, W_local, H = X.unsqueeze(0), torch.as_tensor(W_local).unsqueeze(0), torch.as_tensor(H).unsqueeze(0)
        self.B, self.T, self.m = X.shape
        self.dtype = X.dtype
        W = torch.as_tensor(W_local).to(self.dev, self.dtype)
        self.k = W.shape[2]
        self.wide = _is_wide(self.m, self.k, self.kl)
        if self.wide:
            vec = 4 if self.dtype == torch.float32 else 2
            self.ld = self.T
            self.ldx = (self.m + vec - 1) // vec * vec
            self.kp = (self.k + 15) // 16 * 16
            self.Xc = torch.zeros((self.B, self.T, self.ldx), dtype=self.dtype, device=self.dev)
            self.Xc[:, :, : self.m] = X
            self.Wc = torch.zeros((self.B, self.T, self.kp), dtype=self.dtype, device=self.dev)
            self.Wc[:, :, : self.k] = W
        else:
            # native layouts; the shard is zero-padded to a multiple of 4 rows (zero rows of X with zero rows of
            # W stay zero under the update and contribute nothing to any sum)
            self.ld = (self.T + 3) // 4 * 4
            self.Xc = torch.zeros((self.B, self.m, self.ld), dtype=self.dtype, device=self.dev)
            self.Xc[:, :, : self.T] = X.transpose(1, 2)
            self.Wc = torch.zeros((self.B, self.k, self.ld), dtype=self.dtype, device=self.dev)
            self.Wc[:, :, : self.T] = W.transpose(1, 2)
        self.H = torch.as_tensor(H).to(self.dev, self.dtype).contiguous().clone()
        self.sums = torch.empty((self.B, self.k * self.m + self.k * self.k), dtype=self.dtype, device=self.dev)
        self.sse = torch.empty((self.B, self.m), dtype=self.dtype, device=self.dev)
        self.xsq = torch.empty((self.B, self.m), dtype=self.dtype, device=self.dev)
        # a private handle bound to torch's current stream, asynchronous: kernels, torch ops and the RCCL
        # all-reduce are then ordered by that one stream, with no host synchronisation inside an iteration
        self.handle = _lib.Handle(self.dev.index)
        self.handle.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.handle.set_async(True)
        if self.wide:
            self.p = make_problem(self.B, self.T, self.m, self.k, x_layout=_lib.X_ROW_MAJOR, ldx=self.ldx,
                                  x_batch_stride=self.T * self.ldx, w_layout=_lib.W_ROW_MAJOR_PAD16, update_H=update_H,
                                  max_iter=1, tol=0.0, l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H, l2_reg_W=l2_reg_W,
                                  l2_reg_H=l2_reg_H, loss=_lib.LOSS_KL if self.kl else _lib.LOSS_FROBENIUS)
        else:
            self.p = make_problem(self.B, self.ld, self.m, self.k, x_layout=_lib.X_CHANNEL_MAJOR, ldx=self.ld,
                                  x_batch_stride=self.m * self.ld, w_layout=_lib.W_COMPONENT_MAJOR, update_H=update_H,
                                  max_iter=1, tol=0.0, l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H, l2_reg_W=l2_reg_W,
                                  l2_reg_H=l2_reg_H)
        sfx = "f32" if self.dtype == torch.float32 else "f64"
        lib = _lib.load()
        self._pass = getattr(lib, f"hipnmf_shard_pass_{sfx}")
        self._hupd = getattr(lib, f"hipnmf_shard_hupdate_{sfx}")
        self._res = getattr(lib, f"hipnmf_shard_residual_{sfx}")
        torch.cuda.synchronize(self.dev)

    @classmethod
    def from_native(cls, Xc, Wc, H, *, T=None, l1_reg_W=0.0, l1_reg_H=0.0, l2_reg_W=0.0, l2_reg_H=0.0,
                    update_H=True):
        """Wrap device tensors that already are in the engine's layouts -- ``Xc [B, m, ld]`` channel-major,
        ``Wc [B, k, ld]`` component-major (updated in place), ``H [B, k, m]`` (updated in place; several shards of
        one rank may share the same tensor) -- without the padded copies the ordinary constructor makes.
        ``ld % 4 == 0``; rows ``>= T`` (default ``ld``) must be zero in both X and W."""
        import torch

        self = cls.__new__(cls)
        self.torch = torch
        self.wide = False  # (the wide layouts go through the ordinary constructor)
        self.kl = False
        self.dev = Xc.device
        if not (Xc.is_cuda and Wc.is_cuda and H.is_cuda and Xc.is_contiguous() and Wc.is_contiguous()
                and H.is_contiguous()):
            raise ValueError("from_native needs contiguous device tensors")
        self.B, self.m, self.ld = Xc.shape
        if self.ld % 4 or Wc.shape[0] != self.B or Wc.shape[2] != self.ld or H.shape != (self.B, Wc.shape[1], self.m):
            raise ValueError("from_native: need Xc [B, m, ld], Wc [B, k, ld], H [B, k, m] with ld % 4 == 0")
        self.T = self.ld if T is None else int(T)
        self.dtype = Xc.dtype
        self.k = Wc.shape[1]
        self.Xc, self.Wc, self.H = Xc, Wc, H
        self.sums = torch.empty((self.B, self.k * self.m + self.k * self.k), dtype=self.dtype, device=self.dev)
        self.sse = torch.empty((self.B, self.m), dtype=self.dtype, device=self.dev)
        self.xsq = torch.empty((self.B, self.m), dtype=self.dtype, device=self.dev)
        self.handle = _lib.Handle(self.dev.index)
        self.handle.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.handle.set_async(True)
        self.p = make_problem(self.B, self.ld, self.m, self.k, x_layout=_lib.X_CHANNEL_MAJOR, ldx=self.ld,
                              x_batch_stride=self.m * self.ld, w_layout=_lib.W_COMPONENT_MAJOR, update_H=update_H,
                              max_iter=1, tol=0.0, l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H, l2_reg_W=l2_reg_W,
                              l2_reg_H=l2_reg_H)
        sfx = "f32" if self.dtype == torch.float32 else "f64"
        lib = _lib.load()
        self._pass = getattr(lib, f"hipnmf_shard_pass_{sfx}")
        self._hupd = getattr(lib, f"hipnmf_shard_hupdate_{sfx}")
        self._res = getattr(lib, f"hipnmf_shard_residual_{sfx}")
        return self

    def fit_native(self, *, max_iter: int = 200, tol: float = 1e-4, check_every: int = 10, group=None,
                   all_reduce: Optional[Callable] = None, collective_fn=None) -> "ShardedResult":
        """The whole sharded fit inside the library (``hipnmf_fit_tsharded_*``): same loop, same kernels and the same
        collective as :func:`fit_tsharded`, but driven from C++ -- the per-iteration Python overhead (three ctypes
        calls, a tensor clone) is gone and the entry point is usable from any host language.  ``all_reduce(tensor)``
        must sum ``tensor`` in place over all ranks; the default is ``torch.distributed.all_reduce`` over ``group``
        when a process group is initialised, nothing otherwise.  Collective: every rank calls it.
        ``collective_fn``: a ready-made C callback (``_lib.ALLREDUCE_FN``, e.g. ``rccl.RcclComm.callback()``: ``ncclAllReduce``
        on the stream the library hands over) used instead of ``all_reduce`` -- no Python tensor, no torch.distributed."""
        torch = self.torch
        if all_reduce is None and collective_fn is None:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized():
                def all_reduce(t):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        dev, dtype = self.dev, self.dtype
        failure = []

        class _DeviceBuffer:  # zero-copy view of the library's buffer for torch (CUDA array interface, version 2)
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4" if dtype == torch.float32 else "<f8",
                                                 "data": (int(ptr), False), "version": 2}

        def _cb(buf, count, elem_size, stream, user):
            try:  # the handle runs on torch's current stream (see __init__): the collective is ordered behind the kernels
                all_reduce(torch.as_tensor(_DeviceBuffer(buf, count), device=dev))
                return 0
            except Exception as e:  # noqa: BLE001 -- must not propagate through the C frames
                failure.append(e)
                return 1

        if collective_fn is not None:
            cb = collective_fn
        else:
            cb = _lib.ALLREDUCE_FN(_cb) if all_reduce is not None else _lib.ALLREDUCE_FN()  # NULL: single rank
        p = _lib.Problem.from_buffer_copy(self.p)
        p.max_iter, p.tol, p.check_every = int(max_iter), float(tol), int(check_every)
        err = torch.empty((self.B,), dtype=dtype, device=dev)
        n_iter = torch.empty((self.B,), dtype=torch.int32, device=dev)
        sse = torch.empty((self.B, self.m), dtype=dtype, device=dev)
        xsq = torch.empty((self.B, self.m), dtype=dtype, device=dev)
        fn = getattr(_lib.load(), "hipnmf_fit_tsharded_f32" if dtype == torch.float32 else "hipnmf_fit_tsharded_f64")
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _lib.ALLREDUCE_FN,
                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        fn.restype = ctypes.c_int
        rc = fn(self.handle.ptr, ctypes.addressof(p), self.Xc.data_ptr(), self.Wc.data_ptr(), self.H.data_ptr(), cb, None,
                err.data_ptr(), n_iter.data_ptr(), sse.data_ptr(), xsq.data_ptr())
        if failure:
            raise failure[0]
        _lib.check(rc)
        vaf = torch.cat([(1 - sse.sum(dim=1) / xsq.sum(dim=1)).unsqueeze(1), 1 - sse / xsq], dim=1)
        return ShardedResult(self.result_W(), self.result_H(), int(n_iter[0]), err, vaf)

    def shard_pass(self):
        _lib.check(self._pass(self.handle.ptr, ctypes.byref(self.p), self.Xc.data_ptr(), self.Wc.data_ptr(),
                              self.H.data_ptr(), self.sums.data_ptr()))
        return self.sums

    def h_update(self, sums):
        _lib.check(self._hupd(self.handle.ptr, ctypes.byref(self.p), self.H.data_ptr(), sums.data_ptr()))

    def residual(self):
        """Per-column sums of this shard: ``(sse, xsq)``; with the Kullback-Leibler loss the first is the generalised
        divergence per column (``reconstruction_err = sqrt(2 * sum)``), :meth:`residual_squared` the squared error."""
        _lib.check(self._res(self.handle.ptr, ctypes.byref(self.p), self.Xc.data_ptr(), self.Wc.data_ptr(),
                             self.H.data_ptr(), self.sse.data_ptr(), self.xsq.data_ptr()))
        return self.sse, self.xsq

    def residual_squared(self):
        """``(sse, xsq)`` of the squared error whatever the loss (the columns VAF is made of)."""
        if not self.kl:
            return self.residual()
        q = _lib.Problem.from_buffer_copy(self.p)
        q.loss = _lib.LOSS_FROBENIUS
        _lib.check(self._res(self.handle.ptr, ctypes.byref(q), self.Xc.data_ptr(), self.Wc.data_ptr(),
                             self.H.data_ptr(), self.sse.data_ptr(), self.xsq.data_ptr()))
        return self.sse, self.xsq

    def result_W(self):
        if self.wide:
            return self.Wc[:, :, : self.k].contiguous()
        return self.Wc[:, :, : self.T].transpose(1, 2).contiguous()  # [B, T_local, k]

    def result_H(self):
        return self.H


class MultiShardOps:
    """Several consecutive time shards held by ONE rank, presented to :func:`fit_tsharded` as one shard.

    The engine addresses a matrix through 32-bit buffer resources (< 2 GiB of X per shard: 3.3e7 rows at 16
    channels), so a rank that owns more rows than that -- BASELINE config #5 on fewer than 8 GPUs -- keeps them as
    sub-shards: every pass runs over each of them with the same replicated ``H`` (the sub-shards share one ``H``
    tensor), their ``[W^T X | W^T W]`` sums are added on the device in shard order, and the (all-reduced) result
    updates ``H`` once."""

    def __init__(self, shards):
        if not shards:
            raise ValueError("need at least one shard")
        self.shards = list(shards)
        h0 = self.shards[0].H
        for sh in self.shards[1:]:
            if sh.H.data_ptr() != h0.data_ptr():
                raise ValueError("the sub-shards of a rank must share one H tensor (HipShardOps.from_native)")

    def shard_pass(self):
        total = self.shards[0].shard_pass().clone()
        for sh in self.shards[1:]:
            total += sh.shard_pass()
        return total

    def h_update(self, sums):
        self.shards[0].h_update(sums)  # H is shared

    def residual(self):
        sse, xsq = self.shards[0].residual()
        sse, xsq = sse.clone(), xsq.clone()
        for sh in self.shards[1:]:
            a, b = sh.residual()
            sse += a
            xsq += b
        return sse, xsq

    @property
    def kl(self):
        return bool(getattr(self.shards[0], "kl", False))

    def residual_squared(self):
        sse, xsq = self.shards[0].residual_squared()
        sse, xsq = sse.clone(), xsq.clone()
        for sh in self.shards[1:]:
            a, b = sh.residual_squared()
            sse += a
            xsq += b
        return sse, xsq

    def result_W(self):
        return [sh.Wc for sh in self.shards]  # native layout, no concatenation (may be tens of GB)

    def result_H(self):
        return self.shards[0].H


def fit_tsharded(ops, *, max_iter: int = 200, tol: float = 1e-4, check_every: int = 10, update_H: bool = True,
                 group=None, all_reduce: Optional[Callable] = None) -> ShardedResult:
    """Run the sharded solver on this rank.  ``ops`` provides ``shard_pass() -> sums``,
    ``h_update(sums)``, ``residual() -> (sse_col, xsq_col)``, ``result_W()``, ``result_H()``; every rank
    must call this function (collective).  ``all_reduce(tensor)`` defaults to a SUM all-reduce over
    ``group`` with ``torch.distributed`` (backend ``nccl`` = RCCL on ROCm; ``gloo`` in the CPU test)."""
    import torch

    if all_reduce is None:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            def all_reduce(t):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                return t
        else:
            def all_reduce(t):  # single rank
                return t

    kl = bool(getattr(ops, "kl", False))  # the residual's first output is then the divergence per column (_nmf.py:185-189)

    def global_error(squared=False):
        sse, xsq = ops.residual_squared() if squared else ops.residual()
        packed = torch.cat([sse, xsq], dim=1).clone()
        packed = all_reduce(packed)
        m = sse.shape[1]
        return packed[:, :m], packed[:, m:]

    def to_err(cols):
        tot = cols.sum(dim=1)
        return torch.sqrt(2 * torch.clamp(tot, min=0)) if kl else torch.sqrt(tot)

    err_init = prev = None
    if tol > 0:
        sse, _ = global_error()
        err_init = to_err(sse)
        prev = err_init.clone()
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        sums = ops.shard_pass()
        if update_H:
            sums = all_reduce(sums)
            ops.h_update(sums)
        if tol > 0 and n_iter % check_every == 0:
            sse, _ = global_error()
            err = to_err(sse)
            # every matrix of the (small) batch must have converged; with B == 1 this is sklearn's rule
            if bool((((prev - err) / err_init) < tol).all()):
                break
            prev = err
    sse, xsq = global_error()
    err = to_err(sse)
    if kl:  # VAF is made of the squared-error columns: one more residual pass
        sse, xsq = global_error(squared=True)
    vaf = torch.cat([(1 - sse.sum(dim=1) / xsq.sum(dim=1)).unsqueeze(1), 1 - sse / xsq], dim=1)
    return ShardedResult(ops.result_W(), ops.result_H(), n_iter, err, vaf)


def fit_tsharded_hip(X_local, W_local, H, *, max_iter=200, tol=1e-4, check_every=10, update_H=True, group=None,
                     device=None, **reg) -> ShardedResult:
    """Convenience wrapper: build :class:`HipShardOps` for this rank's rows and run :func:`fit_tsharded`
    (``beta_loss='kullback-leibler'`` among the keyword arguments selects that loss)."""
    ops = HipShardOps(X_local, W_local, H, update_H=update_H, device=device, **reg)
    return fit_tsharded(ops, max_iter=max_iter, tol=tol, check_every=check_every, update_H=update_H, group=group)


# ------------------------------------------------------------------------------------------------------------------------
# Config #5 in ONE process: one host thread + handle per device, the per-iteration sums added through host memory.
class HostStagedAllReduce:
    """SUM all-reduce between the shard threads of one process (``fit_tsharded_devices``): every participant copies its
    ``k*m + k*k`` (or ``2m``) values into its slot of one pinned host buffer, a barrier, every participant adds ALL slots in slot
    order -- so each gets bitwise the same sum, whatever the arrival order -- and takes the result back to its device; a second
    barrier frees the slots.  Two tiny copies and two thread barriers per iteration (tens of microseconds) against
    milliseconds of shard pass: the same exchange step RCCL performs between processes (``fit_tsharded``), for the caller
    who has one process -- a notebook -- and several GPUs.  Works with the same device named twice."""

    def __init__(self, n: int, pinned: bool = True):
        import threading

        self.n = int(n)
        self.pinned = pinned
        self.barrier = threading.Barrier(self.n)
        self.slots = None
        self.calls = 0
        self.elements = 0
        self._lock = threading.Lock()

    def _ensure(self, t):
        import torch

        numel = t.numel()
        with self._lock:
            if self.slots is None or self.slots.shape[1] < numel or self.slots.dtype != t.dtype:
                buf = torch.empty((self.n, max(numel, 256)), dtype=t.dtype)
                if self.pinned and t.is_cuda:
                    buf = buf.pin_memory()
                self.slots = buf
        return self.slots

    def reducer(self, index: int):
        """The ``all_reduce(tensor) -> tensor`` callable of participant ``index`` (in place on ``tensor``)."""
        import torch

        def all_reduce(t):
            numel = t.numel()
            if index == 0:
                self._ensure(t)
                self.calls += 1
                self.elements += numel
            self.barrier.wait()  # slots allocated; the previous round's readers are done
            slot = self.slots[index, :numel]
            slot.copy_(t.reshape(-1), non_blocking=False)  # device -> pinned host, waited for (orders behind the shard pass)
            self.barrier.wait()  # all slots written
            total = self.slots[0, :numel].clone()
            for j in range(1, self.n):  # fixed order: every participant computes the very same bits
                total += self.slots[j, :numel]
            t.copy_(total.reshape(t.shape).to(t.device, non_blocking=False) if t.is_cuda else total.reshape(t.shape))
            return t

        return all_reduce

    def abort(self):
        self.barrier.abort()


def fit_tsharded_devices(X, W0, H0, *, devices, max_iter: int = 200, tol: float = 1e-4, check_every: int = 10,
                         update_H: bool = True, native: bool = False, subshard: int = 25_000_000, beta_loss="frobenius",
                         l1_reg_W: float = 0.0, l1_reg_H: float = 0.0, l2_reg_W: float = 0.0, l2_reg_H: float = 0.0,
                         _ops_factory=None) -> ShardedResult:
    """BASELINE.json config #5 from ONE process: factorise one long recording ``X [T, m]`` (host memory: NumPy or a CPU
    tensor -- what the reference holds, ``analysis.py:862-863``) with its rows sharded over ``devices`` -- device indices,
    ``"cuda:i"`` strings or ``"all"``; the same device may be named twice, which is how the exchange is exercised per slice
    on a one-GPU box.  One host thread, one engine handle and one copy of ``H`` per entry; per iteration each thread runs its
    shard pass and the ``k*m + k*k`` sums are added through pinned host memory (:class:`HostStagedAllReduce`), then every
    replica of ``H`` takes the same update.  ``native=True`` drives each shard's loop inside the library
    (``hipnmf_fit_tsharded_*`` with the host-staged sum as its collective callback).  A device's rows beyond ``subshard``
    (< 2 GiB of X per shard) are kept as sub-shards sharing one ``H`` (:class:`MultiShardOps`; narrow shapes).

    Returns a :class:`ShardedResult` whose ``W_local`` is the WHOLE ``W [1, T, k]`` gathered on the host in row order
    (NumPy when NumPy went in), ``H [1, k, m]``, ``reconstruction_err [1]``, ``vaf [1, 1 + m]`` on the host."""
    import threading

    import numpy as np
    import torch

    from .multi_gpu import resolve_devices

    as_numpy = isinstance(X, np.ndarray)
    Xt, Wt, Ht = torch.as_tensor(X), torch.as_tensor(W0), torch.as_tensor(H0)
    if Xt.dim() != 2 or Wt.dim() != 2 or Ht.dim() != 2:
        raise ValueError("fit_tsharded_devices takes ONE recording: X [T, m], W0 [T, k], H0 [k, m]")
    T, m = Xt.shape
    k = Ht.shape[0]
    if Wt.shape != (T, k) or Ht.shape != (k, m):
        raise ValueError(f"shapes do not agree: X {tuple(Xt.shape)}, W0 {tuple(Wt.shape)}, H0 {tuple(Ht.shape)}")
    devs = resolve_devices(devices) if _ops_factory is None else list(devices)
    if not devs:
        raise ValueError("devices must name at least one GPU")
    bounds = [b for b in shard_bounds(T, len(devs)) if b[1] > b[0]]
    devs = devs[: len(bounds)]
    n = len(bounds)
    red = HostStagedAllReduce(n, pinned=_ops_factory is None)
    reg = dict(l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H, l2_reg_W=l2_reg_W, l2_reg_H=l2_reg_H)
    results: list = [None] * n
    errors: list = [None] * n

    def make_ops(i, lo, hi, dev):
        if _ops_factory is not None:
            return _ops_factory(Xt[lo:hi], Wt[lo:hi], Ht, i)
        rows = hi - lo
        wide = _is_wide(m, k, beta_loss != "frobenius")
        if rows <= subshard or wide:
            return HipShardOps(Xt[lo:hi], Wt[lo:hi], Ht, update_H=update_H, device=torch.device("cuda", dev), beta_loss=beta_loss, **reg)
        d = torch.device("cuda", dev)
        Hd = Ht.to(d).unsqueeze(0).contiguous().clone()
        shards, t = [], lo
        while t < hi:
            nr = min(subshard, hi - t)
            ld = (nr + 3) // 4 * 4
            Xc = torch.zeros((1, m, ld), dtype=Xt.dtype, device=d)
            Xc[0, :, :nr] = Xt[t:t + nr].to(d).t()
            Wc = torch.zeros((1, k, ld), dtype=Xt.dtype, device=d)
            Wc[0, :, :nr] = Wt[t:t + nr].to(d, Xt.dtype).t()
            sh = HipShardOps.from_native(Xc, Wc, Hd, T=nr, update_H=update_H, **reg)
            sh._rows = nr
            shards.append(sh)
            t += nr
        return MultiShardOps(shards)

    def gather_w(ops, r):
        if isinstance(ops, MultiShardOps):  # native sub-shards: [1, k, ld] component-major each
            return torch.cat([sh.Wc[0, :, : sh._rows].t().cpu() for sh in ops.shards], dim=0).unsqueeze(0)
        w = r.W_local
        return w.detach().cpu() if hasattr(w, "detach") else torch.as_tensor(w)

    def run(i):
        lo, hi = bounds[i]
        ops = None
        try:
            import contextlib

            side = contextlib.nullcontext()
            if _ops_factory is None:
                torch.cuda.set_device(devs[i])
                # a stream of this thread's own: torch ops and the library's kernels (HipShardOps binds its handle to the
                # thread's current stream) are ordered on it, and two shards that share a device do not serialise on its default stream
                side = torch.cuda.stream(torch.cuda.Stream(torch.device("cuda", devs[i])))
            with side:
                ops = make_ops(i, lo, hi, devs[i])
                ar = red.reducer(i)
                if native and hasattr(ops, "fit_native"):
                    r = ops.fit_native(max_iter=max_iter, tol=tol, check_every=check_every, all_reduce=ar)
                else:
                    r = fit_tsharded(ops, max_iter=max_iter, tol=tol, check_every=check_every, update_H=update_H, all_reduce=ar)
                host = lambda t: t.detach().cpu() if hasattr(t, "detach") else torch.as_tensor(t)  # noqa: E731
                results[i] = (gather_w(ops, r), host(r.H), int(r.n_iter), host(r.reconstruction_err), host(r.vaf))
        except BaseException as e:  # noqa: BLE001 -- re-raised by the caller's thread
            errors[i] = e
            red.abort()  # the other participants leave their barrier with BrokenBarrierError instead of waiting for ever
        finally:
            for sh in (getattr(ops, "shards", None) or ([ops] if ops is not None else [])):
                h = getattr(sh, "handle", None)
                if h is not None:
                    h.close()

    threads = [threading.Thread(target=run, args=(i,), name=f"hipnmf-tshard-{i}") for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    first = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)] or [e for e in errors if e is not None]
    if first:
        raise first[0]
    W = torch.cat([r[0].reshape(1, -1, k) for r in results], dim=1)
    H, n_iter, err, vaf = results[0][1], results[0][2], results[0][3], results[0][4]
    for r in results[1:]:  # the replicas took the same sums in the same order: they must agree to the bit
        if not torch.equal(r[1], H) or r[2] != n_iter:
            raise RuntimeError("fit_tsharded_devices: the replicas of H diverged between shard threads")
    out = ShardedResult(W, H.reshape(1, k, m), n_iter, err.reshape(-1), vaf.reshape(1, -1))
    out.collective = {"backend": "host-staged sum between shard threads of one process", "participants": n,
                      "all_reduce_calls": red.calls, "elements": red.elements}
    if as_numpy:
        out.W_local, out.H, out.reconstruction_err, out.vaf = (out.W_local.numpy(), out.H.numpy(), out.reconstruction_err.numpy(),
                                                               out.vaf.numpy())
    return out
