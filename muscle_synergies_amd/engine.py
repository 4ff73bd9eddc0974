"""Batched host API of the HIP NMF engine.

``fit_batched`` factorises ``B`` independent non-negative matrices ``X[b] (T x m) ~ W[b] (T x k) H[b] (k x m)``
with the Lee-Seung multiplicative updates exactly as ``sklearn.decomposition.NMF(solver='mu',
beta_loss='frobenius', init='custom')`` does for one matrix (``sklearn/decomposition/_nmf.py:731-893``),
the arithmetic the reference reaches from ``src/muscle_synergies/analysis.py:862-863``.

PyTorch-ROCm is used only as allocator / host-device copier: tensors' ``data_ptr()`` are handed to the
C ABI of ``libhip_nmf.so``.  No computation of the solver happens in torch or NumPy.
"""

from __future__ import annotations

import ctypes
import threading
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib


@dataclass
class BatchedResult:
    """Outputs of :func:`fit_batched` (tensors on the compute device unless NumPy went in)."""

    W: "object"  # [B, T, k]   transformed signal (``fit_transform`` output)
    H: "object"  # [B, k, m]   ``components_``
    n_iter: "object"  # [B] int32   ``n_iter_``
    reconstruction_err: "object"  # [B]  ``reconstruction_err_`` = ||X - W H||_F
    vaf: "object"  # [B, 1 + m]  "All signals" then one per column (analysis.py:661-667)
    sse_col: "object"  # [B, m]  per-column sum((X - W H)^2)
    xsq_col: "object"  # [B, m]  per-column sum(X^2)
    kernel_ms: float  # device time of the solver kernels (HIP events)


def _torch():
    import torch

    return torch


def resolve_device(device=None):
    """``torch.device`` of the GPU to use; fails loudly when there is none."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise _lib.HipNmfError(
            _lib.HIPNMF_ERR_NO_DEVICE,
            "no ROCm GPU visible to PyTorch; the HIP NMF engine has no CPU fallback",
        )
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(device)
    if dev.type != "cuda":
        raise ValueError(f"device must be a GPU (got {dev})")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _as_device_tensor(a, dev, dtype=None):
    """NumPy / torch input -> tensor on ``dev`` with strides preserved (dense inputs)."""
    torch = _torch()
    if isinstance(a, torch.Tensor):
        t = a
    else:
        arr = np.asarray(a)
        if any(s < 0 for s in arr.strides):
            arr = np.ascontiguousarray(arr)
        t = torch.from_numpy(arr)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.device != dev:
        t = t.to(dev)
    return t


def _x_layout(Xt):
    """(x_layout, ldx, batch_stride, tensor) for a [B, T, m] tensor; copies only when unavoidable."""
    B, T, m = Xt.shape
    sb, st, sm = Xt.stride()
    if B == 1:
        sb = T * m if sb == 0 else sb
    if sm == 1 and st >= m and (B == 1 or sb >= 1):
        return _lib.X_ROW_MAJOR, st, sb, Xt
    if st == 1 and sm >= T and (B == 1 or sb >= 1):
        return _lib.X_CHANNEL_MAJOR, sm, sb, Xt
    Xc = Xt.contiguous()
    return _lib.X_ROW_MAJOR, m, T * m, Xc


def beta_loss_code(beta_loss) -> int:
    """sklearn's ``beta_loss`` spelling (``_nmf.py:86-92``) -> ``HIPNMF_LOSS_*``; anything but Frobenius (2)
    and Kullback-Leibler (1) is outside this engine."""
    if beta_loss in ("frobenius", 2, 2.0):
        return _lib.LOSS_FROBENIUS
    if beta_loss in ("kullback-leibler", 1, 1.0):
        return _lib.LOSS_KL
    raise NotImplementedError(f"beta_loss={beta_loss!r}: the HIP engine implements 'frobenius' and 'kullback-leibler'")


def make_problem(B, T, m, k, *, x_layout, ldx, x_batch_stride, w_layout=_lib.W_ROW_MAJOR, update_H=True,
                 max_iter=200, tol=1e-4, check_every=10, l1_reg_W=0.0, l1_reg_H=0.0, l2_reg_W=0.0,
                 l2_reg_H=0.0, loss=_lib.LOSS_FROBENIUS) -> _lib.Problem:
    p = _lib.Problem()
    p.struct_size = ctypes.sizeof(_lib.Problem)
    p.batch, p.n_samples, p.n_features, p.n_components = int(B), int(T), int(m), int(k)
    p.x_layout, p.update_h, p.w_layout, p.loss = int(x_layout), int(bool(update_H)), int(w_layout), int(loss)
    p.ldx, p.x_batch_stride = int(ldx), int(x_batch_stride)
    p.max_iter, p.check_every, p.tol = int(max_iter), int(check_every), float(tol)
    p.l1_reg_W, p.l1_reg_H = float(l1_reg_W), float(l1_reg_H)
    p.l2_reg_W, p.l2_reg_H = float(l2_reg_W), float(l2_reg_H)
    return p


def fit_batched(X, W0, H0, *, max_iter: int = 200, tol: float = 1e-4, check_every: int = 10,
                update_H: bool = True, l1_reg_W: float = 0.0, l1_reg_H: float = 0.0, l2_reg_W: float = 0.0,
                l2_reg_H: float = 0.0, beta_loss="frobenius", device=None, handle: Optional[_lib.Handle] = None,
                return_numpy: Optional[bool] = None, overwrite_init: bool = False, devices=None,
                host_chunk: Optional[int] = None, _inputs_ready: bool = False) -> BatchedResult:
    """Factorise a batch of matrices on one GPU -- or, with ``devices=``, scattered by matrix over several
    (:mod:`muscle_synergies_amd.multi_gpu`: contiguous slices, one host thread and handle per device, no collective;
    results on the host in batch order: NumPy when NumPy went in, CPU tensors otherwise).

    Args:
        X: ``[B, T, m]`` (or ``[T, m]``) non-negative float32/float64, NumPy or torch, any dense layout
           (C order = row-major; a transposed view of a ``[B, m, T]`` array = channel-major, the
           layout ``DataFrame.to_numpy()`` produces).  Whichever order the kernel for this shape streams (see
           ``include/hip_nmf.h``) is used in place; the other one costs a single conversion per fit.
        W0: ``[B, T, k]`` initial activations; H0: ``[B, k, m]`` initial synergies (``init='custom'``).
        max_iter, tol: as ``sklearn.decomposition.NMF``; ``tol=0`` runs exactly ``max_iter`` updates.
        update_H: ``False`` keeps H fixed (``NMF.transform``).
        beta_loss: ``'frobenius'`` (the reference's default) or ``'kullback-leibler'``
           (``reconstruction_err`` is then ``sqrt(2 KL(X || WH))``; ``vaf`` stays the squared-error VAF).
        overwrite_init: let the solver update contiguous device tensors ``W0``/``H0`` in place (no copy).
        host_chunk: host-resident (NumPy) batches are fitted in chunks of this many matrices with the transfers of the
            neighbouring chunks overlapped (:func:`_fit_batched_pipelined`); ``None`` = automatic (batches of more than
            :data:`PIPELINE_MIN_BYTES`), ``0`` = one upload, one fit, one download.
    """
    torch = _torch()
    if devices is not None:
        if handle is not None or overwrite_init or device is not None:
            raise ValueError("devices= scatters the batch over one thread, handle and private copy per device: it cannot be "
                             "combined with handle=, overwrite_init=True or device=")
        kw = dict(max_iter=max_iter, tol=tol, check_every=check_every, update_H=update_H, l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H,
                  l2_reg_W=l2_reg_W, l2_reg_H=l2_reg_H, beta_loss=beta_loss, host_chunk=host_chunk)
        return _fit_batched_scattered(X, W0, H0, devices, return_numpy, kw)
    dev = resolve_device(device)
    was_numpy = not isinstance(X, torch.Tensor)
    if return_numpy is None:
        return_numpy = was_numpy
    if was_numpy and return_numpy and host_chunk != 0 and isinstance(W0, np.ndarray) and isinstance(H0, np.ndarray) and np.ndim(X) == 3:
        chunk = _pipeline_chunk(np.asarray(X), host_chunk)
        if chunk:
            kw = dict(max_iter=max_iter, tol=tol, check_every=check_every, update_H=update_H, l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H,
                      l2_reg_W=l2_reg_W, l2_reg_H=l2_reg_H, beta_loss=beta_loss)
            return _fit_batched_pipelined(np.asarray(X), W0, H0, dev, chunk, kw, handle)
    Xt = _as_device_tensor(X, dev)
    if Xt.dim() == 2:
        Xt = Xt.unsqueeze(0)
    if Xt.dim() != 3:
        raise ValueError(f"X must be [B, T, m] or [T, m], got shape {tuple(Xt.shape)}")
    if Xt.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"X must be float32 or float64, got {Xt.dtype}")
    B, T, m = Xt.shape
    if B == 0 or T == 0 or m == 0:
        raise ValueError("empty input")
    Wt = _as_device_tensor(W0, dev, Xt.dtype)
    Ht = _as_device_tensor(H0, dev, Xt.dtype)
    if Wt.dim() == 2:
        Wt = Wt.unsqueeze(0)
    if Ht.dim() == 2:
        Ht = Ht.unsqueeze(0)
    k = Ht.shape[1]
    if tuple(Wt.shape) != (B, T, k) or tuple(Ht.shape) != (B, k, m):
        raise ValueError(f"W0 must be [{B}, {T}, k] and H0 [{B}, k, {m}]; got {tuple(Wt.shape)} and {tuple(Ht.shape)}")
    # W and H are in/out in the C ABI: work on private contiguous copies unless told otherwise
    def private(t, src):
        if isinstance(src, torch.Tensor) and not overwrite_init:
            return t.clone(memory_format=torch.contiguous_format)
        return t.contiguous()

    Wt = private(Wt, W0)
    Ht = private(Ht, H0)
    x_layout, ldx, xbs, Xt = _x_layout(Xt)

    p = make_problem(B, T, m, k, x_layout=x_layout, ldx=ldx, x_batch_stride=xbs, update_H=update_H,
                     max_iter=max_iter, tol=tol, check_every=check_every, l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H,
                     l2_reg_W=l2_reg_W, l2_reg_H=l2_reg_H, loss=beta_loss_code(beta_loss))
    err = torch.empty((B,), dtype=Xt.dtype, device=dev)
    n_iter = torch.empty((B,), dtype=torch.int32, device=dev)
    sse = torch.empty((B, m), dtype=Xt.dtype, device=dev)
    xsq = torch.empty((B, m), dtype=Xt.dtype, device=dev)
    h = handle if handle is not None else _lib.get_handle(dev.index)
    lib = _lib.load()
    fn = lib.hipnmf_fit_batched_f32 if Xt.dtype == torch.float32 else lib.hipnmf_fit_batched_f64
    if not _inputs_ready:  # inputs were produced on torch's stream; the handle has its own.  (The transfer pipeline hands over
        torch.cuda.synchronize(dev)  # tensors whose copies have completed: a device-wide wait here would serialise its stages.)
    _lib.check(fn(h.ptr, ctypes.byref(p), Xt.data_ptr(), Wt.data_ptr(), Ht.data_ptr(), err.data_ptr(),
                  n_iter.data_ptr(), sse.data_ptr(), xsq.data_ptr()))
    vaf = torch.cat([(1 - sse.sum(dim=1) / xsq.sum(dim=1)).unsqueeze(1), 1 - sse / xsq], dim=1)
    res = BatchedResult(Wt, Ht, n_iter, err, vaf, sse, xsq, h.last_kernel_ms())
    if return_numpy:
        res = BatchedResult(*(t.cpu().numpy() for t in (Wt, Ht, n_iter, err, vaf, sse, xsq)), res.kernel_ms)
    return res


# ------------------------------------------------------------------------------------------------
# Host-resident batches.  The reference's input lives in host memory (a DataFrame: analysis.py:739-746); uploading a whole
# batch before the first kernel and downloading W after the last one adds the transfer time to the fit (4096 x (16 x 10 000)
# fp32: 2.6 GB up, 0.8 GB down against ~0.2 s of compute).  Chunks of the batch go through a three-stage pipeline instead:
# worker threads upload chunk i + 1 (and i + 2) and download chunk i - 1 on their own streams while the calling thread fits
# chunk i -- the library call blocks its host thread and releases the GIL, the copies block theirs.  The per-matrix results do
# not depend on the chunking: every chunk is routed as the whole batch would be (hipnmf_set_batch_hint -- a tail chunk below half the
# CUs would otherwise take another kernel family) and a matrix's arithmetic does not depend on its neighbours
# (tests/test_gpu_pipeline.py compares bitwise, tail chunk included).
_never_free: list = []  # arrays whose page lock could not be released: kept alive for the life of the process (see _unregister)


def _unregister(rt, arr, addr: int) -> bool:
    """hipHostUnregister with its status CHECKED.  A page lock that cannot be released (the call fails) while the array goes back to
    the allocator leaves the runtime with a pinned range whose pages are gone; the next array that lands on those addresses is
    then copied through stale bookkeeping (the abort of profiles/r06_abort_hunt.md, defect 4).  If the release fails the array is
    parked in ``_never_free`` instead: its addresses are never handed out again."""
    try:
        ok = int(rt.cudaHostUnregister(addr)) == 0
    except Exception:  # noqa: BLE001
        ok = False
    if not ok:
        _never_free.append(arr)
        import warnings

        warnings.warn("hipHostUnregister failed: the array stays allocated (and page-locked) for the life of the process", RuntimeWarning)
    return ok


_pipeline_trace = None  # development aid (tools/probes/host_pipeline_trace.py): a list to receive (stage, chunk, t_start, t_end)
PIPELINE_MIN_BYTES = 256 << 20   # X smaller than this: one upload, one fit
PIPELINE_MAX_CHUNK_BYTES = 1 << 30  # ... but never more than this per chunk
PIPELINE_CHUNK_BYTES = 192 << 20  # automatic chunk: about this much of X, a multiple of 256 matrices (whole rounds of workgroups)


def _pipeline_chunk(X, host_chunk) -> int:
    """Matrices per chunk, or 0 for no pipeline."""
    B = X.shape[0]
    if X.dtype not in (np.float32, np.float64) or any(st < 0 for st in X.strides):
        return 0  # the ordinary path reports unsupported dtypes; torch.from_numpy cannot view negative strides
    if host_chunk is not None:
        c = int(host_chunk)
        return c if 0 < c < B else 0
    if X.nbytes < PIPELINE_MIN_BYTES:
        return 0
    per = max(1, X.nbytes // B)
    # whole rounds of workgroups: at least 256 matrices per chunk (fewer leave CUs idle in the one-workgroup-per-matrix kernels,
    # and every chunk of the row-sliced general shapes would pay its own graph build); matrices too big for that: one call
    c = max(256, PIPELINE_CHUNK_BYTES // per // 256 * 256)
    if c * per > PIPELINE_MAX_CHUNK_BYTES:
        return 0
    return c if c < B else 0


def _fit_batched_pipelined(X, W0, H0, dev, chunk: int, kw, handle=None, ctx=None) -> BatchedResult:
    from concurrent.futures import ThreadPoolExecutor
    import time as _time

    torch = _torch()
    t_enter = _time.perf_counter()
    W0, H0 = np.asarray(W0), np.asarray(H0)
    B, T, m = X.shape
    if X.dtype not in (np.float32, np.float64):
        raise TypeError(f"X must be float32 or float64, got {X.dtype}")
    if W0.ndim != 3 or H0.ndim != 3 or W0.shape[:2] != (B, T) or H0.shape[0] != B or H0.shape[2] != m or W0.shape[2] != H0.shape[1]:
        raise ValueError(f"W0 must be [{B}, {T}, k] and H0 [{B}, k, {m}]; got {W0.shape} and {H0.shape}")
    k = H0.shape[1]
    dt = X.dtype
    tdt = torch.float32 if dt == np.float32 else torch.float64
    W0, H0 = W0.astype(dt, copy=False), H0.astype(dt, copy=False)
    bounds = [(lo, min(lo + chunk, B)) for lo in range(0, B, chunk)]
    out_pinned = False
    if ctx is not None and ctx.reuse_outputs:
        out, out_pinned = ctx._outputs(B, T, k, m, dt)  # the batch's own result arrays, registered once, reused from call to call
    else:
        w_raw = np.empty(B * T * k * dt.itemsize + 4096, np.uint8)  # W page-aligned: its chunks can be pinned one by one
        w_off = (-w_raw.ctypes.data) % 4096
        out = BatchedResult(w_raw[w_off: w_off + B * T * k * dt.itemsize].view(dt).reshape(B, T, k), np.empty((B, k, m), dt),
                            np.empty((B,), np.int32), np.empty((B,), dt), np.empty((B, 1 + m), dt), np.empty((B, m), dt),
                            np.empty((B, m), dt), 0.0)
    h = handle if handle is not None else _lib.get_handle(dev.index)

    def traced(stage):
        def deco(fn):
            def run(i, *a):
                t0 = _time.perf_counter()
                try:
                    return fn(i, *a)
                finally:
                    if _pipeline_trace is not None:
                        _pipeline_trace.append((stage, i, t0, _time.perf_counter()))
            return run
        return deco

    # Device side: three slots of chunk buffers allocated once (the transfer threads copy on their own streams, and torch's
    # caching allocator keeps a pool per stream: tensors allocated inside the upload threads meant a hipMalloc per chunk --
    # uploads of 7 ms instead of 4, and 20 - 35 ms for the first ones).  X keeps the strides of the caller's array (row- or
    # channel-major per matrix), W and H are updated in place and downloaded from the slot.
    NSLOT = min(3, len(bounds))
    es = X.itemsize
    channel_major = X.strides == (T * m * es, es, T * es)  # every matrix stored m x T (a DataFrame's F order), seen as T x m
    def make_slots():
        made = []
        for _ in range(NSLOT):
            xs = torch.empty((chunk, m, T), dtype=tdt, device=dev).transpose(1, 2) if channel_major else torch.empty((chunk, T, m), dtype=tdt, device=dev)
            made.append((xs, torch.empty((chunk, T, k), dtype=tdt, device=dev), torch.empty((chunk, k, m), dtype=tdt, device=dev)))
        return made

    slots = ctx._slots((NSLOT, chunk, T, m, k, tdt, channel_major), make_slots) if ctx is not None else make_slots()
    # inputs registered by a HostBatch are page-locked already: asynchronous copy-engine transfers, no on-the-fly pinning by the driver
    x_pin = ctx is not None and ctx.is_registered(X)
    w_pin = ctx is not None and ctx.is_registered(W0)
    h_pin = ctx is not None and ctx.is_registered(H0)
    # (no device-wide synchronisation anywhere in this function: with the 32 pooled side streams of a previous call around,
    #  hipDeviceSynchronize alone took 27 - 39 ms; every hand-over below is an event or a stream synchronisation)
    if _pipeline_trace is not None:
        _pipeline_trace.append(("alloc", 0, t_enter, _time.perf_counter()))

    @traced("upload")
    def upload(i, after):
        if after is not None:
            after.result()  # the slot's previous chunk has left it
        lo, hi = bounds[i]
        n = hi - lo
        torch.cuda.set_device(dev)
        xs, ws, hs = slots[i % NSLOT]
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            xs[:n].copy_(torch.from_numpy(X[lo:hi]), non_blocking=x_pin)
            ws[:n].copy_(torch.from_numpy(W0[lo:hi]), non_blocking=w_pin)
            hs[:n].copy_(torch.from_numpy(H0[lo:hi]), non_blocking=h_pin)
        st.synchronize()
        return xs[:n], ws[:n], hs[:n]

    # Results.  W (T x k per matrix: 0.8 GB at the headline batch) goes home chunk by chunk, each chunk's destination pinned in
    # place just before its copy (hipHostRegister, ~2 ms per 51 MB; the result array is page-aligned and the chunks are whole
    # pages) so that the copy is an asynchronous copy-engine transfer.  A copy straight into pageable, never-touched memory runs
    # at the page-fault rate and, as a shader copy, waits for the fit kernel that holds every CU (measured 3.5 - 7 GB/s: the
    # downloads fell 60 ms behind the fits).  Also measured and not kept: pinning the whole array in one call (34 - 42 ms; it
    # stalls the uploads, which pin their own pages at the same time: one 24 ms hole in the fits), pinned staging slots + a host
    # copy (erratic: stalls of 40 - 90 ms), two fitting lanes (two kernels in flight push the matrices being streamed out of the
    # Infinity Cache: 0.82 vs 0.86 of the device-resident rate).  The small outputs stay on the device until the last fit has
    # ended: one copy each.
    rt = torch.cuda.cudart()
    SMALL = ("H", "n_iter", "reconstruction_err", "vaf", "sse_col", "xsq_col")
    small_parts = [None] * len(bounds)
    import os as _os

    # HIPNMF_PIPELINE_PIN=0: never page-lock the result array chunk by chunk (pageable downloads: slower, and no register /
    # unregister traffic at all -- the switch for a long-lived host that prefers that; profiles/r06_abort_hunt.md, defect 4)
    page_ok = (out.W.ctypes.data % 4096 == 0 and (chunk * T * k * out.W.itemsize) % 4096 == 0
               and _os.environ.get("HIPNMF_PIPELINE_PIN", "1") != "0")

    @traced("download")
    def download(i, Wd, ready):
        lo, hi = bounds[i]
        torch.cuda.set_device(dev)
        st = torch.cuda.Stream(dev)
        st.wait_event(ready)
        dst = out.W[lo:hi]
        pinned = False
        if out_pinned:  # (HostBatch(reuse_outputs=True): the whole result array is registered for the batch's lifetime)
            with torch.cuda.stream(st):
                torch.from_numpy(dst).copy_(Wd, non_blocking=True)
            st.synchronize()
            return
        if page_ok:
            try:
                pinned = int(rt.cudaHostRegister(dst.ctypes.data, dst.nbytes, 0)) == 0
            except Exception:  # noqa: BLE001 -- the pageable copy below works, only slower
                pinned = False
        try:
            with torch.cuda.stream(st):
                torch.from_numpy(dst).copy_(Wd, non_blocking=pinned)
            st.synchronize()
        finally:
            if pinned:
                _unregister(rt, out.W, dst.ctypes.data)

    ms_total = 0.0
    # every chunk is fitted by the kernel the WHOLE batch would get (hipnmf_set_batch_hint): a tail chunk below half the CUs
    # would otherwise be routed like a small batch -- a valid fit, but not the bits of the one-call fit
    h.set_batch_hint(B)
    try:
        ms_total = 0.0
        with ThreadPoolExecutor(max_workers=4, thread_name_prefix="hipnmf-xfer") as pool:
            downs = {}
            ups = {0: pool.submit(upload, 0, None)}
            if len(bounds) > 1:  # one after the other: side by side the first two share the link and chunk 0 arrives twice as late
                ups[1] = pool.submit(upload, 1, ups[0])
            for i in range(len(bounds)):
                Xd, Wd, Hd = ups.pop(i).result()
                if i + 2 < len(bounds):  # (its slot was chunk i - 1's: free once that chunk's W is home)
                    ups[i + 2] = pool.submit(upload, i + 2, downs.get(i + 2 - NSLOT))
                r = fit_batched(Xd, Wd, Hd, device=dev, handle=h, return_numpy=False, overwrite_init=True, _inputs_ready=True, **kw)
                ms_total += r.kernel_ms
                # H leaves the slot first (a copy on torch's stream), THEN the event: the upload that reuses this slot is gated on the
                # download, the download on this event -- so it also orders the clone before the slot is overwritten (round-5 advisor
                # finding: the event used to be recorded before the clone was enqueued, and on the calling thread's current device)
                small_parts[i] = {name: getattr(r, name).clone() if name == "H" else getattr(r, name) for name in SMALL}
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(dev))
                downs[i] = pool.submit(download, i, r.W, ready)
                del Xd, Wd, Hd, r
            for f in downs.values():
                f.result()
        t_sm = _time.perf_counter()
        for name in SMALL:  # the device is idle now: one copy per small output
            getattr(out, name)[...] = torch.cat([p[name] for p in small_parts]).cpu().numpy()
        if _pipeline_trace is not None:
            _pipeline_trace.append(("small", 0, t_sm, _time.perf_counter()))
        out.kernel_ms = ms_total
        return out
    finally:
        h.set_batch_hint(0)


class HostBatch:
    """A host-resident batch that is fitted MORE THAN ONCE -- a rank range (``analysis.py:907-912`` fits the same frame for every
    rank), restarts, a parameter study: the caller's arrays are page-locked ONCE (``hipHostRegister`` in place, no copy) and the
    registration, the device slots of the transfer pipeline and (``reuse_outputs=True``) the page-locked result arrays are kept
    from call to call.  What a one-shot ``fit_batched(X_numpy, ...)`` pays on every call -- the driver pinning the caller's pages
    on the fly for the first chunk's upload (10 - 30 ms of an otherwise idle GPU), device allocations, per-chunk pinning of
    the result -- is paid once here.

        hb = HostBatch(X, W0, H0)                 # X [B, T, m] float32 / float64, dense
        r5 = hb.fit(max_iter=500, tol=0)          # NumPy in, NumPy out, chunks uploaded / fitted / downloaded concurrently
        r6 = hb.fit(W0_k6, H0_k6, max_iter=500)   # other starting points (registered on first sight, remembered by identity)
        hb.close()                                # or ``with HostBatch(...) as hb:``

    ``reuse_outputs=True``: ``fit`` returns views of result arrays owned by this object, one set per ``n_components``; a later
    ``fit`` with the same ``n_components`` overwrites them (copy what must survive).  ``keep_on_device=True`` additionally keeps
    X in HBM after the first call (288 GB hold any batch the reference's users have): later calls upload only W0 / H0.
    Results are bitwise those of ``fit_batched`` on the same arrays (tests/test_gpu_pipeline.py).  One ``fit`` at a time per object (the
    device slots and result arrays are the object's); several objects may be used from several threads."""

    MIN_REGISTER_BYTES = 1 << 20  # arrays below this are never page-locked (see _register)

    def __init__(self, X, W0=None, H0=None, *, device=None, host_chunk: Optional[int] = None, reuse_outputs: bool = False,
                 keep_on_device: bool = False):
        torch = _torch()
        X = np.asarray(X)
        if X.ndim != 3 or X.dtype not in (np.float32, np.float64):
            raise ValueError("HostBatch takes X [B, T, m] float32 / float64 in host memory")
        self.dev = resolve_device(device)
        self.X = X
        self.reuse_outputs = bool(reuse_outputs)
        self.keep_on_device = bool(keep_on_device)
        self.host_chunk = host_chunk
        self._rt = torch.cuda.cudart()
        self._reg: dict = {}      # id(base buffer) -> (array kept alive, address, bytes)
        self._slot_cache: dict = {}
        self._out_cache: dict = {}
        self._Xd = None
        self._lock = threading.Lock()
        self.register_seconds = 0.0
        self._closed = False
        self._register(X)
        self.W0, self.H0 = (None if W0 is None else np.asarray(W0)), (None if H0 is None else np.asarray(H0))
        for a in (self.W0, self.H0):
            if a is not None:
                self._register(a)

    # -- registration -----------------------------------------------------------------------------------------------
    @staticmethod
    def _span(a):
        """(address, bytes) of the memory a dense array occupies (any axis order, positive strides)."""
        if a.size == 0 or any(st < 0 for st in a.strides):
            return None
        lo = a.ctypes.data
        hi = lo + sum((n - 1) * st for n, st in zip(a.shape, a.strides)) + a.itemsize
        return lo, hi - lo

    def _register(self, a) -> bool:
        import time as _time

        span = self._span(a)
        if span is None or span[0] in self._reg:
            return span is not None and span[0] in self._reg
        if span[1] < self.MIN_REGISTER_BYTES:
            # A small array lives on pages of the allocator's heap that it shares with other objects (and with other small arrays
            # this batch might register): page-locking such pages twice and unlocking them once, or leaving a range of the heap
            # known to the runtime as pinned, is how a LATER pageable copy of an unrelated array aborted inside the HIP runtime
            # (3 of 8 full-suite runs, round 6: profiles/r06_abort_hunt.md).  Only arrays that own their pages -- mmap'ed
            # allocations, far beyond the allocator's threshold -- are registered; the small ones cost microseconds to copy anyway.
            return False
        t0 = _time.perf_counter()
        try:
            ok = int(self._rt.cudaHostRegister(span[0], span[1], 0)) == 0
        except Exception:  # noqa: BLE001 -- an unregistered array still works (pageable copies), only slower
            ok = False
        self.register_seconds += _time.perf_counter() - t0
        if ok:
            self._reg[span[0]] = (a, span[1])
        return ok

    def is_registered(self, a) -> bool:
        span = self._span(np.asarray(a))
        if span is None:
            return False
        for addr, (_, nbytes) in self._reg.items():
            if addr <= span[0] and span[0] + span[1] <= addr + nbytes:
                return True
        return False

    # -- caches used by _fit_batched_pipelined ------------------------------------------------------------------------
    def _slots(self, key, make):
        with self._lock:
            if key not in self._slot_cache:
                self._slot_cache.clear()  # one geometry at a time: a rank range changes k, the old slots are released
                self._slot_cache[key] = make()
            return self._slot_cache[key]

    def _outputs(self, B, T, k, m, dt):
        key = (B, T, k, m, np.dtype(dt).str)
        with self._lock:
            hit = self._out_cache.get(key)
            if hit is None:
                raw = np.empty(B * T * k * dt.itemsize + 4096, np.uint8)
                off = (-raw.ctypes.data) % 4096
                W = raw[off: off + B * T * k * dt.itemsize].view(dt).reshape(B, T, k)
                out = BatchedResult(W, np.empty((B, k, m), dt), np.empty((B,), np.int32), np.empty((B,), dt), np.empty((B, 1 + m), dt),
                                    np.empty((B, m), dt), np.empty((B, m), dt), 0.0)
                hit = (out, self._register(W), raw)
                self._out_cache[key] = hit
            return hit[0], hit[1]

    # -- the call -----------------------------------------------------------------------------------------------------
    def fit(self, W0=None, H0=None, *, handle: Optional[_lib.Handle] = None, **kw) -> BatchedResult:
        """``fit_batched(self.X, W0, H0, **kw)`` through the kept registration; ``W0`` / ``H0`` default to the constructor's."""
        if self._closed:
            raise ValueError("HostBatch is closed")
        W0 = self.W0 if W0 is None else np.asarray(W0)
        H0 = self.H0 if H0 is None else np.asarray(H0)
        if W0 is None or H0 is None:
            raise ValueError("no starting point: pass W0 and H0 here or to the constructor")
        self._register(W0), self._register(H0)
        B = self.X.shape[0]
        for name in ("devices", "device", "return_numpy", "overwrite_init", "host_chunk"):
            if name in kw:
                raise ValueError(f"HostBatch.fit does not take {name}=")
        solver_kw = dict(max_iter=kw.pop("max_iter", 200), tol=kw.pop("tol", 1e-4), check_every=kw.pop("check_every", 10),
                         update_H=kw.pop("update_H", True), l1_reg_W=kw.pop("l1_reg_W", 0.0), l1_reg_H=kw.pop("l1_reg_H", 0.0),
                         l2_reg_W=kw.pop("l2_reg_W", 0.0), l2_reg_H=kw.pop("l2_reg_H", 0.0), beta_loss=kw.pop("beta_loss", "frobenius"))
        if kw:
            raise TypeError(f"unexpected arguments {sorted(kw)}")
        if self.keep_on_device:
            return self._fit_resident(W0, H0, solver_kw, handle)
        chunk = _pipeline_chunk(self.X, self.host_chunk)
        if not chunk:  # small batches: one upload (from page-locked memory), one fit, one download
            chunk = B
        return _fit_batched_pipelined(self.X, W0, H0, self.dev, chunk, solver_kw, handle, ctx=self)

    def _fit_resident(self, W0, H0, solver_kw, handle):
        torch = _torch()
        if self._Xd is None:
            self._Xd = _as_device_tensor(self.X, self.dev)
        def up(a):  # asynchronous only straight out of the registered array itself (a contiguous copy is an unregistered temporary)
            return torch.from_numpy(np.ascontiguousarray(a)).to(self.dev, self._Xd.dtype, non_blocking=a.flags.c_contiguous and self.is_registered(a))

        Wd, Hd = up(W0), up(H0)
        r = fit_batched(self._Xd, Wd, Hd, device=self.dev, handle=handle, return_numpy=False, overwrite_init=True, **solver_kw)
        if self.reuse_outputs:
            B, T, m = self.X.shape
            out, pinned = self._outputs(B, T, Hd.shape[1], m, self.X.dtype)
            torch.from_numpy(out.W).copy_(r.W, non_blocking=pinned)
            for name in ("H", "n_iter", "reconstruction_err", "vaf", "sse_col", "xsq_col"):
                getattr(out, name)[...] = getattr(r, name).cpu().numpy()
            torch.cuda.current_stream(self.dev).synchronize()
            out.kernel_ms = r.kernel_ms
            return out
        return BatchedResult(*(t.cpu().numpy() for t in (r.W, r.H, r.n_iter, r.reconstruction_err, r.vaf, r.sse_col, r.xsq_col)), r.kernel_ms)

    def close(self):
        """Give the page locks and the device memory back (idempotent).  The arrays themselves are the caller's, untouched."""
        if self._closed:
            return
        self._closed = True
        if _lib._closed:  # the interpreter is past the engine's exit hook: the page locks go with the process, no runtime call any more
            self._reg.clear()
            self._slot_cache.clear()
            self._out_cache.clear()
            self._Xd = None
            return
        try:
            _torch().cuda.synchronize(self.dev)
        except Exception:  # noqa: BLE001
            pass
        for addr, (arr, _n) in list(self._reg.items()):
            _unregister(self._rt, arr, addr)
        self._reg.clear()
        self._slot_cache.clear()
        self._out_cache.clear()
        self._Xd = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def _gather_batched(parts, as_numpy: bool, ragged: bool = False) -> BatchedResult:
    """``parts``: the per-device :class:`BatchedResult` objects in batch order -> one result on the host."""
    from .multi_gpu import cat_host, to_host

    def cat(name):
        return cat_host([to_host(getattr(r, name), as_numpy) for r in parts])

    if ragged:
        W = [to_host(w, as_numpy) for r in parts for w in r.W]
    else:
        W = cat("W")
    return BatchedResult(W, cat("H"), cat("n_iter"), cat("reconstruction_err"), cat("vaf"), cat("sse_col"), cat("xsq_col"),
                         max(r.kernel_ms for r in parts))


def _fit_batched_scattered(X, W0, H0, devices, return_numpy, kw) -> BatchedResult:
    from .multi_gpu import resolve_devices, scatter

    torch = _torch()
    devs = resolve_devices(devices)
    as_numpy = (not isinstance(X, torch.Tensor)) if return_numpy is None else bool(return_numpy)
    if not isinstance(X, torch.Tensor):
        X = np.asarray(X)
    if X.ndim == 2:
        X, W0, H0 = X[None], W0[None], H0[None]
    if X.ndim != 3:
        raise ValueError(f"X must be [B, T, m] or [T, m], got shape {tuple(X.shape)}")
    B = X.shape[0]
    if B == 0:
        raise ValueError("empty input")
    if len(W0) != B or len(H0) != B:
        raise ValueError("W0 and H0 must hold one starting point per matrix of X")

    host_in = isinstance(X, np.ndarray)  # host-resident slices go through the chunked transfer pipeline of their device

    def work(lo, hi, d):
        return fit_batched(X[lo:hi], W0[lo:hi], H0[lo:hi], device=f"cuda:{d}", return_numpy=host_in, **kw)

    parts = [r for _, _, _, r in scatter(B, devs, work)]
    return _gather_batched(parts, as_numpy)


def fit_batched_multi_gpu(X, W0, H0, *, devices="all", **kw) -> BatchedResult:
    """``fit_batched(..., devices=devices)`` (default: every visible GPU).  The multi-process flavour (one rank per GPU
    under ``torch.distributed.run``) lives in ``bench.py``."""
    return fit_batched(X, W0, H0, devices=devices, **kw)


def partition(n_items: int, n_parts: int):
    """Contiguous, balanced ``[lo, hi)`` ranges (first ``n_items % n_parts`` parts get one extra)."""
    base, extra = divmod(int(n_items), int(n_parts))
    out, lo = [], 0
    for i in range(n_parts):
        hi = lo + base + (1 if i < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


# ------------------------------------------------------------------------------------------------
# Device-side initialisation and the per-trial rank sweep (BASELINE.json config #4)
def random_init_batched(X, k: int, *, seed: int = 0):
    """sklearn's ``init='random'`` scaling (``_nmf.py:303-314``: ``sqrt(X.mean()/k) * |N(0,1)|``) for every
    matrix of a device batch ``X [B, T, m]``, drawn from torch's generator on the device (same law as
    sklearn, not the same stream).  Returns ``(W0 [B, T, k], H0 [B, k, m])``."""
    torch = _torch()
    B, T, m = X.shape
    g = torch.Generator(device=X.device)
    g.manual_seed(int(seed))
    avg = torch.sqrt(X.mean(dim=(1, 2)) / k).view(B, 1, 1)
    H0 = avg * torch.randn((B, k, m), generator=g, device=X.device, dtype=X.dtype).abs_()
    W0 = avg * torch.randn((B, T, k), generator=g, device=X.device, dtype=X.dtype).abs_()
    return W0, H0


def random_init_device(X, k: int, *, seed: int = 0, first_matrix: int = 0, handle: Optional[_lib.Handle] = None):
    """The same starting points from the library's own generator (``hipnmf_random_init_*``: Philox4x32-10 keyed by
    ``(seed, first_matrix + b, element)``), i.e. independent of the layout and of how a batch is split over GPUs;
    this is what ``hipnmf_rank_sweep_*`` draws internally.  ``X [B, T, m]`` device tensor (any dense layout).
    Returns ``(W0 [B, T, k], H0 [B, k, m])``."""
    torch = _torch()
    Xt = X if X.dim() == 3 else X.unsqueeze(0)
    B, T, m = Xt.shape
    layout, ldx, xbs, Xt = _x_layout(Xt)
    p = make_problem(B, T, m, k, x_layout=layout, ldx=ldx, x_batch_stride=xbs, w_layout=_lib.W_ROW_MAJOR)
    W0 = torch.empty((B, T, k), dtype=Xt.dtype, device=Xt.device)
    H0 = torch.empty((B, k, m), dtype=Xt.dtype, device=Xt.device)
    h = handle if handle is not None else _lib.get_handle(Xt.device.index)
    fn = getattr(_lib.load(), "hipnmf_random_init_f32" if Xt.dtype == torch.float32 else "hipnmf_random_init_f64")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    torch.cuda.synchronize(Xt.device)
    _lib.check(fn(h.ptr, ctypes.addressof(p), int(seed), int(first_matrix), Xt.data_ptr(), W0.data_ptr(), H0.data_ptr()))
    return W0, H0


def rank_sweep_native(X, k_min: int, k_max: int, *, vaf_threshold: float = 0.90, max_iter: int = 500, tol: float = 1e-4,
                      seed: int = 0, first_matrix: int = 0, device=None, handle: Optional[_lib.Handle] = None,
                      stop_at_threshold: bool = False, devices=None) -> "RankSweepResult":
    """:func:`rank_sweep_batched` as ONE library call (``hipnmf_rank_sweep_*``: random starting points, fits, VAF
    table and threshold selection inside the library; usable from any host language).  Frobenius loss,
    ``init='random'`` from the library's generator; ``vaf[k]`` holds the all-muscles column only.

    ``stop_at_threshold=True`` (``hipnmf_rank_sweep_stop_*``): a trial whose VAF has reached ``vaf_threshold`` is not
    fitted at the higher ranks (BASELINE.json config #4's "stop"); the skipped (trial, rank) pairs report NaN VAF /
    residual, 0 iterations and zero components, ``selected`` is identical to the compute-all mode.  ``kernel_ms``
    is not available from the native sweep (several launches); time the call.

    ``devices=``: trials scattered over several GPUs, results on the host.  The library's generator is keyed by
    ``(seed, first_matrix + b, element)``, so the scattered sweep returns exactly what the one-device call returns."""
    torch = _torch()
    if devices is not None:
        def one(Xs, lo, d):
            return rank_sweep_native(Xs, k_min, k_max, vaf_threshold=vaf_threshold, max_iter=max_iter, tol=tol, seed=seed,
                                     first_matrix=first_matrix + lo, device=f"cuda:{d}", stop_at_threshold=stop_at_threshold)
        return _rank_sweep_scattered(X, devices, one)
    dev = resolve_device(device)
    Xt = _as_device_tensor(X, dev)
    if Xt.dim() == 2:
        Xt = Xt.unsqueeze(0)
    if Xt.dtype not in (torch.float32, torch.float64):
        Xt = Xt.to(torch.float64)
    B, T, m = Xt.shape
    if not 1 <= k_min <= k_max <= m:
        raise ValueError("invalid number of components")
    layout, ldx, xbs, Xt = _x_layout(Xt)
    nk = k_max - k_min + 1
    p = make_problem(B, T, m, k_max, x_layout=layout, ldx=ldx, x_batch_stride=xbs, w_layout=_lib.W_ROW_MAJOR,
                     max_iter=max_iter, tol=tol)
    W_ws = torch.empty((B, T, k_max), dtype=Xt.dtype, device=dev)
    H_all = torch.empty((B * m * sum(range(k_min, k_max + 1)),), dtype=Xt.dtype, device=dev)
    vaf = torch.empty((B, nk), dtype=Xt.dtype, device=dev)
    err = torch.empty((B, nk), dtype=Xt.dtype, device=dev)
    n_iter = torch.empty((B, nk), dtype=torch.int32, device=dev)
    sel = torch.empty((B,), dtype=torch.int32, device=dev)
    h = handle if handle is not None else _lib.get_handle(dev.index)
    name = "hipnmf_rank_sweep_stop" if stop_at_threshold else "hipnmf_rank_sweep"
    fn = getattr(_lib.load(), name + ("_f32" if Xt.dtype == torch.float32 else "_f64"))
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_uint64, ctypes.c_int32] + \
                  [ctypes.c_void_p] * 7
    fn.restype = ctypes.c_int
    torch.cuda.synchronize(dev)
    _lib.check(fn(h.ptr, ctypes.addressof(p), int(k_min), int(k_max), float(vaf_threshold), int(seed), int(first_matrix),
                  Xt.data_ptr(), W_ws.data_ptr(), H_all.data_ptr(), vaf.data_ptr(), sel.data_ptr(), err.data_ptr(), n_iter.data_ptr()))
    ranks = list(range(k_min, k_max + 1))
    comps, off = {}, 0
    for k in ranks:
        comps[k] = H_all[off:off + B * k * m].view(B, k, m)
        off += B * k * m
    return RankSweepResult(ranks, vaf, {k: vaf[:, i:i + 1] for i, k in enumerate(ranks)}, {k: n_iter[:, i] for i, k in enumerate(ranks)},
                           {k: err[:, i] for i, k in enumerate(ranks)}, comps, sel.to(torch.int64), float("nan"))


@dataclass
class RankSweepResult:
    """Per-trial rank sweep: ``vaf_all[b, i]`` is the "All signals" VAF of trial ``b`` at rank ``ranks[i]``;
    ``selected[b]`` the first rank whose VAF reaches ``vaf_threshold`` (``-1`` if none does) -- the
    thresholding the reference leaves to the user (``analysis.py:753-756``, Rabbi et al. 2020 protocol)."""

    ranks: list
    vaf_all: "object"  # [B, n_ranks]
    vaf: dict  # rank -> [B, 1 + m]
    n_iter: dict  # rank -> [B]
    reconstruction_err: dict  # rank -> [B]
    components: dict  # rank -> [B, k, m]
    selected: "object"  # [B] int64
    kernel_ms: float


def rank_sweep_batched(X, k_min: int, k_max: int, *, vaf_threshold: float = 0.90, max_iter: int = 500,
                       tol: float = 1e-4, seed: int = 0, device=None, keep_W: bool = False,
                       init: str = "random", beta_loss="frobenius", devices=None) -> RankSweepResult:
    """``find_synergies(df, k_min, k_max)`` for a whole batch of trials: one batched fit per rank, VAF per
    trial and rank, and the smallest rank with VAF >= ``vaf_threshold``.  ``init='random'`` draws the starting
    factors on the device; ``'nndsvda'`` / ``'nndsvd'`` (sklearn's default family) use the on-device NNDSVD.

    ``devices=``: trials scattered over several GPUs, results on the host.  The slice that starts at trial ``lo`` draws its
    random starting points with ``seed + lo`` (torch's generator is a stream, not a counter: the draws of a slice cannot
    be those of the same trials inside a longer batch; same law) -- the NNDSVD inits do not depend on the split."""
    torch = _torch()
    if devices is not None:
        def one(Xs, lo, d):
            return rank_sweep_batched(Xs, k_min, k_max, vaf_threshold=vaf_threshold, max_iter=max_iter, tol=tol, seed=seed + lo,
                                      device=f"cuda:{d}", keep_W=keep_W, init=init, beta_loss=beta_loss)
        return _rank_sweep_scattered(X, devices, one)
    dev = resolve_device(device)
    Xt = _as_device_tensor(X, dev)
    if Xt.dim() == 2:
        Xt = Xt.unsqueeze(0)
    B, T, m = Xt.shape
    if not 1 <= k_min <= k_max <= m:
        raise ValueError("invalid number of components")
    ranks = list(range(k_min, k_max + 1))
    vaf, n_iter, err, comps, Ws = {}, {}, {}, {}, {}
    total_ms = 0.0
    for k in ranks:
        if init == "random":
            W0, H0 = random_init_batched(Xt, k, seed=seed + k)
        else:
            from .init import nndsvd_init_batched

            W0, H0 = nndsvd_init_batched(Xt, k, init=init, device=dev)
        r = fit_batched(Xt, W0, H0, max_iter=max_iter, tol=tol, beta_loss=beta_loss, device=dev, return_numpy=False,
                        overwrite_init=True)
        vaf[k], n_iter[k], err[k], comps[k] = r.vaf, r.n_iter, r.reconstruction_err, r.H
        if keep_W:
            Ws[k] = r.W
        total_ms += r.kernel_ms
    vaf_all = torch.stack([vaf[k][:, 0] for k in ranks], dim=1)
    ok = vaf_all >= vaf_threshold
    first = torch.where(ok.any(dim=1), ok.float().argmax(dim=1), torch.full((B,), -1, device=dev, dtype=torch.long))
    selected = torch.where(first >= 0, first + k_min, first)
    res = RankSweepResult(ranks, vaf_all, vaf, n_iter, err, comps, selected, total_ms)
    if keep_W:
        res.W = Ws
    return res


def _rank_sweep_scattered(X, devices, one) -> RankSweepResult:
    """Trials of ``X [B, T, m]`` in contiguous slices over ``devices``; ``one(X_slice, lo, device_index)`` is the one-device
    sweep of a slice.  Gathers on the host (CPU tensors) in trial order."""
    from .multi_gpu import cat_host, resolve_devices, scatter, to_host

    torch = _torch()
    devs = resolve_devices(devices)
    Xa = X if isinstance(X, torch.Tensor) else np.asarray(X)
    if Xa.ndim == 2:
        Xa = Xa[None]
    parts = [r for _, _, _, r in scatter(Xa.shape[0], devs, lambda lo, hi, d: one(Xa[lo:hi], lo, d))]
    ranks = parts[0].ranks

    def cat(get):
        return cat_host([to_host(get(r), False) for r in parts])

    res = RankSweepResult(ranks, cat(lambda r: r.vaf_all), {k: cat(lambda r: r.vaf[k]) for k in ranks},
                          {k: cat(lambda r: r.n_iter[k]) for k in ranks}, {k: cat(lambda r: r.reconstruction_err[k]) for k in ranks},
                          {k: cat(lambda r: r.components[k]) for k in ranks}, cat(lambda r: r.selected),
                          max(r.kernel_ms for r in parts))
    if all(hasattr(r, "W") for r in parts):
        res.W = {k: cat(lambda r: r.W[k]) for k in ranks}
    return res


# ------------------------------------------------------------------------------------------------
# Ragged batches: trials of unequal length (gait cycles, segments of a recording)
def fit_ragged(Xs, W0s, H0s, *, max_iter: int = 200, tol: float = 1e-4, check_every: int = 10,
               update_H: bool = True, l1_reg_W: float = 0.0, l1_reg_H: float = 0.0, l2_reg_W: float = 0.0,
               l2_reg_H: float = 0.0, beta_loss="frobenius", device=None, handle: Optional[_lib.Handle] = None, devices=None):
    """Factorise ``B`` matrices with different numbers of rows in one launch (``devices=``: one launch per GPU, the
    matrices scattered in contiguous runs balanced by their rows; results on the host as CPU tensors).

    ``Xs[b]`` is ``(T_b, m)``, ``W0s[b]`` ``(T_b, k)``, ``H0s[b]`` ``(k, m)`` (NumPy or torch; one dtype, one
    ``m`` and one ``k`` for the whole batch).  The matrices are packed on the device in the engine-native
    layouts (each padded to a multiple of 4 rows) and handed to ``hipnmf_fit_ragged_*``; one workgroup per
    matrix runs the whole fit, whatever the mix of lengths.  Returns a :class:`BatchedResult` whose ``W`` is
    a list of ``(T_b, k)`` tensors; the other fields are batched tensors as in :func:`fit_batched`.
    """
    torch = _torch()
    B = len(Xs)
    if B == 0 or len(W0s) != B or len(H0s) != B:
        raise ValueError("Xs, W0s and H0s must be non-empty lists of equal length")
    if devices is not None:
        from .multi_gpu import resolve_devices, scatter

        kw = dict(max_iter=max_iter, tol=tol, check_every=check_every, update_H=update_H, l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H,
                  l2_reg_W=l2_reg_W, l2_reg_H=l2_reg_H, beta_loss=beta_loss)
        parts = scatter(B, resolve_devices(devices),
                        lambda lo, hi, d: fit_ragged(Xs[lo:hi], W0s[lo:hi], H0s[lo:hi], device=f"cuda:{d}", **kw),
                        weights=[int(x.shape[0]) for x in Xs])
        return _gather_batched([r for _, _, _, r in parts], False, ragged=True)
    dev = resolve_device(device)
    first = _as_device_tensor(Xs[0], dev)
    dtype = first.dtype if first.dtype in (torch.float32, torch.float64) else torch.float64
    m = first.shape[1]
    k = _as_device_tensor(H0s[0], dev).shape[0]
    Ts = [int(x.shape[0]) for x in Xs]
    lds = [(t + 3) // 4 * 4 for t in Ts]
    x_off, w_off, xo, wo = [], [], 0, 0
    for ld in lds:
        x_off.append(xo)
        w_off.append(wo)
        xo += m * ld
        wo += (k * ld + 3) // 4 * 4
    Xp = torch.zeros((xo,), dtype=dtype, device=dev)
    Wp = torch.zeros((wo,), dtype=dtype, device=dev)
    Hp = torch.empty((B, k, m), dtype=dtype, device=dev)
    for b in range(B):
        xb = _as_device_tensor(Xs[b], dev, dtype)
        wb = _as_device_tensor(W0s[b], dev, dtype)
        hb = _as_device_tensor(H0s[b], dev, dtype)
        if tuple(xb.shape) != (Ts[b], m) or tuple(wb.shape) != (Ts[b], k) or tuple(hb.shape) != (k, m):
            raise ValueError(f"matrix {b}: expected X ({Ts[b]}, {m}), W0 ({Ts[b]}, {k}), H0 ({k}, {m})")
        Xp[x_off[b]: x_off[b] + m * lds[b]].view(m, lds[b])[:, : Ts[b]] = xb.t()
        Wp[w_off[b]: w_off[b] + k * lds[b]].view(k, lds[b])[:, : Ts[b]] = wb.t()
        Hp[b] = hb
    desc = np.array([[Ts[b], x_off[b], lds[b], w_off[b]] for b in range(B)], dtype=np.int64)
    Tmax = max(Ts)
    p = make_problem(B, Tmax, m, k, x_layout=_lib.X_CHANNEL_MAJOR, ldx=Tmax, x_batch_stride=1,
                     w_layout=_lib.W_COMPONENT_MAJOR, update_H=update_H, max_iter=max_iter, tol=tol,
                     check_every=check_every, l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H, l2_reg_W=l2_reg_W, l2_reg_H=l2_reg_H,
                     loss=beta_loss_code(beta_loss))
    err = torch.empty((B,), dtype=dtype, device=dev)
    n_iter = torch.empty((B,), dtype=torch.int32, device=dev)
    sse = torch.empty((B, m), dtype=dtype, device=dev)
    xsq = torch.empty((B, m), dtype=dtype, device=dev)
    h = handle if handle is not None else _lib.get_handle(dev.index)
    lib = _lib.load()
    fn = lib.hipnmf_fit_ragged_f32 if dtype == torch.float32 else lib.hipnmf_fit_ragged_f64
    torch.cuda.synchronize(dev)
    _lib.check(fn(h.ptr, ctypes.byref(p), desc.ctypes.data_as(ctypes.c_void_p), Xp.data_ptr(), Wp.data_ptr(),
                  Hp.data_ptr(), err.data_ptr(), n_iter.data_ptr(), sse.data_ptr(), xsq.data_ptr()))
    Ws = [Wp[w_off[b]: w_off[b] + k * lds[b]].view(k, lds[b])[:, : Ts[b]].t().contiguous() for b in range(B)]
    vaf = torch.cat([(1 - sse.sum(dim=1) / xsq.sum(dim=1)).unsqueeze(1), 1 - sse / xsq], dim=1)
    return BatchedResult(Ws, Hp, n_iter, err, vaf, sse, xsq, h.last_kernel_ms())


# ------------------------------------------------------------------------------------------------
# Multi-restart fits: the standard remedy for the local minima of the
# multiplicative updates -- R random starts per trial, keep the one with the smallest residual.
@dataclass
class RestartResult:
    """Best-of-R fit per trial: ``best`` holds the winning restart's factors (a :class:`BatchedResult` with
    ``W [B, T, k]``), ``restart_err [B, R]`` the final residual of every restart and ``chosen [B]`` the index
    of the winner."""

    best: BatchedResult
    restart_err: "object"
    chosen: "object"
    kernel_ms: float


def fit_restarts(X, k: int, n_restarts: int = 8, *, seed: int = 0, max_iter: int = 200, tol: float = 1e-4,
                 check_every: int = 10, beta_loss="frobenius", l1_reg_W: float = 0.0, l1_reg_H: float = 0.0,
                 l2_reg_W: float = 0.0, l2_reg_H: float = 0.0, device=None, devices=None) -> RestartResult:
    """``n_restarts`` random-init factorisations (sklearn's ``init='random'`` law, drawn on the device) of every
    matrix of ``X [B, T, m]`` in ONE launch, and the best of them per matrix by final residual.

    The ``B * R`` factorisations are independent units for the engine (one workgroup each); the restarts of a
    trial read the same copy of its X through the per-matrix descriptors of ``hipnmf_fit_ragged_*``, so X is
    neither duplicated in HBM nor re-uploaded.

    ``devices=``: trials scattered over several GPUs (all restarts of a trial stay together), results on the host; the slice
    that starts at trial ``lo`` draws with ``seed + lo``.
    """
    torch = _torch()
    if devices is not None:
        from .multi_gpu import cat_host, resolve_devices, scatter, to_host

        Xa = X if isinstance(X, torch.Tensor) else np.asarray(X)
        if Xa.ndim == 2:
            Xa = Xa[None]
        kw = dict(max_iter=max_iter, tol=tol, check_every=check_every, beta_loss=beta_loss, l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H,
                  l2_reg_W=l2_reg_W, l2_reg_H=l2_reg_H)
        parts = [r for _, _, _, r in scatter(Xa.shape[0], resolve_devices(devices),
                                             lambda lo, hi, d: fit_restarts(Xa[lo:hi], k, n_restarts, seed=seed + lo, device=f"cuda:{d}", **kw))]
        best = _gather_batched([r.best for r in parts], False)
        return RestartResult(best, cat_host([to_host(r.restart_err, False) for r in parts]),
                             cat_host([to_host(r.chosen, False) for r in parts]), max(r.kernel_ms for r in parts))
    dev = resolve_device(device)
    Xt = _as_device_tensor(X, dev)
    if Xt.dim() == 2:
        Xt = Xt.unsqueeze(0)
    if Xt.dim() != 3:
        raise ValueError(f"X must be [B, T, m] or [T, m], got shape {tuple(Xt.shape)}")
    if Xt.dtype not in (torch.float32, torch.float64):
        Xt = Xt.to(torch.float64)
    B, T, m = Xt.shape
    R = int(n_restarts)
    if R < 1:
        raise ValueError("n_restarts must be >= 1")
    if not 1 <= k <= m:
        raise ValueError(f"k must be in [1, {m}]")
    ld = (T + 3) // 4 * 4
    # channel-major packed copy of X with a leading dimension that is a multiple of 4 (in place when it already is)
    if Xt.stride(1) == 1 and Xt.stride(2) == ld and Xt.stride(0) == m * ld and Xt.data_ptr() % 16 == 0:
        Xp = Xt
    else:
        Xp = torch.zeros((B, m, ld), dtype=Xt.dtype, device=dev)
        Xp[:, :, :T] = Xt.transpose(1, 2)
        Xp = Xp.transpose(1, 2)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    avg = torch.sqrt(Xt.mean(dim=(1, 2)) / k).view(B, 1, 1, 1)
    H = (avg * torch.randn((B, R, k, m), generator=g, device=dev, dtype=Xt.dtype).abs_()).view(B * R, k, m)
    W = torch.zeros((B, R, k, ld), dtype=Xt.dtype, device=dev)
    W[..., :T] = avg * torch.randn((B, R, k, T), generator=g, device=dev, dtype=Xt.dtype).abs_()
    desc = np.empty((B * R, 4), dtype=np.int64)
    bb, rr = np.divmod(np.arange(B * R), R)
    desc[:, 0], desc[:, 1], desc[:, 2], desc[:, 3] = T, bb * (m * ld), ld, (bb * R + rr) * (k * ld)
    p = make_problem(B * R, T, m, k, x_layout=_lib.X_CHANNEL_MAJOR, ldx=T, x_batch_stride=1,
                     w_layout=_lib.W_COMPONENT_MAJOR, max_iter=max_iter, tol=tol, check_every=check_every,
                     l1_reg_W=l1_reg_W, l1_reg_H=l1_reg_H, l2_reg_W=l2_reg_W, l2_reg_H=l2_reg_H,
                     loss=beta_loss_code(beta_loss))
    err = torch.empty((B * R,), dtype=Xt.dtype, device=dev)
    n_iter = torch.empty((B * R,), dtype=torch.int32, device=dev)
    sse = torch.empty((B * R, m), dtype=Xt.dtype, device=dev)
    xsq = torch.empty((B * R, m), dtype=Xt.dtype, device=dev)
    h = _lib.get_handle(dev.index)
    lib = _lib.load()
    fn = lib.hipnmf_fit_ragged_f32 if Xt.dtype == torch.float32 else lib.hipnmf_fit_ragged_f64
    torch.cuda.synchronize(dev)
    x_base = Xp.data_ptr() if Xp is Xt else Xp.transpose(1, 2).data_ptr()
    _lib.check(fn(h.ptr, ctypes.byref(p), desc.ctypes.data_as(ctypes.c_void_p), x_base, W.data_ptr(), H.data_ptr(),
                  err.data_ptr(), n_iter.data_ptr(), sse.data_ptr(), xsq.data_ptr()))
    ms = h.last_kernel_ms()
    err2 = err.view(B, R)
    chosen = torch.argmin(err2, dim=1)
    flat = torch.arange(B, device=dev) * R + chosen
    Wb = W.view(B * R, k, ld)[flat][:, :, :T].transpose(1, 2).contiguous()
    sse_b, xsq_b = sse[flat], xsq[flat]
    vaf = torch.cat([(1 - sse_b.sum(dim=1) / xsq_b.sum(dim=1)).unsqueeze(1), 1 - sse_b / xsq_b], dim=1)
    best = BatchedResult(Wb, H[flat], n_iter[flat], err[flat], vaf, sse_b, xsq_b, ms)
    return RestartResult(best, err2, chosen, ms)
