"""ctypes binding of ``libhip_nmf.so`` (the C ABI declared in ``include/hip_nmf.h``).

The library is loaded lazily and exactly once.  There is no fallback: if the shared object is missing
or no GPU is visible, compute calls raise :class:`HipNmfError` -- nothing here ever routes to a CPU
implementation.

``torch`` is imported *before* the library on purpose: PyTorch-ROCm ships its own ``libamdhip64.so.7``;
loading it first makes the dynamic linker resolve our ``DT_NEEDED libamdhip64.so.7`` to the copy that is
already mapped, so torch's device pointers and streams and our kernels live in one HIP runtime.
"""

from __future__ import annotations

import atexit
import ctypes
import os
import threading

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libhip_nmf.so")

HIPNMF_OK = 0
HIPNMF_ERR_BAD_ARG = -1
HIPNMF_ERR_HIP = -2
HIPNMF_ERR_UNSUPPORTED = -3
HIPNMF_ERR_NO_DEVICE = -4

LOSS_FROBENIUS = 0
LOSS_KL = 1
X_ROW_MAJOR = 0
X_CHANNEL_MAJOR = 1
W_ROW_MAJOR = 0
W_COMPONENT_MAJOR = 1
W_ROW_MAJOR_PAD16 = 2  # [T][round_up(k, 16)], zero padding: general-shape shard entry points only

#: every symbol ``include/hip_nmf.h`` declares (checked by tests/test_abi.py)
EXPORTS = (
    "hipnmf_version", "hipnmf_last_error", "hipnmf_device_count", "hipnmf_create", "hipnmf_destroy",
    "hipnmf_set_stream", "hipnmf_workspace_bytes", "hipnmf_last_kernel_ms", "hipnmf_last_kernel", "hipnmf_set_async",
    "hipnmf_set_tuning", "hipnmf_set_batch_hint", "hipnmf_routes_describe",
    "hipnmf_fit_batched_f32", "hipnmf_fit_batched_f64", "hipnmf_fit_ragged_f32", "hipnmf_fit_ragged_f64",
    "hipnmf_shard_pass_f32", "hipnmf_shard_hupdate_f32", "hipnmf_shard_residual_f32",
    "hipnmf_shard_pass_f64", "hipnmf_shard_hupdate_f64", "hipnmf_shard_residual_f64",
    "hipnmf_fit_tsharded_f32", "hipnmf_fit_tsharded_f64",
    "hipnmf_random_init_f32", "hipnmf_random_init_f64", "hipnmf_rank_sweep_f32", "hipnmf_rank_sweep_f64",
    "hipnmf_rank_sweep_stop_f32", "hipnmf_rank_sweep_stop_f64",
    "hipnmf_emg_envelope_f32", "hipnmf_emg_envelope_f64", "hipnmf_resample_weights_f32", "hipnmf_resample_weights_f64", "hipnmf_sosfilt_f32", "hipnmf_sosfilt_f64",
    "hipnmf_gram_f32", "hipnmf_gram_f64", "hipnmf_nndsvd_stats_f32", "hipnmf_nndsvd_stats_f64",
    "hipnmf_nndsvd_write_f32", "hipnmf_nndsvd_write_f64", "hipnmf_diag_stream_gbs",
)


# int (*hipnmf_allreduce_fn)(void* device_buf, size_t count, int elem_size, void* hip_stream, void* user)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)


class HipNmfError(RuntimeError):
    """Error reported by libhip_nmf (carries the C error code in ``.code``)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libhip_nmf error {code}: {message}")
        self.code = code


class Problem(ctypes.Structure):
    """Mirror of ``struct hipnmf_problem`` (include/hip_nmf.h)."""

    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("batch", ctypes.c_int32),
        ("n_samples", ctypes.c_int64),
        ("n_features", ctypes.c_int32),
        ("n_components", ctypes.c_int32),
        ("x_layout", ctypes.c_int32),
        ("update_h", ctypes.c_int32),
        ("w_layout", ctypes.c_int32),
        ("loss", ctypes.c_int32),
        ("ldx", ctypes.c_int64),
        ("x_batch_stride", ctypes.c_int64),
        ("max_iter", ctypes.c_int32),
        ("check_every", ctypes.c_int32),
        ("tol", ctypes.c_double),
        ("l1_reg_W", ctypes.c_double),
        ("l1_reg_H", ctypes.c_double),
        ("l2_reg_W", ctypes.c_double),
        ("l2_reg_H", ctypes.c_double),
    ]


_lock = threading.Lock()
_lib = None


class _Gate:
    """Counts the native calls in flight and, once :func:`shutdown` has begun, parks every OTHER thread that tries to start one.

    Why: a thread that is inside a HIP call while the main thread runs the process's exit handlers takes the process down
    (``terminate called without an active exception`` -> SIGABRT, 1 run in 8 with a daemon thread fitting in a loop:
    profiles/r06_abort_hunt.md) -- the HIP runtime tears itself down under it.  The interpreter joins its non-daemon threads before
    ``atexit`` runs but knows nothing about daemon threads; with the gate the exit hook lets the call that is in flight return
    (milliseconds), keeps the thread from starting another one, and only then destroys the handles."""

    def __init__(self):
        self.cond = threading.Condition(threading.Lock())
        self.inflight = 0
        self.closing = False
        self.closer = None
        self.parked: set = set()

    def enter(self):
        with self.cond:
            if self.closing and threading.get_ident() != self.closer:
                park = True
                self.parked.add(threading.get_ident())
                self.cond.notify_all()
            else:
                park = False
                self.inflight += 1
        if park:  # the process is exiting: this (daemon) thread must not touch the runtime again
            import time

            while True:
                time.sleep(3600)

    def leave(self):
        with self.cond:
            self.inflight -= 1
            if self.inflight == 0:
                self.cond.notify_all()

    def close(self, timeout: float) -> bool:
        """Begin the shutdown; True when no native call of another thread is left in flight after at most ``timeout`` s."""
        with self.cond:
            self.closing = True
            self.closer = threading.get_ident()
            return self.cond.wait_for(lambda: self.inflight == 0, timeout)

    def wait_parked(self, idents, timeout: float) -> bool:
        """Wait until every thread of ``idents`` (threads that own a handle and are still running: daemon threads) has arrived
        at the gate.  Between two native calls such a thread runs Python / torch code; CPython 3.10 ends a daemon thread that
        wants the GIL during finalisation with pthread_exit, and unwinding through torch's C++ frames is ``terminate called
        without an active exception`` (native backtrace: profiles/r06_abort_hunt.md).  A thread that fits in a loop reaches
        the gate within one call; one that never calls again costs the exit this timeout."""
        with self.cond:
            return self.cond.wait_for(lambda: set(idents) <= self.parked, timeout)


_gate = _Gate()


class _Fn:
    """A ctypes function behind the gate; ``argtypes`` / ``restype`` read and written through."""

    __slots__ = ("_f",)

    def __init__(self, f):
        object.__setattr__(self, "_f", f)

    def __call__(self, *args):
        _gate.enter()
        try:
            return self._f(*args)
        finally:
            _gate.leave()

    def __getattr__(self, name):
        return getattr(self._f, name)

    def __setattr__(self, name, value):
        setattr(self._f, name, value)


class _GatedLib:
    """``ctypes.CDLL`` whose functions are :class:`_Fn` objects (what :func:`load` returns)."""

    def __init__(self, cdll):
        self._cdll = cdll
        self._fns: dict = {}

    def __getattr__(self, name):
        fns = self.__dict__["_fns"]
        f = fns.get(name)
        if f is None:
            f = fns[name] = _Fn(getattr(self.__dict__["_cdll"], name))
        return f


def _declare(lib):
    vp, ip = ctypes.c_void_p, ctypes.c_int
    pp = ctypes.POINTER(Problem)
    lib.hipnmf_version.restype = ip
    lib.hipnmf_version.argtypes = []
    lib.hipnmf_last_error.restype = ctypes.c_char_p
    lib.hipnmf_last_error.argtypes = []
    lib.hipnmf_device_count.restype = ip
    lib.hipnmf_device_count.argtypes = []
    lib.hipnmf_create.restype = ip
    lib.hipnmf_create.argtypes = [ip, ctypes.POINTER(vp)]
    lib.hipnmf_destroy.restype = ip
    lib.hipnmf_destroy.argtypes = [vp]
    lib.hipnmf_set_stream.restype = ip
    lib.hipnmf_set_stream.argtypes = [vp, vp]
    lib.hipnmf_workspace_bytes.restype = ctypes.c_size_t
    lib.hipnmf_workspace_bytes.argtypes = [pp, ip]
    lib.hipnmf_last_kernel_ms.restype = ip
    lib.hipnmf_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.hipnmf_last_kernel.restype = ctypes.c_char_p
    lib.hipnmf_last_kernel.argtypes = [vp]
    lib.hipnmf_set_async.restype = ip
    lib.hipnmf_set_async.argtypes = [vp, ip]
    lib.hipnmf_set_tuning.restype = ip
    lib.hipnmf_set_tuning.argtypes = [vp, ip, ip, ip]
    lib.hipnmf_set_batch_hint.restype = ip
    lib.hipnmf_set_batch_hint.argtypes = [vp, ip]
    lib.hipnmf_routes_describe.restype = ctypes.c_char_p
    lib.hipnmf_routes_describe.argtypes = []
    lib.hipnmf_diag_stream_gbs.restype = ip
    lib.hipnmf_diag_stream_gbs.argtypes = [vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]
    for sfx in ("f32", "f64"):
        f = getattr(lib, f"hipnmf_fit_batched_{sfx}")
        f.restype = ip
        f.argtypes = [vp, pp, vp, vp, vp, vp, vp, vp, vp]
        f = getattr(lib, f"hipnmf_fit_ragged_{sfx}")
        f.restype = ip
        f.argtypes = [vp, pp, vp, vp, vp, vp, vp, vp, vp, vp]
        f = getattr(lib, f"hipnmf_shard_pass_{sfx}")
        f.restype = ip
        f.argtypes = [vp, pp, vp, vp, vp, vp]
        f = getattr(lib, f"hipnmf_shard_hupdate_{sfx}")
        f.restype = ip
        f.argtypes = [vp, pp, vp, vp]
        f = getattr(lib, f"hipnmf_shard_residual_{sfx}")
        f.restype = ip
        f.argtypes = [vp, pp, vp, vp, vp, vp, vp]
        f = getattr(lib, f"hipnmf_emg_envelope_{sfx}")
        f.restype = ip
        f.argtypes = [vp, vp, vp, vp]
        f = getattr(lib, f"hipnmf_resample_weights_{sfx}")
        f.restype = ip
        f.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int32, vp]
        f = getattr(lib, f"hipnmf_sosfilt_{sfx}")
        f.restype = ip
        f.argtypes = [vp, vp, vp, vp, vp, vp]
        f = getattr(lib, f"hipnmf_gram_{sfx}")
        f.restype = ip
        f.argtypes = [vp, pp, vp, vp, vp]
        f = getattr(lib, f"hipnmf_nndsvd_stats_{sfx}")
        f.restype = ip
        f.argtypes = [vp, pp, vp, vp, vp, vp]
        f = getattr(lib, f"hipnmf_nndsvd_write_{sfx}")
        f.restype = ip
        f.argtypes = [vp, pp, vp, vp, vp, vp, vp, ctypes.c_double, vp]


def load():
    """Return the loaded ``ctypes.CDLL`` (loading it on first use)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = os.environ.get("HIPNMF_LIBRARY", LIB_PATH)
        if not os.path.exists(path):
            raise HipNmfError(
                HIPNMF_ERR_NO_DEVICE,
                f"{path} not found: build the HIP extension first (python -m muscle_synergies_amd.build). "
                "There is no CPU fallback for solver='mu'.",
            )
        import torch  # noqa: F401  -- must precede the CDLL (shared HIP runtime, see module docstring)

        lib = ctypes.CDLL(path)
        _declare(lib)
        _lib = _GatedLib(lib)
        # registered AFTER torch's import, so (atexit is last-in-first-out) it runs BEFORE any exit handler of torch: every
        # handle this process still caches is destroyed while the HIP runtime torch shares with us is fully alive
        atexit.register(shutdown)
    return _lib


def check(code: int):
    if code != HIPNMF_OK:
        msg = load().hipnmf_last_error()
        raise HipNmfError(code, msg.decode("utf-8", "replace") if msg else "unknown error")


class Handle:
    """RAII wrapper of ``hipnmf_handle`` (one per device and host thread)."""

    def __init__(self, device: int = 0):
        lib = load()
        self._h = ctypes.c_void_p()
        if _closed:
            raise HipNmfError(HIPNMF_ERR_HIP, "the interpreter is shutting down: libhip_nmf's handles are closed")
        check(lib.hipnmf_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)
        self._owner = threading.get_ident()  # the thread that created (and, for cached handles, alone drives) this handle
        _live.add(self)

    @property
    def ptr(self):
        return self._h

    def set_tuning(self, threads: int = 0, max_slices: int = 0, variant: int = 0):
        check(load().hipnmf_set_tuning(self._h, int(threads), int(max_slices), int(variant)))

    def set_batch_hint(self, batch: int = 0):
        """``hipnmf_set_batch_hint``: route the following fits as for a batch of this many matrices (0 = off)."""
        check(load().hipnmf_set_batch_hint(self._h, int(batch)))

    def set_stream(self, stream_ptr):
        """``stream_ptr``: a ``hipStream_t`` as int; 0 = the device's default (null) stream; None = the
        handle's own stream."""
        if stream_ptr is None:
            arg = None
        else:
            arg = ctypes.c_void_p(int(stream_ptr) if stream_ptr else 1)  # HIPNMF_STREAM_NULL
        check(load().hipnmf_set_stream(self._h, arg))

    def set_async(self, enable: bool):
        check(load().hipnmf_set_async(self._h, int(bool(enable))))

    def last_kernel_ms(self) -> float:
        ms = ctypes.c_float()
        check(load().hipnmf_last_kernel_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def stream_gbs(self, region_bytes: int, regions: int, passes: int = 20) -> float:
        """``hipnmf_diag_stream_gbs``: the memory system's rate for the batched solver's access pattern, measured now."""
        out = ctypes.c_double()
        check(load().hipnmf_diag_stream_gbs(self._h, int(region_bytes), int(regions), int(passes), ctypes.byref(out)))
        return float(out.value)

    def last_kernel(self) -> str:
        """Instance name of the solver kernel the last fit launched (``hipnmf_last_kernel``)."""
        name = load().hipnmf_last_kernel(self._h)
        return name.decode() if name else ""

    def close(self):
        """Destroy the native handle (stream, events, workspaces).  After :func:`shutdown` (interpreter exit) this is a no-op:
        a handle that is only collected during module teardown must not call into a HIP runtime that may be unloading."""
        h = getattr(self, "_h", None)
        if _gate.closing and threading.get_ident() != _gate.closer:
            return  # the exit hook owns the teardown now (a destructor must never park its thread at the gate)
        if h is not None and h and not _closed:
            self._h = ctypes.c_void_p()
            _live.discard(self)
            load().hipnmf_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_handles: dict = {}
_handles_lock = threading.Lock()
_closed = False  # set by shutdown(): no native call is made after it
import weakref  # noqa: E402

_live: "weakref.WeakSet" = weakref.WeakSet()  # every open Handle of the process (cached or not)
_shutdown_hooks: list = []  # callables run first by shutdown() (analysis.py: its rank-range worker pool)


def shutdown() -> None:
    """Orderly teardown, run by ``atexit`` (and callable by a host that unloads the engine earlier): stop the package's worker
    threads, then destroy EVERY open handle of the process -- the per-thread cache, including those of threads that are
    gone, and handles user code still holds -- while the HIP runtime is still loaded.  Afterwards ``Handle.close`` /
    ``__del__`` do nothing.  (handles used to be destroyed from ``__del__`` during module teardown, in no
    defined order relative to torch's own exit handlers and to worker threads that were still alive.)"""
    global _closed
    for hook in list(_shutdown_hooks):
        try:
            hook()
        except Exception:  # noqa: BLE001
            pass
    # no other thread starts a native call from here on; the ones in flight (a daemon thread inside a fit) get 10 s to return
    drained = _gate.close(10.0)
    me0 = threading.get_ident()
    owners = {getattr(h, "_owner", me0) for h in list(_live)}
    still = {t.ident for t in threading.enumerate() if t.is_alive() and t.ident != me0 and t.ident in owners}
    if still:
        _gate.wait_parked(still, 2.0)
    # (a thread that slipped past the check in get_handle and parked inside hipnmf_create holds this lock for ever: do not wait for it)
    locked = _handles_lock.acquire(timeout=2.0)
    try:
        cached = list(_handles.values())
        _handles.clear()
        others = [h for h in list(_live) if h not in cached]
    finally:
        if locked:
            _handles_lock.release()
    # not drained (a call that does not come back): a handle whose owner thread is still running may be in use this very moment --
    # it is left alone, its device memory goes with the process, and it is never touched again
    me = threading.get_ident()
    running = set() if drained else {t.ident for t in threading.enumerate() if t.is_alive()} - {me}
    for h in cached + others:
        if getattr(h, "_owner", me) in running:
            continue
        try:
            h.close()
        except Exception:  # noqa: BLE001
            pass
    _closed = True


def get_handle(device: int = 0) -> Handle:
    """Per-(thread, device) cached handle."""
    key = (threading.get_ident(), int(device))
    if _gate.closing and threading.get_ident() != _gate.closer:
        _gate.enter()  # parks this thread for good -- BEFORE it takes the lock the exit hook needs (never while holding it)
    with _handles_lock:
        h = _handles.get(key)
        if h is None:
            h = Handle(device)
            _handles[key] = h
        return h


def release_thread_handles() -> None:
    """Destroy the calling thread's cached handles (their streams and workspaces): worker threads call this when done."""
    me = threading.get_ident()
    with _handles_lock:
        mine = [k for k in _handles if k[0] == me]
        hs = [_handles.pop(k) for k in mine]
    for h in hs:
        h.close()


def routes() -> dict:
    """The routing constants in force (``hipnmf_routes_describe``): defaults of ``hipnmf_route_table`` + ``HIPNMF_ROUTES``."""
    text = load().hipnmf_routes_describe().decode()
    return {k: float(v) for k, v in (item.split("=") for item in text.split(","))}


def device_count() -> int:
    n = load().hipnmf_device_count()
    if n < 0:
        check(n)
    return n
