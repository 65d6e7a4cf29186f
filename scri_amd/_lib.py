"""ctypes binding of libscri_amd.so (C ABI declared in include/scri_amd.h).

There is no CPU fallback: importing this module needs the built shared library, and creating a
context needs an MI355X (gfx950).  Both failures raise immediately with the library's message.
"""
import collections
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCRI_AMD_LIB_PATH") or os.path.join(_HERE, "libscri_amd.so")  # (the override: A/B builds of one kernel)

BMS_HOST, BMS_DEVICE = 0, 1
BMS_TERM_NONE, BMS_TERM_H, BMS_TERM_SIGMA, BMS_TERM_PSI = 0, 1, 2, 3
BMS_ERR_INVALID, BMS_ERR_HIP, BMS_ERR_NOMEM, BMS_ERR_UNSUPPORTED, BMS_ERR_NODEVICE, BMS_ERR_INTERNAL = -1, -2, -3, -4, -5, -6

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_vp = ctypes.c_void_p
c_dp = ctypes.POINTER(ctypes.c_double)


class bms_wm_input(ctypes.Structure):
    _fields_ = [
        ("n_times", c_i64),
        ("t", c_dp),
        ("data", c_vp),
        ("ld", c_i64),
        ("mem", c_int),
        ("ell_min", c_int),
        ("ell_max", c_int),
        ("spin_weight", c_int),
        ("conformal_weight", c_int),
        ("type_term", c_int),
        ("n_aux", c_int),
        ("aux_data", c_vp * 4),
        ("aux_ld", c_i64 * 4),
        ("aux_ell_min", c_int * 4),
        ("aux_ell_max", c_int * 4),
        ("aux_spin", c_int * 4),
        ("aux_coeff", ctypes.c_double * 4),
        ("aux_power", c_int * 4),
    ]


class bms_transformation(ctypes.Structure):
    _fields_ = [
        ("supertranslation", c_vp),
        ("ell_max_supertranslation", c_int),
        ("frame_rotation", ctypes.c_double * 4),
        ("boost_velocity", ctypes.c_double * 3),
        ("n_theta", c_int),
        ("n_phi", c_int),
        ("ell_max_out", c_int),
    ]


class bms_shard(ctypes.Structure):
    _fields_ = [("data_row0", c_i64), ("data_rows", c_i64), ("out_i0", c_i64), ("out_i1", c_i64), ("col_part", ctypes.c_int32),
                ("col_parts", ctypes.c_int32)]


KERNEL_TAGS = ("rotate", "setup", "gemm_synthesis", "spline_forward", "spline_backward", "gemm_analysis", "pointwise", "theta_quadrature", "analysis_fused", "analysis_large")

# every symbol include/scri_amd.h declares: (restype, argtypes)
SIGNATURES = {
    "bms_version": (c_int, []),
    "bms_ctx_create": (c_int, [c_int, ctypes.POINTER(c_vp)]),
    "bms_ctx_destroy": (None, [c_vp]),
    "bms_last_error": (ctypes.c_char_p, [c_vp]),
    "bms_ctx_set_stream": (c_int, [c_vp, c_vp]),
    "bms_ctx_use_default_stream": (c_int, [c_vp]),
    "bms_ctx_set_option": (c_int, [c_vp, ctypes.c_char_p, c_i64]),
    "bms_ctx_get_option": (c_int, [c_vp, ctypes.c_char_p, ctypes.POINTER(c_i64)]),
    "bms_ctx_set_workspace_limit": (c_int, [c_vp, ctypes.c_uint64]),
    "bms_ctx_reserve": (c_int, [c_vp, ctypes.c_uint64]),
    "bms_ctx_get_eval_stats": (c_int, [c_vp, ctypes.POINTER(c_i64), c_int]),
    "bms_ctx_synchronize": (c_int, [c_vp]),
    "bms_ctx_enable_timing": (c_int, [c_vp, c_int]),
    "bms_ctx_get_timing": (c_int, [c_vp, c_dp, ctypes.POINTER(c_i64), c_int]),
    "bms_shard_plan": (
        c_int,
        [c_vp, c_dp, c_i64, ctypes.POINTER(bms_transformation), c_i64, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)],
    ),
    "bms_host_alloc": (c_vp, [ctypes.c_uint64]),
    "bms_host_free": (None, [c_vp]),
    "bms_host_register": (c_int, [c_vp, ctypes.c_uint64]),
    "bms_host_unregister": (c_int, [c_vp]),
    "bms_modes_to_grid": (
        c_int,
        [c_vp, ctypes.POINTER(bms_wm_input), ctypes.POINTER(bms_transformation), c_dp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_output_window": (c_int, [c_vp, c_dp, c_i64, ctypes.POINTER(bms_transformation), c_int, ctypes.POINTER(c_i64)]),
    "bms_transform_modes_shard": (
        c_int,
        [c_vp, ctypes.POINTER(bms_wm_input), ctypes.POINTER(bms_transformation), ctypes.POINTER(bms_shard), c_dp, c_vp,
         ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)],
    ),
    "bms_transform_modes_pipelined": (
        c_int,
        [c_vp, ctypes.POINTER(bms_wm_input), ctypes.POINTER(bms_transformation), c_int, c_dp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_transform_abd_pipelined": (
        c_int,
        [c_vp, c_dp, c_vp, c_i64, c_int, ctypes.POINTER(bms_transformation), c_int, c_dp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_transform_modes_series": (
        c_int,
        [c_vp, ctypes.POINTER(bms_wm_input), c_int, ctypes.POINTER(bms_transformation), c_dp, c_vp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_transform_modes_pipelined_part": (
        c_int,
        [c_vp, ctypes.POINTER(bms_wm_input), ctypes.POINTER(bms_transformation), c_int, c_int, c_int, c_dp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_transform_abd_pipelined_part": (
        c_int,
        [c_vp, c_dp, c_vp, c_i64, c_int, ctypes.POINTER(bms_transformation), c_int, c_int, c_int, c_dp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_transform_modes_multi": (
        c_int,
        [ctypes.POINTER(c_vp), c_int, ctypes.POINTER(bms_wm_input), ctypes.POINTER(bms_transformation), c_int, c_dp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_transform_abd_multi": (
        c_int,
        [ctypes.POINTER(c_vp), c_int, c_dp, c_vp, c_i64, c_int, ctypes.POINTER(bms_transformation), c_int, c_dp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_rotate_const": (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_int, c_int, c_dp]),
    "bms_rotate_series": (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_int, c_int, c_vp]),
    "bms_rotate_const_D": (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_int, c_int, c_vp]),
    "bms_wigner_D": (c_int, [c_vp, c_dp, c_int, c_int, c_vp]),
    "bms_transform_modes": (
        c_int,
        [c_vp, ctypes.POINTER(bms_wm_input), ctypes.POINTER(bms_transformation), c_dp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_transform_abd": (
        c_int,
        [c_vp, c_dp, c_vp, c_int, c_i64, c_int, ctypes.POINTER(bms_transformation), c_dp, c_vp, ctypes.POINTER(c_i64)],
    ),
    "bms_transform_abd_shard": (
        c_int,
        [c_vp, c_dp, c_vp, c_int, c_i64, c_int, ctypes.POINTER(bms_transformation), ctypes.POINTER(bms_shard), c_dp, c_vp,
         ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)],
    ),
    "bms_rotor_grid": (c_int, [c_vp, c_dp, c_dp, c_int, c_int, c_dp]),
    "bms_ring_colatitudes": (c_int, [c_dp, c_dp, c_int, c_int, c_dp]),
    "bms_conformal_factors": (c_int, [c_vp, c_dp, c_dp, c_i64, c_dp, c_vp, c_dp, c_dp]),
    "bms_swsh_grid": (c_int, [c_vp, c_dp, c_i64, c_int, c_int, c_int, c_vp]),
    "bms_map2salm": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "bms_cubic_spline": (c_int, [c_vp, c_dp, c_i64, c_vp, c_i64, c_i64, c_int, c_dp, c_i64, c_vp]),
    "bms_spline_derivative": (c_int, [c_vp, c_dp, c_i64, c_vp, c_i64, c_i64, c_int, c_dp, c_i64, c_int, c_vp]),
    "bms_angular_velocity": (c_int, [c_vp, c_dp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_dp, c_dp, c_dp]),
    "bms_integrate_angular_velocity": (c_int, [c_vp, c_dp, c_i64, c_dp, c_dp, ctypes.c_double, c_dp]),
    "bms_xor_timeseries": (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_int]),
    "bms_multishuffle": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, ctypes.POINTER(c_int), c_int, c_int]),
    "bms_fletcher32": (c_int, [c_vp, c_vp, c_int, c_i64, ctypes.POINTER(ctypes.c_uint32)]),
    "bms_salm2map": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int, c_int, c_vp]),
    "bms_evaluate_modes": (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_int, c_int, c_int, c_dp, c_i64, c_vp]),
    "bms_mode_map": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_i64, ctypes.POINTER(ctypes.c_int32), c_vp, c_int, c_vp, c_i64,
                              ctypes.POINTER(ctypes.c_int32), c_vp, c_int, c_vp, c_int]),
    "bms_row_norm": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp]),
    "bms_grid_multiply": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_i64, c_int, c_int, c_vp]),
}

_lib = None
_lock = threading.Lock()


class BMSError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so; if this library pulled in
    the system copy first and torch were imported afterwards, the process would hold two runtimes and the second one to
    initialise sees no GPU ("No HIP GPUs are available").  Loading torch's copy first (same soname) makes the dynamic
    linker resolve this library's dependency to it, whichever of the two packages the user touches first.  Without a
    torch installation the system runtime is used."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for root in (spec.submodule_search_locations if spec and spec.submodule_search_locations else ()):
        path = os.path.join(root, "lib", "libamdhip64.so")
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load():
    """Load libscri_amd.so (once).  Raises ImportError with build instructions if it is missing."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                    "(or `python -c 'import __graft_entry__ as g; g.build()'`).  scri_amd has no CPU fallback."
                )
            _share_hip_runtime_with_torch()
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


class _PinnedBlock:
    """A page-locked host allocation that numpy arrays can sit on (their .base keeps it alive).  Freed blocks of the sizes
    in recent use are kept for the next result of that size: pinning hundreds of MB costs as much as copying them.
    The pool holds at most `pool_limit()` bytes -- SCRI_AMD_PINNED_POOL_BYTES, default twice the largest block of the last
    few requests (so a workflow that alternates two result sizes keeps both) -- and gives back the least recently used
    blocks beyond that; `trim()` (also called by Context.close) releases everything."""

    _pool = collections.OrderedDict()  # nbytes -> [ptr, ...], least recently used size first
    _pool_lock = threading.RLock()  # re-entrant: a collection inside the critical section may finalise another block
    _pooled_bytes = 0
    _recent = collections.deque(maxlen=8)  # sizes of the last requests

    @classmethod
    def pool_limit(cls):
        env = os.environ.get("SCRI_AMD_PINNED_POOL_BYTES")
        if env is not None:
            return int(env)
        return 2 * max(cls._recent, default=0)

    @classmethod
    def _evict(cls, limit):
        """Release least recently used blocks until the pool holds at most `limit` bytes (lock held by the caller)."""
        while cls._pooled_bytes > limit and cls._pool:
            size, ptrs = next(iter(cls._pool.items()))
            ptr = ptrs.pop()
            if not ptrs:
                del cls._pool[size]
            cls._pooled_bytes -= size
            load().bms_host_free(ptr)

    @classmethod
    def trim(cls):
        with cls._pool_lock:
            cls._evict(0)

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        with _PinnedBlock._pool_lock:
            _PinnedBlock._recent.append(self.nbytes)
            free = _PinnedBlock._pool.get(self.nbytes)
            self.ptr = free.pop() if free else None
            if self.ptr is not None:
                _PinnedBlock._pooled_bytes -= self.nbytes
                if not free:
                    del _PinnedBlock._pool[self.nbytes]
        if self.ptr is None:
            self.ptr = load().bms_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("page-locked allocation failed")
        self.__array_interface__ = {"data": (int(self.ptr), False), "shape": (self.nbytes,), "typestr": "|u1", "version": 3}

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if not ptr:
            return
        try:
            with _PinnedBlock._pool_lock:
                limit = _PinnedBlock.pool_limit()
                if self.nbytes <= limit:
                    _PinnedBlock._pool.setdefault(self.nbytes, []).append(ptr)
                    _PinnedBlock._pool.move_to_end(self.nbytes)
                    _PinnedBlock._pooled_bytes += self.nbytes
                    _PinnedBlock._evict(limit)
                    return
            load().bms_host_free(ptr)
        except Exception:  # interpreter shutdown
            pass


# ---- page-locking of caller-owned input arrays that are transformed repeatedly
# An input is "the same array again" only if it is the same live OBJECT (identity of the array that owns the memory, held by a weak
# reference) with the same address and size: a temporary whose freed block malloc hands out again at the same address is a new
# object and starts from zero, so one-off arrays (copies made by the shim, the fresh copy an adapter builds per call) are never
# page-locked, and a range that moved (ndarray.resize) is released before anything else happens.
_reg_lock = threading.RLock()  # re-entrant: a finalizer (_release) can fire on this thread while the lock is held (cyclic GC during an allocation)
_seen_inputs = collections.OrderedDict()  # id(owner) -> [weakref to owner, address, nbytes, sightings] (small LRU)
_registered = {}  # id(owner) -> (address, nbytes, weakref.finalize handle)
REGISTER_MIN_BYTES = 32 << 20


def _unregister_range(address):
    try:
        load().bms_host_unregister(c_vp(address))
    except Exception:  # interpreter shutdown
        pass


def register_if_reused(a):
    """Page-lock the host array `a` in place the SECOND time the same array object is handed in (a one-off array would pay the
    registration for nothing: it costs about one upload); released when the array is garbage collected or found resized.  Returns
    True when `a` is page-locked after the call.  SCRI_AMD_NO_REGISTER disables."""
    import weakref

    if os.environ.get("SCRI_AMD_NO_REGISTER") or a.nbytes < REGISTER_MIN_BYTES or not a.flags.c_contiguous:
        return False
    owner = a if a.base is None else a.base  # the object whose lifetime covers the memory
    while isinstance(owner, np.ndarray) and owner.base is not None:
        owner = owner.base
    oid, address, nbytes = id(owner), a.ctypes.data, a.nbytes
    with _reg_lock:
        reg = _registered.get(oid)
        if reg is not None:
            if reg[0] == address and reg[1] == nbytes:
                return True
            # the object's memory moved or changed size since it was page-locked: that registration is stale
            _registered.pop(oid, None)
            reg[2].detach()
            _unregister_range(reg[0])
            _seen_inputs.pop(oid, None)
        seen = _seen_inputs.pop(oid, None)
        if seen is not None and (seen[0]() is not owner or seen[1] != address or seen[2] != nbytes):
            seen = None  # a recycled id, or the same object with other memory: not a second sighting
        if seen is None:
            try:
                seen = [weakref.ref(owner), address, nbytes, 0]
            except TypeError:  # the owner cannot be weakly referenced: its release could never be observed
                return False
        seen[3] += 1
        _seen_inputs[oid] = seen
        while len(_seen_inputs) > 16:
            _seen_inputs.popitem(last=False)
        if seen[3] < 2:
            return False
        if load().bms_host_register(c_vp(address), nbytes) != 0:
            return False

        def _release(k=oid, addr=address):
            with _reg_lock:
                cur = _registered.get(k)
                if cur is not None and cur[0] == addr:
                    _registered.pop(k, None)
                _seen_inputs.pop(k, None)
            _unregister_range(addr)

        _registered[oid] = (address, nbytes, weakref.finalize(owner, _release))
        _seen_inputs.pop(oid, None)
        return True


def pinned_empty(shape, dtype):
    """np.empty(shape, dtype) in page-locked host memory (device-to-host copies land there at PCIe rate); falls back to
    ordinary memory for small arrays or when pinning fails."""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes < (1 << 20) or os.environ.get("SCRI_AMD_NO_PINNED"):
        return np.empty(shape, dtype=dtype)
    try:
        block = _PinnedBlock(nbytes)
    except (MemoryError, OSError):
        return np.empty(shape, dtype=dtype)
    return np.asarray(block).view(dtype).reshape(shape)


def _raise(code, ctx, what):
    msg = load().bms_last_error(ctx)
    msg = msg.decode() if msg else ""
    text = f"{what}: {msg} (status {code})"
    if code in (BMS_ERR_INVALID,):
        raise ValueError(text)
    if code == BMS_ERR_NOMEM:
        raise MemoryError(text)
    if code == BMS_ERR_UNSUPPORTED:
        raise NotImplementedError(text)
    raise BMSError(text)


_live_contexts = __import__("weakref").WeakSet()  # every Context of the process that has not been closed


def live_contexts():
    return [c for c in list(_live_contexts) if getattr(c, "_h", None)]


class Context:
    """Owns one bms_ctx (one GPU, one stream).  `stream`: optional hipStream_t handle (int), e.g.
    torch.cuda.current_stream().cuda_stream, so that work is ordered with the caller's."""

    def __init__(self, device=0, stream=None, workspace_limit=None):
        lib = load()
        h = c_vp()
        rc = lib.bms_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            _raise(rc, None, "bms_ctx_create")
        self._h = h
        _live_contexts.add(self)
        self.device = int(device)
        self.stream_handle = None  # the hipStream_t the caller handed over (0: the device's default stream); None: the context's own
        if stream is not None:
            self.set_stream(stream)
        if workspace_limit:
            self.check(lib.bms_ctx_set_workspace_limit(self._h, int(workspace_limit)), "bms_ctx_set_workspace_limit")

    def check(self, rc, what):
        if rc != 0:
            _raise(rc, self._h, what)

    def set_stream(self, stream):
        """stream: a hipStream_t handle; None = the context's own stream; 0 = the device's default (null) stream, which is what
        torch.cuda.current_stream().cuda_stream reports unless the caller switched streams"""
        self.stream_handle = None if stream is None else int(stream)
        if stream is None:
            self.check(load().bms_ctx_set_stream(self._h, c_vp(0)), "bms_ctx_set_stream")
        elif int(stream) == 0:
            self.check(load().bms_ctx_use_default_stream(self._h), "bms_ctx_use_default_stream")
        else:
            self.check(load().bms_ctx_set_stream(self._h, c_vp(int(stream))), "bms_ctx_set_stream")

    def synchronize(self):
        self.check(load().bms_ctx_synchronize(self._h), "bms_ctx_synchronize")

    def option(self, name, value=None):
        """Route option of THIS context (scri_amd/csrc/env.h lists them; "NO_GEMM_EVAL" or "SCRI_AMD_NO_GEMM_EVAL").  With a value:
        set it (flags 0 / 1) and return the previous value; without: read it.  Contexts take their defaults from the environment
        when they are created and never look at it again."""
        key = name.encode()
        old = c_i64(0)
        self.check(load().bms_ctx_get_option(self._h, key, ctypes.byref(old)), "bms_ctx_get_option")
        if value is not None:
            self.check(load().bms_ctx_set_option(self._h, key, int(value)), "bms_ctx_set_option")
        return int(old.value)

    def options(self, **kw):
        """with ctx.options(NO_GEMM_EVAL=1): ...   -- route options for the duration of a block"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            saved = {k: self.option(k, v) for k, v in kw.items()}
            try:
                yield self
            finally:
                for k, v in saved.items():
                    self.option(k, v)

        return scope()

    def eval_stats(self, reset=True):
        """(tiles launched, tiles off the LDS path, marches continued from global memory) of the evaluating product since the last reset"""
        out = (c_i64 * 3)()
        self.check(load().bms_ctx_get_eval_stats(self._h, out, 1 if reset else 0), "bms_ctx_get_eval_stats")
        return tuple(int(v) for v in out)

    def reserve(self, nbytes=0):
        """One device allocation of `nbytes` (0: one and a half times the work-space cap) that the context's work-space buffers are carved from
        afterwards: the first full-size call of the process then allocates nothing (bms_ctx_reserve)."""
        self.check(load().bms_ctx_reserve(self._h, int(nbytes)), "bms_ctx_reserve")

    def enable_timing(self, on=True):
        self.check(load().bms_ctx_enable_timing(self._h, 1 if on else 0), "bms_ctx_enable_timing")

    def get_timing(self, reset=True):
        """{kernel class: (milliseconds, launches)} accumulated by HIP events since the last reset."""
        n = len(KERNEL_TAGS)
        ms = (ctypes.c_double * n)()
        calls = (c_i64 * n)()
        self.check(load().bms_ctx_get_timing(self._h, ms, calls, 1 if reset else 0), "bms_ctx_get_timing")
        return {k: (ms[i], calls[i]) for i, k in enumerate(KERNEL_TAGS)}

    @property
    def handle(self):
        return self._h

    def close(self):
        peer = self.__dict__.pop("_pipeline_peer", None)  # second context of engine._transform_modes_pipelined
        if peer is not None:
            peer.close()
        if getattr(self, "_h", None):
            load().bms_ctx_destroy(self._h)
            self._h = None
            _PinnedBlock.trim()  # page-locked result blocks kept for reuse go back to the OS with the context

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """Process-wide context on device int(os.environ.get('SCRI_AMD_DEVICE', LOCAL_RANK or 0))."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("SCRI_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)
    return _default_ctx


def as_c16(a):
    """C-contiguous complex128 view/copy of `a` (the shim copies if the last-dim stride is not 16 B)."""
    return np.ascontiguousarray(a, dtype=np.complex128)


def dptr(a):
    return a.ctypes.data_as(c_dp)


def vptr(a):
    return c_vp(a.ctypes.data)
