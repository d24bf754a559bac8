"""ctypes loader for ``libbioen_hip.so`` (C ABI: ``include/bioen_hip.h``).

There is deliberately NO CPU fallback: if the HIP library is missing, or no
MI355X is visible, every compute entry point raises.  No torch, no numpy
compute -- this module only moves pointers.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BIOEN_HIP_LIBRARY: another build of the same library (diagnostic / A-B builds of tools/); never a fallback
LIB_PATH = os.environ.get("BIOEN_HIP_LIBRARY") or os.path.join(_HERE, "libbioen_hip.so")

dp = C.POINTER(C.c_double)
ctx_p = C.c_void_p


class LbfgsConfig(C.Structure):
    # field order = reference lbfgs_config_params (c_bioen_common.h:69-79)
    _fields_ = [("linesearch", C.c_int), ("max_iterations", C.c_int),
                ("delta", C.c_double), ("epsilon", C.c_double),
                ("ftol", C.c_double), ("gtol", C.c_double),
                ("wolfe", C.c_double), ("past", C.c_int),
                ("max_linesearch", C.c_int)]


class GslConfig(C.Structure):
    # field order = reference gsl_config_params (c_bioen_common.h:62-67)
    _fields_ = [("step_size", C.c_double), ("tol", C.c_double), ("max_iterations", C.c_int),
                ("algorithm", C.c_int)]


class VisualParams(C.Structure):
    # reference visual_params (c_bioen_common.h:89-92)
    _fields_ = [("debug", C.c_size_t), ("verbose", C.c_size_t)]


class OptResult(C.Structure):
    _fields_ = [("fmin", C.c_double), ("chi2", C.c_double), ("kl", C.c_double),
                ("seconds", C.c_double), ("lbfgs_code", C.c_int), ("iterations", C.c_int),
                ("evaluations", C.c_int), ("reserved", C.c_int)]


class BioenHipError(RuntimeError):
    """Failure inside libbioen_hip (HIP runtime, allocation, bad argument)."""


# every symbol include/bioen_hip.h declares: name -> (restype, argtypes)
_SIGNATURES = {
    "bioen_hip_version": (C.c_char_p, []),
    "bioen_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "bioen_hip_strerror": (C.c_char_p, [C.c_int]),
    "bioen_hip_last_error": (C.c_char_p, []),
    "bioen_hip_lbfgs_strerror": (C.c_char_p, [C.c_int]),
    "bioen_hip_set_fast_openmp_flag": (None, [C.c_int]),
    "bioen_hip_get_fast_openmp_flag": (C.c_int, []),
    "bioen_hip_ctx_create": (C.c_int, [C.c_int, C.c_int, dp, dp, C.c_int, C.POINTER(ctx_p)]),
    "bioen_hip_ctx_create_synthetic": (C.c_int, [C.c_int, C.c_int, dp, dp, dp, dp, C.c_ulonglong, C.c_int,
                                                 C.POINTER(ctx_p)]),
    "bioen_hip_ctx_create_sharded": (C.c_int, [C.c_int, C.c_longlong, dp, dp, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(ctx_p)]),
    "bioen_hip_ctx_create_synthetic_sharded": (C.c_int, [C.c_int, C.c_longlong, dp, dp, dp, dp, C.c_ulonglong,
                                                         C.c_int, C.c_int, C.c_int, C.POINTER(ctx_p)]),
    "bioen_hip_ctx_set_exchange_callback": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p]),
    "bioen_hip_ctx_shard": (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong),
                                      C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "bioen_hip_ctx_set_force_exchange": (C.c_int, [ctx_p, C.c_int]),
    "bioen_hip_ctx_set_mirror_exchange": (C.c_int, [ctx_p, C.c_int]),
    "bioen_hip_exchange_counts": (C.c_int, [ctx_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "bioen_hip_ctx_destroy": (C.c_int, [ctx_p]),
    "bioen_hip_ctx_shape": (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bioen_hip_ctx_read_ytilde": (C.c_int, [ctx_p, C.c_int, C.c_int, C.c_int, C.c_int, dp]),
    "bioen_hip_ctx_footprint": (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "bioen_hip_ctx_set_ytilde_target": (C.c_int, [ctx_p, dp]),
    "bioen_hip_ctx_set_affine": (C.c_int, [ctx_p, dp, dp]),
    "bioen_hip_ctx_set_direction_mode": (C.c_int, [ctx_p, C.c_int]),
    "bioen_hip_ctx_set_storage": (C.c_int, [ctx_p, C.c_int]),
    "bioen_hip_ctx_set_one_copy": (C.c_int, [ctx_p, C.c_int]),
    "bioen_hip_synchronize": (C.c_int, [ctx_p]),
    "bioen_hip_logw_weights": (C.c_int, [ctx_p, dp, dp, dp]),
    "bioen_hip_logw_fdf": (C.c_int, [ctx_p, dp, dp, C.c_double, dp, dp]),
    "bioen_hip_opt_lbfgs_logw": (C.c_int, [ctx_p, dp, dp, C.c_double, C.POINTER(LbfgsConfig),
                                           C.POINTER(VisualParams), dp, dp, C.POINTER(OptResult)]),
    "bioen_hip_opt_lbfgs_logw_batch": (C.c_int, [ctx_p, C.c_int, dp, dp, C.c_size_t, dp, C.POINTER(LbfgsConfig),
                                                 C.POINTER(VisualParams), C.c_int, dp, dp, C.POINTER(OptResult)]),
    "bioen_hip_forces_weights": (C.c_int, [ctx_p, dp, dp, dp]),
    "bioen_hip_forces_fdf": (C.c_int, [ctx_p, dp, dp, C.c_double, dp, dp]),
    "bioen_hip_forces_fdf_batch": (C.c_int, [ctx_p, C.c_int, dp, dp, dp, dp, dp]),
    "bioen_hip_opt_lbfgs_forces": (C.c_int, [ctx_p, dp, dp, C.c_double, C.POINTER(LbfgsConfig),
                                             C.POINTER(VisualParams), dp, dp, C.POINTER(OptResult)]),
    "bioen_hip_opt_lbfgs_forces_batch": (C.c_int, [ctx_p, C.c_int, dp, dp, C.c_size_t, dp, C.POINTER(LbfgsConfig),
                                                   C.POINTER(VisualParams), C.c_int, dp, dp, C.POINTER(OptResult)]),
    "bioen_hip_chi_squared": (C.c_int, [ctx_p, dp, dp, dp]),
    "bioen_hip_last_average": (C.c_int, [ctx_p, dp, dp]),
    "bioen_hip_kernel_stats": (C.c_int, [ctx_p, C.c_int, dp, C.POINTER(C.c_longlong)]),
    "bioen_hip_kernel_stats_ex": (C.c_int, [ctx_p, C.c_int, dp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "bioen_hip_kernel_stats_reset": (C.c_int, [ctx_p]),
    "bioen_hip_kernel_stats_enable": (C.c_int, [ctx_p, C.c_int]),
    "bioen_hip_speculation_stats": (C.c_int, [ctx_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "bioen_hip_debug_strip_stamps": (C.c_int, [ctx_p, C.c_int, C.POINTER(C.c_longlong), C.c_int]),
    "bioen_hip_ctx_layout": (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bioen_hip_debug_pass_probe": (C.c_int, [ctx_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "bioen_hip_ctx_create_raw": (C.c_int, [C.c_int, C.c_longlong, C.c_int, dp, dp, dp, C.c_int, C.POINTER(ctx_p)]),
    "bioen_hip_gsl_strerror": (C.c_char_p, [C.c_int]),
    "bioen_hip_opt_gsl_logw": (C.c_int, [ctx_p, dp, dp, C.c_double, C.POINTER(GslConfig), C.POINTER(VisualParams),
                                         dp, dp, C.POINTER(OptResult)]),
    "bioen_hip_opt_gsl_forces": (C.c_int, [ctx_p, dp, dp, C.c_double, C.POINTER(GslConfig), C.POINTER(VisualParams),
                                           dp, dp, C.POINTER(OptResult)]),
    "bioen_hip_selftest_multimin": (C.c_int, [C.c_int, C.c_int, dp, dp, C.POINTER(OptResult)]),
    "bioen_hip_multimin_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, dp, C.POINTER(GslConfig), dp,
                                          C.POINTER(OptResult)]),
    "bioen_hip_selftest_lbfgs": (C.c_int, [C.c_int, C.c_int, dp, C.POINTER(LbfgsConfig), dp, C.POINTER(OptResult)]),
    "bioen_hip_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "bioen_hip_comm_init": (C.c_int, [ctx_p, C.POINTER(C.c_ubyte), C.c_int, C.c_int]),
    "bioen_hip_comm_allgather": (C.c_int, [ctx_p, dp, C.c_size_t, dp]),
    "bioen_hip_comm_init_abandoned": (C.c_int, []),
    "bioen_hip_exchange_probe": (C.c_int, [ctx_p, C.c_size_t, C.c_int, dp]),
    "bioen_hip_read_probe": (C.c_int, [ctx_p, C.c_int, C.c_int, dp, C.POINTER(C.c_longlong)]),
    "bioen_hip_comm_destroy": (C.c_int, [ctx_p]),
    "bioen_hip_p2p_export": (C.c_int, [ctx_p, C.POINTER(C.c_ubyte)]),
    "bioen_hip_p2p_attach": (C.c_int, [ctx_p, C.POINTER(C.c_ubyte)]),
    "bioen_hip_p2p_detach": (C.c_int, [ctx_p]),
    "bioen_hip_exchange_transport": (C.c_int, [ctx_p]),
    "bioen_hip_exchange_selftest": (C.c_int, [ctx_p, C.c_int, C.POINTER(C.c_longlong)]),
    "bioen_hip_exchange_counts3": (C.c_int, [ctx_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                             C.POINTER(C.c_longlong)]),
    "bioen_hip_ctx_set_wait_timeout": (C.c_int, [ctx_p, C.c_double]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load libbioen_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise BioenHipError(
                "bioen_amd: %s not found -- build it with `make -C bioen_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as e; e.build()'`). "
                "There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        L = lib()
        raise BioenHipError("libbioen_hip: %s (%d): %s" % (L.bioen_hip_strerror(rc).decode(), rc,
                                                          L.bioen_hip_last_error().decode()))


def leave_process(status=0):
    """End the process the safe way for a rank whose RCCL initialisation was abandoned at its time bound: a helper thread
    is then still blocked inside ncclCommInitRank, and normal interpreter teardown would run librccl's / HIP's static
    destructors under it (hang or crash at exit).  Flushes stdio and leaves through os._exit in that case; a plain
    sys.exit otherwise.  Never re-executes anything."""
    import sys
    abandoned = False
    try:
        abandoned = bool(_lib is not None and _lib.bioen_hip_comm_init_abandoned())
    except Exception:
        abandoned = False
    if abandoned:
        try:
            sys.stdout.flush()
            sys.stderr.flush()
        finally:
            os._exit(int(status))
    sys.exit(int(status))


def column_segments(n, world=1):
    """The canonical column segments of an n-structure problem on `world` ranks (csrc/ctx.hpp: XStage; api.hip:
    segment_geometry): -> (nseg, segcols, columns per rank).  nseg = 8 whenever world divides 8 -- every sum over
    structures then has ONE shape on 1, 2, 4 and 8 GPUs, and their results are bit-identical --, else world;
    segcols = ceil(n / nseg) rounded up to 128; a rank holds nseg / world consecutive segments."""
    nseg = 8 if (world <= 8 and 8 % world == 0) else world
    if world == 1 and os.environ.get("BIOEN_HIP_SEGMENTS") == "1":      # the single-GPU opt-out (api.hip: segment_geometry)
        nseg = 1
    segcols = (-(-n // nseg) + 127) // 128 * 128
    return nseg, segcols, segcols * (nseg // world)


def device_count():
    n = C.c_int(0)
    check(lib().bioen_hip_device_count(C.byref(n)))
    return n.value


def as_f64(x, shape=None):
    """float64, C-contiguous, np.matrix-safe view/copy of x (the reference's pyx reads
    `.data` unchecked, c_bioen.pyx:274-290; we normalise instead)."""
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def ptr(a):
    return a.ctypes.data_as(dp)


def lbfgs_config(params):
    c = LbfgsConfig()
    for k in ("linesearch", "max_iterations", "past", "max_linesearch"):
        setattr(c, k, int(params[k]))
    for k in ("delta", "epsilon", "ftol", "gtol", "wolfe"):
        setattr(c, k, float(params[k]))
    return c


def selftest_lbfgs(kind, x0, params):
    """Run the library's L-BFGS driver on a built-in host objective (no GPU). -> (x, OptResult)"""
    x0 = as_f64(x0).ravel()
    cfg = lbfgs_config(params)
    out = np.empty_like(x0)
    info = OptResult()
    check(lib().bioen_hip_selftest_lbfgs(int(kind), x0.size, ptr(x0), C.byref(cfg), ptr(out), C.byref(info)))
    return out, info


GSL_ALGORITHMS = {"conjugate_fr": 0, "conjugate_pr": 1, "bfgs2": 2, "bfgs": 3, "steepest_descent": 4}


def gsl_config(algorithm, params):
    """algorithm: id 0..4 or one of GSL_ALGORITHMS; params: step_size, tol, max_iterations"""
    c = GslConfig()
    c.step_size, c.tol = float(params["step_size"]), float(params["tol"])
    c.max_iterations = int(params["max_iterations"])
    c.algorithm = GSL_ALGORITHMS[algorithm] if isinstance(algorithm, str) else int(algorithm)
    return c


def selftest_multimin(algorithm, kind, x0):
    """GSL's multimin test programme on the library's minimizers (host vectors, no GPU).
    -> (x, OptResult); OptResult.lbfgs_code = GSL status, .reserved = gradient evaluations"""
    x0 = as_f64(x0).ravel()
    out = np.empty_like(x0)
    info = OptResult()
    alg = GSL_ALGORITHMS[algorithm] if isinstance(algorithm, str) else int(algorithm)
    check(lib().bioen_hip_selftest_multimin(alg, int(kind), ptr(x0), ptr(out), C.byref(info)))
    return out, info


HOST_OBJECTIVE = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, dp, dp)


def multimin_host(objective, user, x0, algorithm, params):
    """The library's GSL-style minimizers on a host objective (no GPU).  `objective`: address of a C
    function ``int(void* user, const double* x, double* f, double* grad_or_NULL)`` (an int / c_void_p /
    ctypes function pointer) or a Python callable ``(x ndarray) -> f`` / ``(x, grad_out) -> f``
    wrapped here.  -> (x, OptResult); .lbfgs_code = GSL status, .reserved = gradient evaluations"""
    x0 = as_f64(x0).ravel()
    n = x0.size
    keep = None
    if callable(objective) and not isinstance(objective, C._CFuncPtr):
        pyfn = objective

        def _cb(_user, xp, fp, gp):
            try:
                x = np.ctypeslib.as_array(xp, shape=(n,))
                g = np.ctypeslib.as_array(gp, shape=(n,)) if gp else None
                fp[0] = float(pyfn(x, g))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        keep = HOST_OBJECTIVE(_cb)
        fn_addr = C.cast(keep, C.c_void_p)
    else:
        fn_addr = C.cast(objective, C.c_void_p) if not isinstance(objective, int) else C.c_void_p(objective)
    out = np.empty_like(x0)
    info = OptResult()
    cfg = gsl_config(algorithm, params)
    check(lib().bioen_hip_multimin_host(n, fn_addr, user, ptr(x0), C.byref(cfg), ptr(out), C.byref(info)))
    return out, info


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, C.c_size_t)


class Context(object):
    """Device-resident problem: yTilde (M x N) lives in HBM until close().

    rank/world > 1: the structures (columns) are sharded over `world` GPUs, one process each;
    this process keeps block `rank`.  N-vector arguments stay GLOBAL on every rank.  Before the
    first evaluation give the context a communicator: ``comm_init`` (RCCL) or ``set_exchange``
    (host-staged through any object with ``allgather_array``)."""

    def __init__(self, yTilde=None, YTilde=None, device=0, rank=0, world=1, _handle=None, _shape=None):
        L = lib()
        self._h = None
        self._exchange_keepalive = None
        self.rank, self.world = int(rank), int(world)
        if _handle is not None:
            self._h = _handle
            self.m, self.n = _shape
            self._query_shard()
            return
        yT = as_f64(yTilde)
        if yT.ndim != 2:
            raise ValueError("yTilde must be a 2-D (M x N) array")
        self.m, self.n = yT.shape
        YT = as_f64(YTilde).ravel()
        if YT.size != self.m:
            raise ValueError("YTilde must have M = %d entries" % self.m)
        h = ctx_p()
        check(L.bioen_hip_ctx_create_sharded(self.m, self.n, ptr(yT), ptr(YT), int(device), self.rank, self.world,
                                             C.byref(h)))
        self._h = h
        self._query_shard()

    def _query_shard(self):
        r, w, nl = C.c_int(0), C.c_int(1), C.c_int(0)
        ng, c0 = C.c_longlong(0), C.c_longlong(0)
        check(lib().bioen_hip_ctx_shard(self._h, C.byref(r), C.byref(w), C.byref(ng), C.byref(c0), C.byref(nl)))
        self.rank, self.world, self.col0, self.n_local = r.value, w.value, c0.value, nl.value

    @classmethod
    def synthetic(cls, m, n, YTrue, sig_sim, sig_exp, YTilde, seed=12345, device=0, rank=0, world=1):
        L = lib()
        a = [as_f64(v).ravel() for v in (YTrue, sig_sim, sig_exp, YTilde)]
        for v in a:
            if v.size != m:
                raise ValueError("synthetic(): every per-observable vector needs M entries")
        h = ctx_p()
        check(L.bioen_hip_ctx_create_synthetic_sharded(int(m), int(n), ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]),
                                                       C.c_ulonglong(seed), int(device), int(rank), int(world),
                                                       C.byref(h)))
        return cls(rank=rank, world=world, _handle=h, _shape=(int(m), int(n)))

    @classmethod
    def from_raw(cls, sim, exp, exp_err, structure_major=False, device=0):
        """Context from RAW observables: yTilde = sim / exp_err and YTilde = exp / exp_err are formed on
        the device.  sim: (M, N), or (N, M) with structure_major=True (one structure's observables
        contiguous -- uploaded in chunks and transposed on the device)."""
        sim = as_f64(sim)
        if sim.ndim != 2:
            raise ValueError("sim must be 2-D")
        n, m = sim.shape if structure_major else sim.shape[::-1]
        exp, err = as_f64(exp).ravel(), as_f64(exp_err).ravel()
        if exp.size != m or err.size != m:
            raise ValueError("exp / exp_err need one entry per observable")
        h = ctx_p()
        check(lib().bioen_hip_ctx_create_raw(int(m), int(n), 1 if structure_major else 0, ptr(sim), ptr(exp), ptr(err),
                                             int(device), C.byref(h)))
        return cls(_handle=h, _shape=(int(m), int(n)))

    def set_exchange(self, comm):
        """Complete cross-rank reductions through `comm.allgather_array` (host-staged; for ranks
        that cannot share an RCCL communicator).  `comm` = None removes the hook."""
        if comm is None:
            self._exchange_keepalive = None
            check(lib().bioen_hip_ctx_set_exchange_callback(self._h, None, None))
            return
        world, rank = self.world, self.rank

        def _cb(user, buf, count):
            try:
                arr = np.ctypeslib.as_array(buf, shape=(world * count,)).reshape(world, count)
                arr[:] = comm.allgather_array(arr[rank].copy())
                return 0
            except Exception:            # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        fn = EXCHANGE_FN(_cb)
        self._exchange_keepalive = fn
        check(lib().bioen_hip_ctx_set_exchange_callback(self._h, C.cast(fn, C.c_void_p), None))

    # -- lifetime ---------------------------------------------------------------
    def close(self):
        if self._h is not None:
            lib().bioen_hip_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _nvec(self, x, name):
        a = as_f64(x).ravel()
        if a.size != self.n:
            raise ValueError("%s must have N = %d entries, got %d" % (name, self.n, a.size))
        return a

    def _mvec(self, x, name):
        a = as_f64(x).ravel()
        if a.size != self.m:
            raise ValueError("%s must have M = %d entries, got %d" % (name, self.m, a.size))
        return a

    # -- data ---------------------------------------------------------------------
    def read_ytilde(self, row0=0, rows=None, col0=0, cols=None):
        """Block of the resident matrix; on a sharded context columns are relative to this rank's block."""
        rows = self.m - row0 if rows is None else rows
        cols = self.n_local - col0 if cols is None else cols
        out = np.empty((rows, cols))
        check(lib().bioen_hip_ctx_read_ytilde(self._h, row0, rows, col0, cols, ptr(out)))
        return out

    def footprint(self):
        """(forms, bytes) of the resident copies of the matrix: forms is a set out of {"rowmajor", "strips", "strips_colsum"}"""
        f, b = C.c_int(0), C.c_longlong(0)
        check(lib().bioen_hip_ctx_footprint(self._h, C.byref(f), C.byref(b)))
        names = {1: "rowmajor", 2: "strips", 4: "strips_colsum", 8: "reduced"}
        return {names[k] for k in names if f.value & k}, b.value

    def layout(self):
        """How the strip copies are held: {"one_copy": 0 | 1, "interleave": segments interleaved in the row-sum order copy,
        "relayouts": times that copy was moved to the other method's layout} (bioen_hip_ctx_layout)"""
        a, b, c_ = C.c_int(0), C.c_int(1), C.c_int(0)
        check(lib().bioen_hip_ctx_layout(self._h, C.byref(a), C.byref(b), C.byref(c_)))
        return {"one_copy": a.value, "interleave": b.value, "relayouts": c_.value}

    def set_target(self, YTilde):
        check(lib().bioen_hip_ctx_set_ytilde_target(self._h, ptr(self._mvec(YTilde, "YTilde"))))

    def set_affine(self, row_offset=None, row_scale=None):
        """Optimise against row_offset[i] + row_scale[i] * yTilde[i][j] without touching the resident
        matrix (DEER modulation depths, SAXS scaling factor).  None -> 0 resp. 1; a scalar
        row_scale applies to every row."""
        off = None if row_offset is None else self._mvec(row_offset, "row_offset")
        if row_scale is not None and np.ndim(row_scale) == 0:
            row_scale = np.full(self.m, float(row_scale))
        sc = None if row_scale is None else self._mvec(row_scale, "row_scale")
        check(lib().bioen_hip_ctx_set_affine(self._h, ptr(off) if off is not None else None,
                                             ptr(sc) if sc is not None else None))

    def set_direction_mode(self, mode):
        """'auto' (= 'gram'), 'twoloop' (liblbfgs' literal order of operations on the vectors) or
        'gram' (see include/bioen_hip.h: bioen_hip_ctx_set_direction_mode)."""
        code = {"auto": 0, "twoloop": 1, "gram": 2}[mode]
        check(lib().bioen_hip_ctx_set_direction_mode(self._h, code))

    STORAGE_FORMATS = {"f64": 0, "split": 1, "fp32": 2}

    def set_storage(self, fmt):
        """EXPERIMENT (opt-in, never the default): the log-weights matrix passes stream the centred matrix as
        'split' (fp32 + bf16 residual, 6 bytes per element, 2^-33 relative) or 'fp32' (4 bytes) instead of 'f64';
        reassembled to FP64 in registers, all sums FP64.  M <= 1024; served: the log-weights passes and the forces method's
        fused strip passes (forces_fdf(_batch), the forces optimizers); everything else of the forces method (forces_weights)
        raises BioenHipError (invalid state) while a reduced format is selected (the contract as implemented)."""
        check(lib().bioen_hip_ctx_set_storage(self._h, self.STORAGE_FORMATS[fmt]))

    def set_one_copy(self, on=True):
        """The log-weights method with ONE strip copy of the matrix resident (M <= 1024; 1 x instead of 2 x the matrix): the
        adjoint runs on the row-sum order copy, 1-3 % slower per launch; same minima, last bits differ from the two-copy
        default.  Call before the context's first gradient evaluation.  (Also: BIOEN_HIP_ONE_COPY=1 at creation; and a context
        takes this form by itself when the second copy does not fit.)"""
        check(lib().bioen_hip_ctx_set_one_copy(self._h, int(bool(on))))

    def synchronize(self):
        check(lib().bioen_hip_synchronize(self._h))

    # -- log-weights --------------------------------------------------------------
    def logw_weights(self, g):
        g = self._nvec(g, "g")
        w = np.empty(self.n)
        logs = C.c_double(0.0)
        check(lib().bioen_hip_logw_weights(self._h, ptr(g), ptr(w), C.byref(logs)))
        return w, logs.value

    def logw_fdf(self, g, G, theta, need_f=True, need_grad=True):
        g, G = self._nvec(g, "g"), self._nvec(G, "G")
        f = C.c_double(0.0)
        grad = np.empty(self.n) if need_grad else None
        check(lib().bioen_hip_logw_fdf(self._h, ptr(g), ptr(G), float(theta), C.byref(f),
                                       ptr(grad) if need_grad else None))
        return (f.value if need_f else None), grad

    def opt_lbfgs_logw(self, g0, G, theta, params, verbose=False, debug=False, want_weights=True):
        g0, G = self._nvec(g0, "g0"), self._nvec(G, "G")
        cfg = lbfgs_config(params)
        vis = VisualParams(int(bool(debug)), int(bool(verbose)))
        res = np.empty(self.n)
        w = np.empty(self.n) if want_weights else None
        info = OptResult()
        check(lib().bioen_hip_opt_lbfgs_logw(self._h, ptr(g0), ptr(G), float(theta), C.byref(cfg), C.byref(vis),
                                             ptr(res), ptr(w) if want_weights else None, C.byref(info)))
        return res, w, info

    def opt_gsl_logw(self, g0, G, theta, algorithm, params, verbose=False, debug=False, want_weights=True):
        """GSL-style minimizer (conjugate_fr/pr, bfgs2, bfgs, steepest_descent) with all vectors in HBM.
        -> (gopt, w or None, OptResult with the GSL status in .lbfgs_code)"""
        g0, G = self._nvec(g0, "g0"), self._nvec(G, "G")
        cfg = gsl_config(algorithm, params)
        vis = VisualParams(int(bool(debug)), int(bool(verbose)))
        res = np.empty(self.n)
        w = np.empty(self.n) if want_weights else None
        info = OptResult()
        check(lib().bioen_hip_opt_gsl_logw(self._h, ptr(g0), ptr(G), float(theta), C.byref(cfg), C.byref(vis),
                                           ptr(res), ptr(w) if want_weights else None, C.byref(info)))
        return res, w, info

    def opt_lbfgs_logw_batch(self, thetas, g0, G, params, max_batch=8, verbose=False, debug=False,
                             want_weights=True):
        """Solve a whole theta series; up to `max_batch` (<= 8) thetas share every pass over yTilde.
        g0: (n,) shared start or (ntheta, n) one start per theta.
        -> (results[ntheta, n], weights[ntheta, n] or None, [OptResult] * ntheta)"""
        thetas = as_f64(thetas).ravel()
        nt = thetas.size
        g0 = as_f64(g0)
        if g0.size == self.n:
            g0, stride = g0.ravel(), 0
        elif g0.shape == (nt, self.n):
            stride = self.n
        else:
            raise ValueError("g0 must have shape (n,) or (ntheta, n)")
        G = self._nvec(G, "G")
        cfg = lbfgs_config(params)
        vis = VisualParams(int(bool(debug)), int(bool(verbose)))
        res = np.empty((nt, self.n))
        w = np.empty((nt, self.n)) if want_weights else None
        infos = (OptResult * nt)()
        check(lib().bioen_hip_opt_lbfgs_logw_batch(self._h, nt, ptr(thetas), ptr(g0), stride, ptr(G), C.byref(cfg),
                                                   C.byref(vis), int(max_batch), ptr(res),
                                                   ptr(w) if want_weights else None, infos))
        return res, w, list(infos)

    # -- forces -------------------------------------------------------------------
    def forces_weights(self, forces, w0):
        f, w0 = self._mvec(forces, "forces"), self._nvec(w0, "w0")
        w = np.empty(self.n)
        check(lib().bioen_hip_forces_weights(self._h, ptr(f), ptr(w0), ptr(w)))
        return w

    def forces_fdf(self, forces, w0, theta, need_f=True, need_grad=True):
        fo, w0 = self._mvec(forces, "forces"), self._nvec(w0, "w0")
        f = C.c_double(0.0)
        grad = np.empty(self.m) if need_grad else None
        check(lib().bioen_hip_forces_fdf(self._h, ptr(fo), ptr(w0), float(theta), C.byref(f),
                                         ptr(grad) if need_grad else None))
        return (f.value if need_f else None), grad

    def forces_fdf_batch(self, forces, w0, thetas, need_grad=True):
        """K <= 8 evaluations sharing every matrix pass.  forces: (K, m), thetas: (K,) -> (f[K], grad[K, m] or None)"""
        thetas = as_f64(thetas).ravel()
        k = thetas.size
        fo = as_f64(forces).reshape(k, self.m)
        w0 = self._nvec(w0, "w0")
        f = np.empty(k)
        grad = np.empty((k, self.m)) if need_grad else None
        check(lib().bioen_hip_forces_fdf_batch(self._h, k, ptr(fo), ptr(w0), ptr(thetas), ptr(f),
                                               ptr(grad) if need_grad else None))
        return f, grad

    def opt_lbfgs_forces(self, forces0, w0, theta, params, verbose=False, debug=False, want_weights=True):
        f0, w0 = self._mvec(forces0, "forces0"), self._nvec(w0, "w0")
        cfg = lbfgs_config(params)
        vis = VisualParams(int(bool(debug)), int(bool(verbose)))
        res = np.empty(self.m)
        w = np.empty(self.n) if want_weights else None
        info = OptResult()
        check(lib().bioen_hip_opt_lbfgs_forces(self._h, ptr(f0), ptr(w0), float(theta), C.byref(cfg), C.byref(vis),
                                               ptr(res), ptr(w) if want_weights else None, C.byref(info)))
        return res, w, info

    def opt_gsl_forces(self, forces0, w0, theta, algorithm, params, verbose=False, debug=False, want_weights=True):
        f0, w0 = self._mvec(forces0, "forces0"), self._nvec(w0, "w0")
        cfg = gsl_config(algorithm, params)
        vis = VisualParams(int(bool(debug)), int(bool(verbose)))
        res = np.empty(self.m)
        w = np.empty(self.n) if want_weights else None
        info = OptResult()
        check(lib().bioen_hip_opt_gsl_forces(self._h, ptr(f0), ptr(w0), float(theta), C.byref(cfg), C.byref(vis),
                                             ptr(res), ptr(w) if want_weights else None, C.byref(info)))
        return res, w, info

    def opt_lbfgs_forces_batch(self, thetas, forces0, w0, params, max_batch=8, verbose=False, debug=False,
                               want_weights=True):
        """theta series of the forces method; up to `max_batch` (<= 8) thetas share every matrix pass.
        forces0: (m,) shared start or (ntheta, m).  -> (forces[ntheta, m], weights[ntheta, n] or None, infos)"""
        thetas = as_f64(thetas).ravel()
        nt = thetas.size
        f0 = as_f64(forces0)
        if f0.size == self.m:
            f0, stride = f0.ravel(), 0
        elif f0.shape == (nt, self.m):
            stride = self.m
        else:
            raise ValueError("forces0 must have shape (m,) or (ntheta, m)")
        w0 = self._nvec(w0, "w0")
        cfg = lbfgs_config(params)
        vis = VisualParams(int(bool(debug)), int(bool(verbose)))
        res = np.empty((nt, self.m))
        w = np.empty((nt, self.n)) if want_weights else None
        infos = (OptResult * nt)()
        check(lib().bioen_hip_opt_lbfgs_forces_batch(self._h, nt, ptr(thetas), ptr(f0), stride, ptr(w0), C.byref(cfg),
                                                     C.byref(vis), int(max_batch), ptr(res),
                                                     ptr(w) if want_weights else None, infos))
        return res, w, list(infos)

    # -- shared -------------------------------------------------------------------
    def last_average(self):
        """(yraw, yeff) of the most recent single-problem call on this context: yraw = yTilde . w of the
        resident matrix at the point that call ended on, yeff = row_offset + row_scale * yraw.  Only
        2 m doubles cross PCIe (no weights)."""
        yraw, yeff = np.empty(self.m), np.empty(self.m)
        check(lib().bioen_hip_last_average(self._h, ptr(yraw), ptr(yeff)))
        return yraw, yeff

    def chi_squared(self, w):
        """-> (0.5 |y_eff - YTilde|^2, yave).  With an affine row model set (set_affine) chi^2 is that of the
        effective observables off + sc * (yTilde . w), while yave is the RAW product yTilde . w of the
        resident matrix (what a nuisance refit needs); last_average() returns both."""
        w = self._nvec(w, "w")
        yave = np.empty(self.m)
        chi2 = C.c_double(0.0)
        check(lib().bioen_hip_chi_squared(self._h, ptr(w), ptr(yave), C.byref(chi2)))
        return chi2.value, yave

    # -- measurement ---------------------------------------------------------------
    def kernel_stats_enable(self, on=True):
        check(lib().bioen_hip_kernel_stats_enable(self._h, 1 if on else 0))

    def kernel_stats_reset(self):
        check(lib().bioen_hip_kernel_stats_reset(self._h))

    def kernel_stats(self):
        out = {}
        for which, name in ((0, "forward"), (1, "adjoint")):
            ms = C.c_double(0.0)
            cnt = C.c_longlong(0)
            pp = C.c_longlong(0)
            check(lib().bioen_hip_kernel_stats_ex(self._h, which, C.byref(ms), C.byref(cnt), C.byref(pp)))
            out[name] = {"total_ms": ms.value, "launches": cnt.value, "problem_passes": pp.value}
        return out

    def speculation_stats(self):
        """(issued, adopted): speculative line-search evaluations of the batch engines (both methods) so far"""
        a, b = C.c_longlong(0), C.c_longlong(0)
        check(lib().bioen_hip_speculation_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- RCCL ------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_ubyte * 128)()
        check(lib().bioen_hip_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        check(lib().bioen_hip_comm_init(self._h, buf, int(rank), int(nranks)))

    def set_force_exchange(self, on=True):
        """world = 1 only: execute the stage all-gathers of the sharded code path anyway (needs comm_init(id, 0, 1)
        or set_exchange); results do not change by a bit.  Puts the RCCL stage path under single-GPU tests."""
        check(lib().bioen_hip_ctx_set_force_exchange(self._h, 1 if on else 0))

    def set_mirror_exchange(self, on=True):
        """Measurement aid: the stage all-gathers of this rank-r-of-world context copy ITS part over every other rank's --
        the work one rank of a `world`-GPU run does per round, alone on one GPU (include/bioen_hip.h)."""
        check(lib().bioen_hip_ctx_set_mirror_exchange(self._h, 1 if on else 0))

    def exchange_counts(self):
        """(through RCCL, through the host callback): stage all-gathers executed on this context so far"""
        a, b = C.c_longlong(0), C.c_longlong(0)
        check(lib().bioen_hip_exchange_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def exchange_probe(self, count=1024, reps=50):
        """microseconds per stage exchange (all-gather of `count` doubles per rank)"""
        us = C.c_double(0.0)
        check(lib().bioen_hip_exchange_probe(self._h, int(count), int(reps), C.byref(us)))
        return us.value

    def read_probe(self, reps=10, form=0):
        """-> (GB/s of a plain read-only stream over the resident matrix, its bytes): the box's measured read ceiling.
        form: 0 = whichever is resident, 1 = row-major, 2 = row-sum strips, 4 = column-sum strips"""
        gbs, nbytes = C.c_double(0.0), C.c_longlong(0)
        check(lib().bioen_hip_read_probe(self._h, int(form), int(reps), C.byref(gbs), C.byref(nbytes)))
        return gbs.value, nbytes.value

    def comm_destroy(self):
        check(lib().bioen_hip_comm_destroy(self._h))

    # -- peer-to-peer stage exchange (hipIpc mailboxes, one kernel per all-gather) ------------------
    def p2p_export(self):
        """64-byte hipIpc handle of this rank's mailbox (allocated on first call)"""
        buf = (C.c_ubyte * 64)()
        check(lib().bioen_hip_p2p_export(self._h, buf))
        return bytes(buf)

    def p2p_attach(self, handles):
        """handles: the `world` mailbox handles in rank order (this rank's own is ignored); None for world = 1"""
        if handles is None:
            check(lib().bioen_hip_p2p_attach(self._h, None))
            return
        blob = b"".join(bytes(h) for h in handles)
        if len(blob) != 64 * self.world:
            raise ValueError("p2p_attach needs world = %d handles of 64 bytes" % self.world)
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        check(lib().bioen_hip_p2p_attach(self._h, buf))

    def p2p_detach(self):
        check(lib().bioen_hip_p2p_detach(self._h))

    TRANSPORTS = {0: "none", 1: "rccl", 2: "host", 3: "p2p"}

    def exchange_transport(self):
        """'none' | 'rccl' | 'host' | 'p2p': what the next stage exchange of this context goes through"""
        return self.TRANSPORTS[lib().bioen_hip_exchange_transport(self._h)]

    def exchange_selftest(self, reps=40):
        """`reps` back-to-back stage exchanges of varying size with a (rank, exchange, index) pattern, checked on the
        device; -> number of wrong doubles (0 = the active transport delivers)"""
        bad = C.c_longlong(0)
        check(lib().bioen_hip_exchange_selftest(self._h, int(reps), C.byref(bad)))
        return bad.value

    def exchange_counts3(self):
        """(RCCL, host callback, peer-to-peer): stage all-gathers executed on this context so far"""
        a, b, c3 = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        check(lib().bioen_hip_exchange_counts3(self._h, C.byref(a), C.byref(b), C.byref(c3)))
        return a.value, b.value, c3.value

    def set_wait_timeout(self, seconds):
        """bound (s) of every wait on a round or a peer; past it the call fails instead of hanging"""
        check(lib().bioen_hip_ctx_set_wait_timeout(self._h, float(seconds)))

    def comm_allgather(self, send, nranks):
        send = as_f64(send).ravel()
        recv = np.empty(send.size * nranks)
        check(lib().bioen_hip_comm_allgather(self._h, ptr(send), send.size, ptr(recv)))
        return recv.reshape(nranks, send.size)
