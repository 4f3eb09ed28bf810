"""ctypes binding of csrc/libcxschur.so (C ABI: include/cxschur.h).

Thin object wrappers used by tests/ and bench.py.  Nothing here computes: every
call goes to the HIP library and raises CxError when it reports a failure (there
is no CPU fallback).
"""
import ctypes
import os

import numpy as np

from .structure import BlockStructure, cx_block_structure

# RCCL between processes: the host driver of this pool supports dmabuf IPC only (must be set before the runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
_HERE = os.path.dirname(os.path.abspath(__file__))
# CXSCHUR_LIB: another build of the same library (A/B runs of compile-time switches on one box)
LIB_PATH = os.environ.get("CXSCHUR_LIB") or os.path.join(_HERE, "csrc", "libcxschur.so")
_lib = None

HOST, DEVICE = 0, 1
DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR = 0, 1, 2, 3
LOSS_NONE, LOSS_HUBER, LOSS_SOFT_L_ONE, LOSS_CAUCHY, LOSS_ARCTAN, LOSS_TOLERANT, LOSS_TUKEY = range(7)
CAMERA_ANGLE_AXIS, CAMERA_QUATERNION_MANIFOLD = 0, 1
IDENTITY, JACOBI, SCHUR_JACOBI, SCHUR_POWER_SERIES_EXPANSION, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL = 0, 1, 2, 3, 4, 5
CANONICAL_VIEWS, SINGLE_LINKAGE = 0, 1
SUCCESS, NO_CONVERGENCE, FAILURE, FATAL_ERROR = 0, 1, 2, 3
NOTE_DOUBLE_PRECISION_FACTOR, NOTE_SPSE_INITIALIZATION_SKIPPED = 1, 2

# every symbol include/cxschur.h declares (tests check that the library exports them all)
EXPORTED_SYMBOLS = [
    "cx_context_create", "cx_sparse_cholesky_distribution_host", "cx_matrix_static_path", "cx_evaluator_set_column_scale", "cx_schur_pair_lists_host", "cx_context_create_multi", "cx_context_num_shards", "cx_matrix_shard_layout", "cx_context_destroy", "cx_comm_unique_id", "cx_context_set_comm", "cx_context_set_comm_callback", "cx_context_rank",
    "cx_context_num_ranks", "cx_allreduce_sum", "cx_malloc", "cx_free", "cx_memcpy_h2d", "cx_memcpy_d2h",
    "cx_memset_zero", "cx_synchronize", "cx_context_stream", "cx_last_error", "cx_device_name",
    "cx_matrix_create", "cx_matrix_destroy", "cx_matrix_num_rows", "cx_matrix_num_cols",
    "cx_matrix_num_nonzeros", "cx_matrix_is_static_239", "cx_matrix_device_values", "cx_matrix_set_values",
    "cx_matrix_get_values", "cx_matrix_values_changed", "cx_matrix_set_zero", "cx_matrix_right_multiply",
    "cx_matrix_left_multiply", "cx_matrix_right_multiply_e", "cx_matrix_right_multiply_f",
    "cx_matrix_left_multiply_e", "cx_matrix_left_multiply_f", "cx_matrix_squared_column_norm", "cx_matrix_scale_columns",
    "cx_matrix_last_kernel_ms", "cx_solver_create", "cx_solver_destroy", "cx_solver_default_options",
    "cx_solver_solve", "cx_solver_last_timing", "cx_solver_sample_next", "cx_solver_kernel_stats", "cx_schur_eliminate_dense", "cx_schur_back_substitute",
    "cx_implicit_schur_multiply", "cx_dense_cholesky_solve", "cx_evaluator_create_bal", "cx_evaluator_destroy",
    "cx_evaluator_jacobian", "cx_evaluator_row_of_observation", "cx_evaluator_evaluate", "cx_evaluator_set_loss", "cx_minimizer_default_options", "cx_minimize", "cx_schur_sparse_structure", "cx_visibility_structure", "cx_visibility_clusters_host",
    "cx_evaluator_set_camera_model", "cx_evaluator_num_parameters", "cx_evaluator_num_effective_parameters", "cx_evaluator_plus",
    "cx_context_set_comm_timeout", "cx_debug_inject_failure", "cx_debug_stall_stream", "cx_debug_force_rank_count", "cx_evaluator_device_residuals_match",
    "cx_matrix_right_multiply_overwrite", "cx_host_registration_policy", "cx_host_register", "cx_host_registrations_release", "cx_transfer_stats_get",
    "cx_evaluator_last_kernel_ms", "cx_evaluator_device_residuals", "cx_evaluator_set_emit_camera_major", "cx_sparse_cholesky_plan_host", "cx_sparse_cholesky_schedule_host", "cx_detect_structure", "cx_partition_points", "cx_stable_schur_ordering",
]


class CxError(RuntimeError):
    pass


class cx_solver_options(ctypes.Structure):
    _fields_ = [
        ("type", ctypes.c_int32),
        ("preconditioner_type", ctypes.c_int32),
        ("min_num_iterations", ctypes.c_int32),
        ("max_num_iterations", ctypes.c_int32),
        ("residual_reset_period", ctypes.c_int32),
        ("num_eliminate_blocks", ctypes.c_int32),
        ("use_mixed_precision_solves", ctypes.c_int32),
        ("max_num_refinement_iterations", ctypes.c_int32),
        ("max_num_spse_iterations", ctypes.c_int32),
        ("use_spse_initialization", ctypes.c_int32),
        ("spse_tolerance", ctypes.c_double),
        ("deterministic", ctypes.c_int32),
        ("use_explicit_schur_complement", ctypes.c_int32),
        ("visibility_clustering_type", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class cx_transfer_stats(ctypes.Structure):
    _fields_ = [
        ("h2d_bytes", ctypes.c_int64), ("d2h_bytes", ctypes.c_int64),
        ("h2d_copies", ctypes.c_int64), ("d2h_copies", ctypes.c_int64),
        ("h2d_registered_bytes", ctypes.c_int64), ("d2h_registered_bytes", ctypes.c_int64),
        ("h2d_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double),
        ("registered_bytes", ctypes.c_int64),
        ("num_registered", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("register_ms", ctypes.c_double),
        ("num_register_calls", ctypes.c_int64),
    ]


class cx_per_solve_options(ctypes.Structure):
    _fields_ = [
        ("D", ctypes.c_void_p),
        ("r_tolerance", ctypes.c_double),
        ("q_tolerance", ctypes.c_double),
        ("memspace", ctypes.c_int32),
        ("b_on_device", ctypes.c_int32),
    ]


class cx_summary(ctypes.Structure):
    _fields_ = [
        ("residual_norm", ctypes.c_double),
        ("num_iterations", ctypes.c_int32),
        ("termination_type", ctypes.c_int32),
        ("message", ctypes.c_char * 256),
        ("notes", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class cx_solve_timing(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in
                ("setup_ms", "eliminate_ms", "reduced_solve_ms", "back_substitute_ms", "total_ms", "allreduce_ms",
                 "allreduce_host_ms", "allreduce_calls", "allreduce_bytes", "sampled")]


class cx_kernel_stat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 64), ("sampled_ms", ctypes.c_double),
                ("sampled_launches", ctypes.c_int32), ("launches", ctypes.c_int32)]


class cx_minimizer_options(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("max_num_iterations", "max_num_consecutive_invalid_steps", "jacobi_scaling",
                 "use_nonmonotonic_steps", "max_consecutive_nonmonotonic_steps", "reserved")] + \
               [(n, ctypes.c_double) for n in
                ("initial_trust_region_radius", "max_trust_region_radius", "min_trust_region_radius",
                 "min_relative_decrease", "min_lm_diagonal", "max_lm_diagonal", "function_tolerance",
                 "gradient_tolerance", "parameter_tolerance", "eta")]


class cx_iteration_summary(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("iteration", "step_is_valid", "step_is_nonmonotonic", "step_is_successful")] + \
               [(n, ctypes.c_double) for n in
                ("cost", "cost_change", "gradient_max_norm", "gradient_norm", "step_norm", "relative_decrease",
                 "trust_region_radius", "eta")] + \
               [("linear_solver_iterations", ctypes.c_int32), ("reserved", ctypes.c_int32)] + \
               [(n, ctypes.c_double) for n in ("iteration_ms", "linear_solver_ms", "jacobian_ms", "residual_ms")]


class cx_minimizer_summary(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("termination_type", "num_successful_steps", "num_unsuccessful_steps", "num_iterations")] + \
               [(n, ctypes.c_double) for n in ("initial_cost", "final_cost", "total_ms")] + \
               [("message", ctypes.c_char * 256)]


CONVERGENCE, MIN_NO_CONVERGENCE, MIN_FAILURE = 0, 1, 2


def minimizer_options(**kw):
    """Solver::Options defaults (include/ceres/solver.h:250-330, 620-640)."""
    o = cx_minimizer_options(50, 5, 1, 0, 5, 0, 1e4, 1e16, 1e-32, 1e-3, 1e-6, 1e32, 1e-6, 1e-10, 1e-8, 1e-1)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _summary_dict(s):
    return {n: (getattr(s, n).decode() if n == "message" else getattr(s, n)) for n, _ in s._fields_ if n != "reserved"}


def library_path():
    return LIB_PATH


def load_library():
    """Load libcxschur.so; raises CxError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CxError("libcxschur.so is missing: run __graft_entry__.build() "
                      "(python ceres-solver-ceres-solver_amd/build.py); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    lib.cx_last_error.restype = ctypes.c_char_p
    lib.cx_matrix_num_rows.restype = ctypes.c_int64
    lib.cx_matrix_num_cols.restype = ctypes.c_int64
    lib.cx_matrix_num_nonzeros.restype = ctypes.c_int64
    lib.cx_matrix_device_values.restype = ctypes.c_void_p
    lib.cx_matrix_last_kernel_ms.restype = ctypes.c_double
    lib.cx_evaluator_last_kernel_ms.restype = ctypes.c_double
    lib.cx_evaluator_device_residuals.restype = ctypes.c_void_p
    lib.cx_evaluator_num_parameters.restype = ctypes.c_int64
    lib.cx_evaluator_num_effective_parameters.restype = ctypes.c_int64
    lib.cx_evaluator_jacobian.restype = ctypes.c_void_p
    lib.cx_context_stream.restype = ctypes.c_void_p
    for name in ("cx_matrix_destroy", "cx_solver_destroy", "cx_evaluator_destroy", "cx_context_destroy",
                 "cx_matrix_num_rows", "cx_matrix_num_cols", "cx_matrix_num_nonzeros", "cx_matrix_is_static_239", "cx_matrix_static_path",
                 "cx_matrix_device_values", "cx_matrix_last_kernel_ms", "cx_evaluator_jacobian",
                 "cx_evaluator_last_kernel_ms", "cx_evaluator_device_residuals", "cx_context_stream", "cx_context_rank",
                 "cx_context_num_ranks", "cx_context_num_shards"):
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise CxError("cxschur error %d: %s" % (rc, load_library().cx_last_error().decode()))


def _ptr(a):
    """Host numpy array (float64, contiguous) or DeviceArray or None -> void*"""
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return ctypes.c_void_p(a.ptr)
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)   # holds a reference: a temporary stays alive for the call


def _space(*arrays):
    spaces = {DEVICE if isinstance(a, DeviceArray) else HOST for a in arrays if a is not None}
    assert len(spaces) <= 1, "mixing host and device arrays in one call"
    return spaces.pop() if spaces else HOST


def _f64(a):
    if a is None or isinstance(a, DeviceArray):
        return a
    return np.ascontiguousarray(a, dtype=np.float64)


def host_registration_policy(sightings=0, min_bytes=256 << 10, max_total_bytes=16 << 30):
    """cx_host_registration_policy: when caller arrays are registered with the HIP runtime (0 never -- the default --, 1 first
    sight, 2 second).  Switching it on is a promise: every array handed to the library stays allocated until
    host_registrations_release()."""
    _check(load_library().cx_host_registration_policy(int(sightings), ctypes.c_int64(min_bytes), ctypes.c_int64(max_total_bytes)))


def host_register(array):
    _check(load_library().cx_host_register(_ptr(array), ctypes.c_size_t(array.nbytes)))


def host_registrations_release():
    _check(load_library().cx_host_registrations_release())


class Context:
    def __init__(self, device=0, devices=None):
        """device: one GPU.  devices = [d0, d1, ...]: several shards behind one set of handles in this process
        (cx_context_create_multi) -- distinct devices exchange through RCCL, a repeated device id means logical
        shards on that GPU with the in-process sum."""
        lib = load_library()
        self._h = ctypes.c_void_p()
        if devices is None:
            _check(lib.cx_context_create(int(device), ctypes.byref(self._h)))
        else:
            ids = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            _check(lib.cx_context_create_multi(len(devices), ids, ctypes.byref(self._h)))
        self.lib = lib

    @property
    def num_shards(self):
        return self.lib.cx_context_num_shards(self._h)

    def close(self):
        if self._h:
            self.lib.cx_context_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def name(self):
        buf = ctypes.create_string_buffer(256)
        _check(self.lib.cx_device_name(self._h, buf, 256))
        return buf.value.decode()

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _check(load_library().cx_comm_unique_id(buf))
        return buf.raw

    def set_comm(self, rank, nranks, unique_id):
        _check(self.lib.cx_context_set_comm(self._h, int(rank), int(nranks), unique_id))

    def set_comm_callback(self, rank, nranks, allreduce):
        """allreduce(host_array) must sum a float64 numpy array over the ranks in place; the
        device buffer is staged through the host (rehearsal transport, see cxschur.h)."""
        def _cb(dptr, n, _user):
            try:
                host = np.empty(n, dtype=np.float64)
                _check(self.lib.cx_memcpy_d2h(self._h, _ptr(host), ctypes.c_void_p(dptr), ctypes.c_size_t(8 * n)))
                allreduce(host)
                _check(self.lib.cx_memcpy_h2d(self._h, ctypes.c_void_p(dptr), _ptr(host), ctypes.c_size_t(8 * n)))
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)(_cb)
        _check(self.lib.cx_context_set_comm_callback(self._h, int(rank), int(nranks), self._cb, None))

    @property
    def rank(self):
        return self.lib.cx_context_rank(self._h)

    @property
    def num_ranks(self):
        return self.lib.cx_context_num_ranks(self._h)

    def synchronize(self):
        _check(self.lib.cx_synchronize(self._h))

    def set_comm_timeout(self, seconds):
        _check(self.lib.cx_context_set_comm_timeout(self._h, ctypes.c_double(seconds)))

    def inject_failure(self, shard, nth_collective):
        """Test hook: shard `shard` fails right before the nth collective it enters from now on (-1 clears)."""
        _check(self.lib.cx_debug_inject_failure(self._h, int(shard), ctypes.c_int64(nth_collective)))

    def stall_stream(self, milliseconds):
        _check(self.lib.cx_debug_stall_stream(self._h, int(milliseconds)))

    def transfer_stats(self, reset=False):
        """What crossed PCIe through this context's calls (cx_transfer_stats_get)."""
        t = cx_transfer_stats()
        _check(self.lib.cx_transfer_stats_get(self._h, ctypes.byref(t), int(bool(reset))))
        return {n: getattr(t, n) for n, _ in cx_transfer_stats._fields_ if n != "reserved"}

    def allreduce_sum(self, dev, offset=0, count=None):
        """In-place sum over the ranks of dev[offset:offset+count] (float64 device array)."""
        count = dev.size - offset if count is None else count
        _check(self.lib.cx_allreduce_sum(self._h, ctypes.c_void_p(dev.ptr + 8 * offset), ctypes.c_int64(count)))

    def empty(self, n, dtype=np.float64):
        return DeviceArray(self, int(n), np.dtype(dtype))

    def zeros(self, n, dtype=np.float64):
        d = self.empty(n, dtype)
        _check(self.lib.cx_memset_zero(self._h, ctypes.c_void_p(d.ptr), ctypes.c_size_t(d.nbytes)))
        return d

    def to_device(self, host):
        host = np.ascontiguousarray(host)
        d = self.empty(host.size, host.dtype)
        _check(self.lib.cx_memcpy_h2d(self._h, ctypes.c_void_p(d.ptr), _ptr(host), ctypes.c_size_t(host.nbytes)))
        return d


class DeviceArray:
    """A device allocation owned by a Context (cx_malloc / cx_free)."""

    def __init__(self, ctx, n, dtype):
        self.ctx, self.size, self.dtype = ctx, n, dtype
        self.nbytes = n * dtype.itemsize
        p = ctypes.c_void_p()
        _check(ctx.lib.cx_malloc(ctx._h, ctypes.c_size_t(max(self.nbytes, 1)), ctypes.byref(p)))
        self.ptr = p.value

    def to_host(self):
        out = np.empty(self.size, dtype=self.dtype)
        _check(self.ctx.lib.cx_memcpy_d2h(self.ctx._h, _ptr(out), ctypes.c_void_p(self.ptr), ctypes.c_size_t(self.nbytes)))
        return out

    def copy_from_host(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.size == self.size
        _check(self.ctx.lib.cx_memcpy_h2d(self.ctx._h, ctypes.c_void_p(self.ptr), _ptr(host), ctypes.c_size_t(self.nbytes)))

    def free(self):
        if self.ptr and self.ctx._h:
            self.ctx.lib.cx_free(self.ctx._h, ctypes.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Matrix:
    """Device BlockSparseMatrix (cx_matrix)."""

    def __init__(self, ctx, bs, num_eliminate_blocks=0, _handle=None):
        self.ctx, self.bs, self.lib = ctx, bs, ctx.lib
        self.num_eliminate_blocks = int(num_eliminate_blocks)
        self._owned = _handle is None
        if _handle is None:
            self._h = ctypes.c_void_p()
            _check(self.lib.cx_matrix_create(ctx._h, bs.c, int(num_eliminate_blocks), ctypes.byref(self._h)))
        else:
            self._h = ctypes.c_void_p(_handle)

    def close(self):
        if self._h and self._owned and self.ctx._h:
            self.lib.cx_matrix_destroy(self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    num_rows = property(lambda s: s.lib.cx_matrix_num_rows(s._h))
    num_cols = property(lambda s: s.lib.cx_matrix_num_cols(s._h))
    num_nonzeros = property(lambda s: s.lib.cx_matrix_num_nonzeros(s._h))
    is_static_239 = property(lambda s: bool(s.lib.cx_matrix_is_static_239(s._h)))
    static_path = property(lambda s: int(s.lib.cx_matrix_static_path(s._h)))   # 0 dynamic-size, 1 native <2,3,9>, 2 embedded
    last_kernel_ms = property(lambda s: s.lib.cx_matrix_last_kernel_ms(s._h))

    def shard_layout(self):
        """(e-block bounds, row-block bounds) of the shards of a matrix on a multi-shard context, n + 1 entries each."""
        eb, rb = np.zeros(17, dtype=np.int32), np.zeros(17, dtype=np.int32)
        n = self.lib.cx_matrix_shard_layout(self._h, _ptr(eb), _ptr(rb), 17)
        return eb[:n + 1].copy(), rb[:n + 1].copy()

    def set_values(self, values):
        values = _f64(values)
        _check(self.lib.cx_matrix_set_values(self._h, _ptr(values), _space(values)))

    def values_changed(self):
        """Tell the matrix its values were rewritten in place (drops the camera-major copy)."""
        _check(self.lib.cx_matrix_values_changed(self._h))

    def get_values(self):
        out = np.empty(self.num_nonzeros)
        _check(self.lib.cx_matrix_get_values(self._h, _ptr(out)))
        return out

    def _vec(self, a, n):
        a = _f64(a)
        assert a.size == n
        return a

    def right_multiply(self, x, y=None):
        """y += A x ; host arrays return a new array, device arrays are updated in place."""
        x = self._vec(x, self.num_cols)
        if isinstance(x, DeviceArray):
            _check(self.lib.cx_matrix_right_multiply(self._h, _ptr(x), _ptr(y), DEVICE))
            return y
        y = np.zeros(self.num_rows) if y is None else np.array(y, dtype=np.float64)
        _check(self.lib.cx_matrix_right_multiply(self._h, _ptr(x), _ptr(y), HOST))
        return y

    def left_multiply(self, x, y=None):
        x = self._vec(x, self.num_rows)
        if isinstance(x, DeviceArray):
            _check(self.lib.cx_matrix_left_multiply(self._h, _ptr(x), _ptr(y), DEVICE))
            return y
        y = np.zeros(self.num_cols) if y is None else np.array(y, dtype=np.float64)
        _check(self.lib.cx_matrix_left_multiply(self._h, _ptr(x), _ptr(y), HOST))
        return y

    def partitioned_multiply(self, part, transpose, x, y):
        """y += E x, F x, E' x or F' x (host numpy arrays; part is "e" or "f")."""
        fn = getattr(self.lib, "cx_matrix_%s_multiply_%s" % ("left" if transpose else "right", part))
        x = _f64(x)
        y = np.array(y, dtype=np.float64)
        _check(fn(self._h, _ptr(x), _ptr(y), HOST))
        return y

    def squared_column_norm(self, out=None):
        if isinstance(out, DeviceArray):
            _check(self.lib.cx_matrix_squared_column_norm(self._h, _ptr(out), DEVICE))
            return out
        out = np.zeros(self.num_cols)
        _check(self.lib.cx_matrix_squared_column_norm(self._h, _ptr(out), HOST))
        return out

    def scale_columns(self, scale):
        scale = self._vec(scale, self.num_cols)
        _check(self.lib.cx_matrix_scale_columns(self._h, _ptr(scale), _space(scale)))


def default_options(**kw):
    o = cx_solver_options()
    load_library().cx_solver_default_options(ctypes.byref(o))
    for k, v in kw.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    return o


class Solver:
    """LinearSolver (cx_solver)."""

    def __init__(self, ctx, **options):
        self.ctx, self.lib = ctx, ctx.lib
        self.options = default_options(**options)
        self._h = ctypes.c_void_p()
        _check(self.lib.cx_solver_create(ctx._h, ctypes.byref(self.options), ctypes.byref(self._h)))

    def close(self):
        if self._h and self.ctx._h:
            self.lib.cx_solver_destroy(self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, A, b, D=None, r_tolerance=-1.0, q_tolerance=0.0, x=None):
        """Returns (x, summary).  Host in -> host out; device in -> device x (must be passed)."""
        b, D = _f64(b), _f64(D)
        space = _space(b, D, x)
        if space == HOST:
            x = np.full(A.num_cols, np.nan)
        else:
            assert x is not None
        ps = cx_per_solve_options()
        ps.D = _ptr(D).value if D is not None else None
        ps.r_tolerance, ps.q_tolerance, ps.memspace = r_tolerance, q_tolerance, space
        s = cx_summary()
        _check(self.lib.cx_solver_solve(self._h, A._h, _ptr(b), ctypes.byref(ps), _ptr(x), ctypes.byref(s)))
        return x, s

    def kernel_stats(self):
        arr = (cx_kernel_stat * 8)()
        n = ctypes.c_int32()
        _check(self.lib.cx_solver_kernel_stats(self._h, arr, 8, ctypes.byref(n)))
        return [dict(name=arr[i].name.decode(), sampled_ms=arr[i].sampled_ms,
                     sampled_launches=arr[i].sampled_launches, launches=arr[i].launches) for i in range(n.value)]

    def sample_next(self):
        """the next solve takes phase times and kernel samples (launch-bound solvers take them every 16th solve only)"""
        _check(self.lib.cx_solver_sample_next(self._h))

    def timing(self):
        t = cx_solve_timing()
        _check(self.lib.cx_solver_last_timing(self._h, ctypes.byref(t)))
        return {n: getattr(t, n) for n, _ in cx_solve_timing._fields_}


def eliminate_dense(ctx, A, b, D, num_cols_f):
    lhs = np.zeros((num_cols_f, num_cols_f))
    rhs = np.zeros(num_cols_f)
    _check(ctx.lib.cx_schur_eliminate_dense(ctx._h, A._h, _ptr(_f64(b)), _ptr(_f64(D)), _ptr(lhs),
                                            _ptr(rhs) if b is not None else None, HOST))
    return lhs, (rhs if b is not None else None)


def back_substitute(ctx, A, b, D, z):
    x = np.zeros(A.num_cols)
    _check(ctx.lib.cx_schur_back_substitute(ctx._h, A._h, _ptr(_f64(b)), _ptr(_f64(D)), _ptr(_f64(z)), _ptr(x), HOST))
    return x


def implicit_schur_multiply(ctx, A, D, b, x, num_cols_f):
    y = np.zeros(num_cols_f)
    rhs = np.zeros(num_cols_f) if b is not None else None
    _check(ctx.lib.cx_implicit_schur_multiply(ctx._h, A._h, _ptr(_f64(D)), _ptr(_f64(b)), _ptr(_f64(x)), _ptr(y),
                                              _ptr(rhs), HOST))
    return y, rhs


def dense_cholesky_solve(ctx, lhs, rhs):
    a = np.array(lhs, dtype=np.float64)
    n = a.shape[0]
    x = np.zeros(n)
    s = cx_summary()
    _check(ctx.lib.cx_dense_cholesky_solve(ctx._h, n, _ptr(a), _ptr(_f64(rhs)), _ptr(x), HOST, ctypes.byref(s)))
    return x, s


class Evaluator:
    """Bundle-adjustment Evaluator (cx_evaluator)."""

    def __init__(self, ctx, problem):
        self.ctx, self.lib, self.problem = ctx, ctx.lib, problem
        self._h = ctypes.c_void_p()
        cam = np.ascontiguousarray(problem.camera_index, dtype=np.int32)
        pt = np.ascontiguousarray(problem.point_index, dtype=np.int32)
        obs = np.ascontiguousarray(problem.observations, dtype=np.float64)
        _check(self.lib.cx_evaluator_create_bal(ctx._h, int(problem.num_cameras), int(problem.num_points),
                                                ctypes.c_int64(problem.num_observations), _ptr(cam), _ptr(pt),
                                                _ptr(obs), ctypes.byref(self._h)))
        self.num_cols = 3 * problem.num_points + 9 * problem.num_cameras
        self.num_rows = 2 * problem.num_observations

    def close(self):
        if self._h and self.ctx._h:
            self.lib.cx_evaluator_destroy(self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def jacobian(self, bs=None):
        return Matrix(self.ctx, bs, _handle=self.lib.cx_evaluator_jacobian(self._h))

    def row_of_observation(self):
        out = np.zeros(self.problem.num_observations, dtype=np.int64)
        _check(self.lib.cx_evaluator_row_of_observation(self._h, _ptr(out)))
        return out

    last_kernel_ms = property(lambda s: s.lib.cx_evaluator_last_kernel_ms(s._h))

    def set_emit_camera_major(self, on):
        """Whether Jacobian evaluations also write the camera-major copy of F (off when ScaleColumns always follows)."""
        _check(self.lib.cx_evaluator_set_emit_camera_major(self._h, int(bool(on))))

    def set_column_scale(self, scale):
        """Jacobian evaluations write J diag(scale) (and the camera-major copy) in one pass; None clears it."""
        scale = _f64(scale)
        _check(self.lib.cx_evaluator_set_column_scale(self._h, _ptr(scale), _space(scale)))

    def set_camera_model(self, model):
        """CAMERA_ANGLE_AXIS (9 parameters) or CAMERA_QUATERNION_MANIFOLD (10 parameters, 9 tangent)."""
        _check(self.lib.cx_evaluator_set_camera_model(self._h, int(model)))

    @property
    def num_parameters(self):
        return int(self.lib.cx_evaluator_num_parameters(self._h))

    @property
    def num_effective_parameters(self):
        return int(self.lib.cx_evaluator_num_effective_parameters(self._h))

    def plus(self, x, delta):
        """Evaluator::Plus on host arrays: x ambient, delta tangent."""
        x, delta = _f64(x), _f64(delta)
        out = np.zeros(self.num_parameters)
        _check(self.lib.cx_evaluator_plus(self._h, _ptr(x), _ptr(delta), _ptr(out), HOST))
        return out

    def set_loss(self, loss_type, a=1.0, b=0.0):
        """loss_type: one of LOSS_* (cx_loss_type); a, b the LossFunction constructor arguments."""
        _check(self.lib.cx_evaluator_set_loss(self._h, int(loss_type), ctypes.c_double(a), ctypes.c_double(b)))

    def evaluate(self, state, want_residuals=True, want_gradient=True, want_jacobian=True, residuals=None,
                 gradient=None):
        """Returns (cost, residuals, gradient); the Jacobian lands in self.jacobian()."""
        state = _f64(state)
        cost = ctypes.c_double()
        if isinstance(state, DeviceArray):
            _check(self.lib.cx_evaluator_evaluate(self._h, _ptr(state), ctypes.byref(cost), _ptr(residuals),
                                                  _ptr(gradient), int(want_jacobian), DEVICE))
            return cost.value, residuals, gradient
        res = np.zeros(self.num_rows) if want_residuals else None
        grad = np.zeros(self.num_cols) if want_gradient else None
        _check(self.lib.cx_evaluator_evaluate(self._h, _ptr(state), ctypes.byref(cost), _ptr(res), _ptr(grad),
                                              int(want_jacobian), HOST))
        return cost.value, res, grad


def minimize(evaluator, solver, state, options=None, max_summaries=None):
    """cx_minimize: TrustRegionMinimizer + LevenbergMarquardtStrategy with device-resident state.
    state: numpy array (updated copy returned) or DeviceArray (updated in place).
    Returns (state, summary dict, list of iteration summary dicts)."""
    lib = evaluator.lib
    if options is None:
        options = cx_minimizer_options()
        lib.cx_minimizer_default_options(ctypes.byref(options))
    cap = int(max_summaries if max_summaries is not None else options.max_num_iterations + 2)
    its = (cx_iteration_summary * cap)()
    summ = cx_minimizer_summary()
    if isinstance(state, DeviceArray):
        _check(lib.cx_minimize(evaluator._h, solver._h, ctypes.byref(options), _ptr(state), DEVICE,
                               ctypes.byref(summ), its, cap))
        out = state
    else:
        out = np.array(state, dtype=np.float64).copy()
        _check(lib.cx_minimize(evaluator._h, solver._h, ctypes.byref(options), _ptr(out), HOST, ctypes.byref(summ),
                               its, cap))
    n = min(cap, summ.num_iterations)
    return out, _summary_dict(summ), [_summary_dict(its[i]) for i in range(n)]


def sparse_cholesky_plan_host(num_cameras, cell_row, cell_col):
    """Host half of the tile-sparse Cholesky plan (no device): dict with camera_first_row, num_tile_rows, num_levels,
    num_tiles, num_tile_pair_updates, tile_row_level, tile_row_start, tile_cols."""
    lib = load_library()
    r = np.ascontiguousarray(cell_row, dtype=np.int32)
    c = np.ascontiguousarray(cell_col, dtype=np.int32)
    T, L = ctypes.c_int32(), ctypes.c_int32()
    nt, npairs = ctypes.c_int64(), ctypes.c_int64()
    first = np.zeros(num_cameras, dtype=np.int32)
    _check(lib.cx_sparse_cholesky_plan_host(int(num_cameras), _ptr(r), _ptr(c), ctypes.c_int64(r.size), _ptr(first), ctypes.byref(T),
                                            ctypes.byref(L), ctypes.byref(nt), ctypes.byref(npairs), None, None, 0, None, ctypes.c_int64(0)))
    level = np.zeros(T.value, dtype=np.int32)
    start = np.zeros(T.value + 1, dtype=np.int32)
    cols = np.zeros(nt.value, dtype=np.int32)
    _check(lib.cx_sparse_cholesky_plan_host(int(num_cameras), _ptr(r), _ptr(c), ctypes.c_int64(r.size), _ptr(first), ctypes.byref(T),
                                            ctypes.byref(L), ctypes.byref(nt), ctypes.byref(npairs), _ptr(level), _ptr(start), T.value,
                                            _ptr(cols), ctypes.c_int64(nt.value)))
    return dict(camera_first_row=first, num_tile_rows=T.value, num_levels=L.value, num_tiles=nt.value,
                num_tile_pair_updates=npairs.value, tile_row_level=level, tile_row_start=start, tile_cols=cols)


def sparse_cholesky_schedule_host(num_cameras, cell_row, cell_col, window, nranks=1, rank=0):
    """The update schedule of the tile-sparse factorisation under a window, checked from its definition (no device): dict with
    num_tile_rows, num_products, num_chains, longest_chain, violations (must be 0)."""
    lib = load_library()
    r = np.ascontiguousarray(cell_row, dtype=np.int32)
    c = np.ascontiguousarray(cell_col, dtype=np.int32)
    prod, chains, bad, longest = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
    T = lib.cx_sparse_cholesky_schedule_host(int(num_cameras), _ptr(r), _ptr(c), ctypes.c_int64(r.size), int(nranks), int(rank), int(window),
                                             ctypes.byref(prod), ctypes.byref(chains), ctypes.byref(longest), ctypes.byref(bad))
    if T < 0:
        _check(T)
    return dict(num_tile_rows=T, num_products=prod.value, num_chains=chains.value, longest_chain=longest.value, violations=bad.value)


def sparse_cholesky_distribution_host(num_cameras, cell_row, cell_col, nranks):
    """How the distributed factorisation divides the tile rows over nranks ranks (no device): dict with updates_per_rank,
    updates_replicated, tiles_replicated, owner (rank of every tile row, -1 = replicated top)."""
    lib = load_library()
    r = np.ascontiguousarray(cell_row, dtype=np.int32)
    c = np.ascontiguousarray(cell_col, dtype=np.int32)
    per = np.zeros(nranks, dtype=np.int64)
    rep, tiles = ctypes.c_int64(), ctypes.c_int64()
    T = lib.cx_sparse_cholesky_distribution_host(int(num_cameras), _ptr(r), _ptr(c), ctypes.c_int64(r.size), int(nranks), _ptr(per),
                                                 ctypes.byref(rep), ctypes.byref(tiles), None, 0)
    if T < 0:
        _check(T)
    owner = np.zeros(T, dtype=np.int32)
    _check(min(0, lib.cx_sparse_cholesky_distribution_host(int(num_cameras), _ptr(r), _ptr(c), ctypes.c_int64(r.size), int(nranks), _ptr(per),
                                                           ctypes.byref(rep), ctypes.byref(tiles), _ptr(owner), T)))
    return dict(updates_per_rank=per, updates_replicated=rep.value, tiles_replicated=tiles.value, owner=owner)


def schur_sparse_structure(A):
    """Cell list (row block, column block) of the block-sparse reduced camera matrix, InitStorage order."""
    lib = A.lib
    n = ctypes.c_int64()
    _check(lib.cx_schur_sparse_structure(A._h, ctypes.byref(n), None, None, ctypes.c_int64(0)))
    r = np.zeros(n.value, dtype=np.int32)
    c = np.zeros(n.value, dtype=np.int32)
    _check(lib.cx_schur_sparse_structure(A._h, ctypes.byref(n), _ptr(r), _ptr(c), ctypes.c_int64(n.value)))
    return r, c


def visibility_structure(A, preconditioner_type, clustering_type=CANONICAL_VIEWS):
    """(membership, num_clusters, cluster pairs [k, 2], block pairs [m, 2]) of a CLUSTER_* preconditioner."""
    lib = A.lib
    membership = np.zeros(A.bs.num_col_blocks - A.num_eliminate_blocks, dtype=np.int32)
    nc, ncp, nbp = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    _check(lib.cx_visibility_structure(A._h, int(preconditioner_type), int(clustering_type), _ptr(membership), ctypes.byref(nc),
                                       ctypes.byref(ncp), None, None, 0, ctypes.byref(nbp), None, None, ctypes.c_int64(0)))
    cp1 = np.zeros(ncp.value, dtype=np.int32)
    cp2 = np.zeros(ncp.value, dtype=np.int32)
    bp1 = np.zeros(nbp.value, dtype=np.int32)
    bp2 = np.zeros(nbp.value, dtype=np.int32)
    _check(lib.cx_visibility_structure(A._h, int(preconditioner_type), int(clustering_type), _ptr(membership), ctypes.byref(nc),
                                       ctypes.byref(ncp), _ptr(cp1), _ptr(cp2), ncp.value, ctypes.byref(nbp), _ptr(bp1),
                                       _ptr(bp2), ctypes.c_int64(nbp.value)))
    return membership, nc.value, np.stack([cp1, cp2], 1), np.stack([bp1, bp2], 1)


def visibility_clusters_host(bs, num_eliminate_blocks, preconditioner_type, clustering_type=CANONICAL_VIEWS):
    """(membership, num_clusters, cluster pairs [k, 2]) from the block structure alone -- runs without a GPU."""
    lib = load_library()
    membership = np.zeros(bs.num_col_blocks - int(num_eliminate_blocks), dtype=np.int32)
    nc, ncp = ctypes.c_int32(), ctypes.c_int32()
    _check(lib.cx_visibility_clusters_host(bs.c, int(num_eliminate_blocks), int(preconditioner_type), int(clustering_type),
                                           _ptr(membership), ctypes.byref(nc), ctypes.byref(ncp), None, None, 0))
    cp1 = np.zeros(ncp.value, dtype=np.int32)
    cp2 = np.zeros(ncp.value, dtype=np.int32)
    _check(lib.cx_visibility_clusters_host(bs.c, int(num_eliminate_blocks), int(preconditioner_type), int(clustering_type),
                                           _ptr(membership), ctypes.byref(nc), ctypes.byref(ncp), _ptr(cp1), _ptr(cp2), ncp.value))
    return membership, nc.value, np.stack([cp1, cp2], 1)


def schur_pair_lists_host(bs, num_eliminate_blocks, want_pairs=False):
    """Host half of the explicit-S gather assembly (no device): (cell_row, cell_col, num_pairs, num_items[, pair_rows])."""
    lib = load_library()
    nc, npairs, ni = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    _check(lib.cx_schur_pair_lists_host(bs.c, int(num_eliminate_blocks), ctypes.byref(nc), ctypes.byref(npairs), ctypes.byref(ni),
                                        None, None, ctypes.c_int64(0), None, ctypes.c_int64(0)))
    if nc.value < 0:
        return None, None, npairs.value, -1
    r, c = np.zeros(nc.value, dtype=np.int32), np.zeros(nc.value, dtype=np.int32)
    pr = np.zeros(2 * npairs.value if want_pairs else 0, dtype=np.int32)
    _check(lib.cx_schur_pair_lists_host(bs.c, int(num_eliminate_blocks), ctypes.byref(nc), ctypes.byref(npairs), ctypes.byref(ni),
                                        _ptr(r), _ptr(c), ctypes.c_int64(r.size), _ptr(pr) if want_pairs else None,
                                        ctypes.c_int64(pr.size)))
    return (r, c, npairs.value, ni.value, pr) if want_pairs else (r, c, npairs.value, ni.value)


def detect_structure(bs, num_eliminate_blocks):
    r, e, f = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    _check(load_library().cx_detect_structure(bs.c, int(num_eliminate_blocks), ctypes.byref(r), ctypes.byref(e),
                                              ctypes.byref(f)))
    return r.value, e.value, f.value


def stable_schur_ordering(num_cameras, num_points, camera_index, point_index):
    camera_index = np.ascontiguousarray(camera_index, dtype=np.int32)
    point_index = np.ascontiguousarray(point_index, dtype=np.int32)
    ordering = np.zeros(num_cameras + num_points, dtype=np.int32)
    k = ctypes.c_int32()
    _check(load_library().cx_stable_schur_ordering(int(num_cameras), int(num_points), ctypes.c_int64(camera_index.shape[0]),
                                                   _ptr(camera_index), _ptr(point_index), _ptr(ordering), ctypes.byref(k)))
    return ordering, k.value


def partition_points(bs, num_eliminate_blocks, nranks):
    bounds = np.zeros(nranks + 1, dtype=np.int32)
    _check(load_library().cx_partition_points(bs.c, int(num_eliminate_blocks), int(nranks), _ptr(bounds)))
    return bounds
