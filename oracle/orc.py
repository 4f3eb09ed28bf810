"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liborc.so (the CPU restatement of the reference
algorithms, see oracle/orc_common.h).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = ctypes.POINTER(ctypes.c_double)


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


class cx_summary(ctypes.Structure):
    _fields_ = [
        ("residual_norm", ctypes.c_double),
        ("num_iterations", ctypes.c_int32),
        ("termination_type", ctypes.c_int32),
        ("message", ctypes.c_char * 256),
    ]


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


EVALUATE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p,
                               ctypes.c_int)
VEC_OUT_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, c_double_p)
VEC_IN_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, c_double_p)
MULT_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, c_double_p, c_double_p)
SOLVE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_double, c_double_p,
                            ctypes.POINTER(ctypes.c_int))


class orc_min_problem(ctypes.Structure):
    _fields_ = [("num_parameters", ctypes.c_int32), ("num_residuals", ctypes.c_int32), ("user", ctypes.c_void_p),
                ("evaluate", EVALUATE_FN), ("squared_column_norm", VEC_OUT_FN), ("scale_columns", VEC_IN_FN),
                ("right_multiply", MULT_FN), ("solve", SOLVE_FN), ("num_effective_parameters", ctypes.c_int32),
                ("plus", ctypes.c_void_p)]

DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR = 0, 1, 2, 3
IDENTITY, JACOBI, SCHUR_JACOBI, SCHUR_POWER_SERIES_EXPANSION, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL = 0, 1, 2, 3, 4, 5
CANONICAL_VIEWS, SINGLE_LINKAGE = 0, 1
SUCCESS, NO_CONVERGENCE, FAILURE, FATAL_ERROR = 0, 1, 2, 3

ALLREDUCE_FN = ctypes.CFUNCTYPE(None, c_double_p, ctypes.c_int64, ctypes.c_void_p)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])


def use_native_build():
    """Build liborc_native.so (-O3 -march=native) on THIS machine and use it from now on: for the CPU baseline,
    which is timed on the host it runs on.  Falls back to the portable liborc.so when the build fails.
    Returns the compiler flags in use."""
    global _LIB
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liborc_native.so"])
        _LIB = None
        _LIB_PATH[0] = os.path.join(_HERE, "liborc_native.so")
        lib()
        return "-O3 -march=native -fopenmp (built on this host)"
    except Exception:
        _LIB = None
        _LIB_PATH[0] = os.path.join(_HERE, "liborc.so")
        return "-O3 -march=x86-64-v3 -fopenmp"


_LIB_PATH = [os.path.join(_HERE, "liborc.so")]


def lib():
    global _LIB
    if _LIB is None:
        path = _LIB_PATH[0]
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_to_crs.restype = ctypes.c_int64
        _LIB.orc_stable_schur_ordering.restype = ctypes.c_int
        _LIB.orc_last_solve_seconds.restype = ctypes.c_double
        _LIB.orc_schur_sparse_structure.restype = ctypes.c_int64
        _LIB.orc_visibility_structure.restype = ctypes.c_int64
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)   # keeps the (possibly temporary) array alive for the call


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def schur_sparse_structure(bs, num_eliminate_blocks):
    n = lib().orc_schur_sparse_structure(bs.c, int(num_eliminate_blocks), None, None, ctypes.c_int64(0))
    r = np.zeros(n, dtype=np.int32)
    c = np.zeros(n, dtype=np.int32)
    lib().orc_schur_sparse_structure(bs.c, int(num_eliminate_blocks), _p(r), _p(c), ctypes.c_int64(n))
    return r, c


def _edges(edges):
    u = np.ascontiguousarray([e[0] for e in edges], dtype=np.int32)
    v = np.ascontiguousarray([e[1] for e in edges], dtype=np.int32)
    w = np.ascontiguousarray([e[2] for e in edges], dtype=np.float64)
    return u, v, w


def schur_complement_graph(bs, num_eliminate_blocks):
    """CreateSchurComplementGraph: list of (u, v, weight), u <= v, self edges included."""
    n = lib().orc_schur_complement_graph(bs.c, int(num_eliminate_blocks), None, None, None, 0)
    u = np.zeros(n, dtype=np.int32)
    v = np.zeros(n, dtype=np.int32)
    w = np.zeros(n)
    lib().orc_schur_complement_graph(bs.c, int(num_eliminate_blocks), _p(u), _p(v), _p(w), n)
    return list(zip(u.tolist(), v.tolist(), w.tolist()))


def canonical_views(n, vertex_weights, edges, min_views=3, size_penalty_weight=5.75, similarity_penalty_weight=100.0,
                    view_score_weight=0.0):
    u, v, w = _edges(edges)
    vw = _f64(vertex_weights)
    centers = np.zeros(n, dtype=np.int32)
    membership = np.zeros(n, dtype=np.int32)
    k = lib().orc_canonical_views(int(n), _p(vw), len(u), _p(u), _p(v), _p(w), int(min_views),
                                  ctypes.c_double(size_penalty_weight), ctypes.c_double(similarity_penalty_weight),
                                  ctypes.c_double(view_score_weight), _p(centers), _p(membership))
    return centers[:k].tolist(), membership.tolist()


def single_linkage(n, edges, min_similarity=0.99):
    u, v, w = _edges(edges)
    membership = np.zeros(n, dtype=np.int32)
    k = lib().orc_single_linkage(int(n), len(u), _p(u), _p(v), _p(w), ctypes.c_double(min_similarity), _p(membership))
    return k, membership.tolist()


def degree2_forest(n, edges):
    u, v, w = _edges(edges)
    fu = np.zeros(max(n, 1), dtype=np.int32)
    fv = np.zeros(max(n, 1), dtype=np.int32)
    k = lib().orc_degree2_forest(int(n), len(u), _p(u), _p(v), _p(w), _p(fu), _p(fv))
    return list(zip(fu[:k].tolist(), fv[:k].tolist()))


def visibility_structure(bs, num_eliminate_blocks, preconditioner_type, clustering_type=0):
    """membership, num_clusters, cluster pairs, block pairs of the CLUSTER_* preconditioner."""
    nf = bs.num_col_blocks - num_eliminate_blocks
    membership = np.zeros(nf, dtype=np.int32)
    nc = ctypes.c_int32(0)
    ncp = ctypes.c_int32(0)
    nbp = lib().orc_visibility_structure(bs.c, int(num_eliminate_blocks), int(preconditioner_type), int(clustering_type),
                                         _p(membership), ctypes.byref(nc), ctypes.byref(ncp), None, None, 0, None, None,
                                         ctypes.c_int64(0))
    cp1 = np.zeros(ncp.value, dtype=np.int32)
    cp2 = np.zeros(ncp.value, dtype=np.int32)
    bp1 = np.zeros(nbp, dtype=np.int32)
    bp2 = np.zeros(nbp, dtype=np.int32)
    lib().orc_visibility_structure(bs.c, int(num_eliminate_blocks), int(preconditioner_type), int(clustering_type),
                                   _p(membership), ctypes.byref(nc), ctypes.byref(ncp), _p(cp1), _p(cp2), ncp.value,
                                   _p(bp1), _p(bp2), ctypes.c_int64(nbp))
    return membership, nc.value, np.stack([cp1, cp2], 1), np.stack([bp1, bp2], 1)


def make_options(type=ITERATIVE_SCHUR, preconditioner_type=JACOBI, num_eliminate_blocks=0,
                 min_num_iterations=0, max_num_iterations=500, residual_reset_period=10,
                 max_num_spse_iterations=5, use_spse_initialization=0, spse_tolerance=0.1,
                 use_explicit_schur_complement=0, visibility_clustering_type=0, use_mixed_precision_solves=0,
                 max_num_refinement_iterations=0):
    o = cx_solver_options()
    o.type = type
    o.preconditioner_type = preconditioner_type
    o.min_num_iterations = min_num_iterations
    o.max_num_iterations = max_num_iterations
    o.residual_reset_period = residual_reset_period
    o.num_eliminate_blocks = num_eliminate_blocks
    o.max_num_spse_iterations = max_num_spse_iterations
    o.use_spse_initialization = use_spse_initialization
    o.spse_tolerance = spse_tolerance
    o.use_explicit_schur_complement = use_explicit_schur_complement
    o.visibility_clustering_type = visibility_clustering_type
    o.use_mixed_precision_solves = use_mixed_precision_solves       # DENSE_SCHUR / SPARSE_SCHUR: float factor (orc_schur.cpp)
    o.max_num_refinement_iterations = max_num_refinement_iterations
    return o


def right_multiply(bs, values, x, y=None):
    values, x = _f64(values), _f64(x)
    y = np.zeros(bs.num_rows) if y is None else _f64(y).copy()
    lib().orc_right_multiply(bs.c, _p(values), _p(x), _p(y))
    return y


def left_multiply(bs, values, x, y=None):
    values, x = _f64(values), _f64(x)
    y = np.zeros(bs.num_cols) if y is None else _f64(y).copy()
    lib().orc_left_multiply(bs.c, _p(values), _p(x), _p(y))
    return y


def squared_column_norm(bs, values):
    values = _f64(values)
    x = np.zeros(bs.num_cols)
    lib().orc_squared_column_norm(bs.c, _p(values), _p(x))
    return x


def scale_columns(bs, values, scale):
    v = _f64(values).copy()
    lib().orc_scale_columns(bs.c, _p(v), _p(_f64(scale)))
    return v


def to_crs(bs, values, transpose=False):
    values = _f64(values)
    n = bs.num_cols if transpose else bs.num_rows
    nnz = bs.num_nonzeros
    rows = np.zeros(n + 1, dtype=np.int32)
    cols = np.zeros(nnz, dtype=np.int32)
    vals = np.zeros(nnz)
    got = lib().orc_to_crs(bs.c, _p(values), int(transpose), _p(rows), _p(cols), _p(vals))
    assert got == nnz
    return rows, cols, vals


def detect_structure(bs, num_eliminate_blocks):
    r, e, f = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib().orc_detect_structure(bs.c, int(num_eliminate_blocks), ctypes.byref(r), ctypes.byref(e), ctypes.byref(f))
    return r.value, e.value, f.value


def _num_cols_f(bs, nelim):
    return int(bs.col_blocks["size"][nelim:].sum())


def schur_eliminate_dense(bs, values, b, D, nelim):
    n = _num_cols_f(bs, nelim)
    lhs = np.zeros((n, n))
    rhs = np.zeros(n) if b is not None else None
    lib().orc_schur_eliminate_dense(bs.c, _p(_f64(values)), _p(_f64(b)), _p(_f64(D)), int(nelim), _p(lhs), _p(rhs))
    return lhs, rhs


def schur_eliminate_diagonal(bs, values, D, nelim):
    sizes = bs.col_blocks["size"][nelim:].astype(np.int64)
    blocks = np.zeros(int((sizes * sizes).sum()))
    lib().orc_schur_eliminate_diagonal(bs.c, _p(_f64(values)), _p(_f64(D)), int(nelim), _p(blocks))
    return blocks


def schur_back_substitute(bs, values, b, D, nelim, z):
    x = np.zeros(bs.num_cols)
    lib().orc_schur_back_substitute(bs.c, _p(_f64(values)), _p(_f64(b)), _p(_f64(D)), int(nelim), _p(_f64(z)), _p(x))
    return x


def dense_cholesky_solve(lhs, rhs):
    a = _f64(lhs).copy()
    n = a.shape[0]
    x = np.zeros(n)
    t = lib().orc_dense_cholesky_solve(n, _p(a), _p(_f64(rhs)), _p(x))
    return x, t


def dense_cholesky_solve_refined(lhs, rhs, use_float, refinements):
    """FloatEigenDenseCholesky / RefinedDenseCholesky (dense_cholesky.cc:180-204, 322-347; iterative_refiner.cc:83-99)."""
    a = _f64(lhs)
    n = a.shape[0]
    x = np.zeros(n)
    t = lib().orc_dense_cholesky_solve_refined(n, _p(a), _p(_f64(rhs)), _p(x), int(bool(use_float)), int(refinements))
    return x, t


def implicit_schur_multiply(bs, values, D, b, nelim, x):
    n = _num_cols_f(bs, nelim)
    y = np.zeros(n)
    rhs = np.zeros(n) if b is not None else None
    lib().orc_implicit_schur_multiply(bs.c, _p(_f64(values)), _p(_f64(D)), _p(_f64(b)), int(nelim), _p(_f64(x)), _p(y), _p(rhs))
    return y, rhs


def block_diagonal_inverses(bs, values, D, nelim, want_ftf=True):
    es = bs.col_blocks["size"][:nelim].astype(np.int64)
    fs = bs.col_blocks["size"][nelim:].astype(np.int64)
    ete = np.zeros(int((es * es).sum()))
    ftf = np.zeros(int((fs * fs).sum())) if want_ftf else None
    lib().orc_block_diagonal_inverses(bs.c, _p(_f64(values)), _p(_f64(D)), int(nelim), _p(ete), _p(ftf))
    return ete, ftf


def solve(bs, values, b, D, options, r_tolerance=-1.0, q_tolerance=0.0, allreduce=None):
    x = np.zeros(bs.num_cols)
    s = cx_summary()
    args = [bs.c, _p(_f64(values)), _p(_f64(b)), _p(_f64(D)), ctypes.byref(options),
            ctypes.c_double(r_tolerance), ctypes.c_double(q_tolerance), _p(x), ctypes.byref(s)]
    if allreduce is None:
        rc = lib().orc_solve(*args)
    else:
        def _cb(buf, n, _user):
            a = np.ctypeslib.as_array(buf, shape=(n,))
            allreduce(a)
        cb = ALLREDUCE_FN(_cb)
        rc = lib().orc_solve_sharded(*args, cb, None)
    assert rc == 0
    return x, s


def sparse_schur_stats():
    """Of the last SPARSE_SCHUR solve: seconds of analysis / eliminate / factor / triangular solves, blocks of L,
    factor flops, elimination-tree heights, cells of S."""
    out = np.zeros(8)
    lib().orc_sparse_schur_stats(_p(out))
    return dict(zip(["analyze_s", "eliminate_s", "factor_s", "solve_s", "factor_blocks", "factor_flops", "etree_heights",
                     "s_cells"], out.tolist()))


def last_solve_seconds():
    return lib().orc_last_solve_seconds()


def cg_dense(A, b, x0, max_num_iterations, r_tolerance=-1.0, q_tolerance=0.0, min_num_iterations=0,
             residual_reset_period=10):
    A = _f64(A)
    x = _f64(x0).copy()
    s = cx_summary()
    lib().orc_cg_dense(A.shape[0], _p(A), _p(_f64(b)), _p(x), int(min_num_iterations),
                       int(max_num_iterations), int(residual_reset_period), ctypes.c_double(r_tolerance), ctypes.c_double(q_tolerance), ctypes.byref(s))
    return x, s


def snavely(camera, point, obs, jacobians=True):
    r = np.zeros(2)
    jc = np.zeros((2, 9)) if jacobians else None
    jp = np.zeros((2, 3)) if jacobians else None
    lib().orc_snavely(_p(_f64(camera)), _p(_f64(point)), _p(_f64(obs)), _p(r), _p(jc), _p(jp))
    return r, jc, jp


def angle_axis_rotate_point(aa, pt):
    out = np.zeros(3)
    lib().orc_angle_axis_rotate_point(_p(_f64(aa)), _p(_f64(pt)), _p(out))
    return out


def bal_residual_order(num_points, point_index):
    point_index = np.ascontiguousarray(point_index, dtype=np.int32)
    order = np.zeros(point_index.shape[0], dtype=np.int64)
    lib().orc_bal_residual_order(int(num_points), ctypes.c_int64(point_index.shape[0]), _p(point_index), _p(order))
    return order


def stable_schur_ordering(num_cameras, num_points, camera_index, point_index):
    camera_index = np.ascontiguousarray(camera_index, dtype=np.int32)
    point_index = np.ascontiguousarray(point_index, dtype=np.int32)
    ordering = np.zeros(num_cameras + num_points, dtype=np.int32)
    k = lib().orc_stable_schur_ordering(int(num_cameras), int(num_points), ctypes.c_int64(camera_index.shape[0]),
                                        _p(camera_index), _p(point_index), _p(ordering))
    return ordering, k


def bal_structure_arrays(num_cameras, num_points, camera_index, point_index, order):
    """Returns (row_blocks, col_blocks, row_cell_begin, cells) numpy arrays."""
    from numpy import zeros
    O = int(camera_index.shape[0])
    blk = np.dtype([("size", np.int32), ("position", np.int32)])
    cel = np.dtype([("block_id", np.int32), ("position", np.int32)])
    rb = zeros(O, dtype=blk)
    cb = zeros(num_points + num_cameras, dtype=blk)
    rcb = zeros(O + 1, dtype=np.int32)
    cells = zeros(2 * O, dtype=cel)
    lib().orc_bal_structure(int(num_cameras), int(num_points), ctypes.c_int64(O),
                            _p(np.ascontiguousarray(camera_index, dtype=np.int32)),
                            _p(np.ascontiguousarray(point_index, dtype=np.int32)),
                            _p(np.ascontiguousarray(order, dtype=np.int64)), _p(rb), _p(cb), _p(rcb), _p(cells))
    return rb, cb, rcb, cells


def loss_evaluate(loss_type, a, b, s):
    rho = np.zeros(3)
    lib().orc_loss_evaluate(int(loss_type), ctypes.c_double(a), ctypes.c_double(b), ctypes.c_double(s), _p(rho))
    return rho


def corrector_apply(sq_norm, rho, residuals, jacobian=None):
    """Returns (corrected residuals, corrected jacobian)."""
    r = np.array(residuals, dtype=np.float64).copy()
    j = None if jacobian is None else np.array(jacobian, dtype=np.float64).copy()
    ncols = 0 if j is None else j.shape[1]
    lib().orc_corrector_apply(ctypes.c_double(sq_norm), _p(_f64(rho)), int(r.shape[0]), int(ncols), _p(r), _p(j))
    return r, j


def bal_evaluate(bs, num_cameras, num_points, camera_index, point_index, observations, order, state,
                 want_residuals=True, want_gradient=True, want_jacobian=True, loss=None):
    if loss is not None:
        O = int(camera_index.shape[0])
        cost = ctypes.c_double()
        res = np.zeros(2 * O) if want_residuals else None
        grad = np.zeros(3 * num_points + 9 * num_cameras) if want_gradient else None
        vals = np.zeros(24 * O) if want_jacobian else None
        ltype, la, lb = loss
        lib().orc_bal_evaluate_robust(bs.c, int(num_cameras), int(num_points), ctypes.c_int64(O),
                                      _p(np.ascontiguousarray(camera_index, dtype=np.int32)),
                                      _p(np.ascontiguousarray(point_index, dtype=np.int32)),
                                      _p(_f64(observations)), _p(np.ascontiguousarray(order, dtype=np.int64)),
                                      _p(_f64(state)), int(ltype), ctypes.c_double(la), ctypes.c_double(lb),
                                      ctypes.byref(cost), _p(res), _p(grad), _p(vals))
        return cost.value, res, grad, vals
    O = int(camera_index.shape[0])
    cost = ctypes.c_double()
    res = np.zeros(2 * O) if want_residuals else None
    grad = np.zeros(3 * num_points + 9 * num_cameras) if want_gradient else None
    vals = np.zeros(24 * O) if want_jacobian else None
    lib().orc_bal_evaluate(bs.c, int(num_cameras), int(num_points), ctypes.c_int64(O),
                           _p(np.ascontiguousarray(camera_index, dtype=np.int32)),
                           _p(np.ascontiguousarray(point_index, dtype=np.int32)),
                           _p(_f64(observations)), _p(np.ascontiguousarray(order, dtype=np.int64)),
                           _p(_f64(state)), ctypes.byref(cost), _p(res), _p(grad), _p(vals))
    return cost.value, res, grad, vals


def minimize_dense(fun, x0, options=None):
    """orc_minimize on a small dense problem: fun(x) -> (residuals[m], jacobian[m, n]).  The linear
    solver stands in for DenseQRSolver (dense_qr_solver.cc): least squares on [J; diag(D)]."""
    x0 = np.array(x0, dtype=np.float64).copy()
    n = x0.size
    r0, _ = fun(x0)
    m = int(np.asarray(r0).size)
    st = {"J": np.zeros((m, n))}

    def evaluate(_u, x, cost, residuals, gradient, want_j):
        xv = np.ctypeslib.as_array(x, (n,))
        r, J = fun(xv.copy())
        r = np.asarray(r, dtype=np.float64)
        cost[0] = 0.5 * float(r @ r)
        if residuals:
            np.ctypeslib.as_array(residuals, (m,))[:] = r
        if gradient:
            np.ctypeslib.as_array(gradient, (n,))[:] = np.asarray(J).T @ r
        if want_j:
            st["J"] = np.array(J, dtype=np.float64)
        return 1

    def sqnorm(_u, out):
        np.ctypeslib.as_array(out, (n,))[:] = (st["J"] ** 2).sum(axis=0)

    def scale(_u, sc):
        st["J"] = st["J"] * np.ctypeslib.as_array(sc, (n,))[None, :]

    def mult(_u, x, y):
        np.ctypeslib.as_array(y, (m,))[:] += st["J"] @ np.ctypeslib.as_array(x, (n,))

    def solve(_u, b, D, _q, x, iters):
        A = np.vstack([st["J"], np.diag(np.ctypeslib.as_array(D, (n,)))])
        rhs = np.concatenate([np.ctypeslib.as_array(b, (m,)), np.zeros(n)])
        np.ctypeslib.as_array(x, (n,))[:] = np.linalg.lstsq(A, rhs, rcond=None)[0]
        iters[0] = 1
        return SUCCESS

    cbs = (EVALUATE_FN(evaluate), VEC_OUT_FN(sqnorm), VEC_IN_FN(scale), MULT_FN(mult), SOLVE_FN(solve))
    prob = orc_min_problem(n, m, None, *cbs, n, None)
    options = options or minimizer_options()
    cap = options.max_num_iterations + 2
    its = (cx_iteration_summary * cap)()
    summ = cx_minimizer_summary()
    lib().orc_minimize(ctypes.byref(prob), ctypes.byref(options), _p(x0), ctypes.byref(summ), its, cap)
    k = min(cap, summ.num_iterations)
    return x0, _summary_dict(summ), [_summary_dict(its[i]) for i in range(k)]


ANGLE_AXIS, QUATERNION_MANIFOLD = 0, 1


def angle_axis_to_rotation_matrix(aa):
    """rotation.h:452-494; returns the 9 entries COLUMN-major, as the reference's tests read them."""
    R = np.zeros(9)
    lib().orc_angle_axis_to_rotation_matrix(_p(_f64(aa)), _p(R))
    return R


def angle_axis_to_quaternion(aa):
    q = np.zeros(4)
    lib().orc_angle_axis_to_quaternion(_p(_f64(aa)), _p(q))
    return q


def quaternion_to_angle_axis(q):
    aa = np.zeros(3)
    lib().orc_quaternion_to_angle_axis(_p(_f64(q)), _p(aa))
    return aa


def quaternion_plus(x, delta):
    out = np.zeros(4)
    lib().orc_quaternion_plus(_p(_f64(x)), _p(_f64(delta)), _p(out))
    return out


def quaternion_plus_jacobian(x):
    out = np.zeros(12)
    lib().orc_quaternion_plus_jacobian(_p(_f64(x)), _p(out))
    return out.reshape(4, 3)


def snavely_quaternion(camera10, point, obs, want_jacobian=True):
    res, jc, jp = np.zeros(2), np.zeros(18), np.zeros(6)
    lib().orc_snavely_quaternion(_p(_f64(camera10)), _p(_f64(point)), _p(_f64(obs)), _p(res),
                                 _p(jc) if want_jacobian else None, _p(jp) if want_jacobian else None)
    return res, jc.reshape(2, 9), jp.reshape(2, 3)


def bal_plus(num_cameras, num_points, camera_model, x, delta):
    out = np.zeros_like(_f64(x))
    lib().orc_bal_plus(int(num_cameras), int(num_points), int(camera_model), _p(_f64(x)), _p(_f64(delta)), _p(out))
    return out


def bal_evaluate_model(bs, num_cameras, num_points, camera_index, point_index, observations, order, state,
                       camera_model, loss=None, want_jacobian=True):
    O = int(camera_index.shape[0])
    cost = ctypes.c_double()
    res = np.zeros(2 * O)
    grad = np.zeros(3 * num_points + 9 * num_cameras)
    vals = np.zeros(24 * O) if want_jacobian else None
    ltype, la, lb = loss if loss is not None else (0, 0.0, 0.0)
    lib().orc_bal_evaluate_model(bs.c, int(num_cameras), int(num_points), ctypes.c_int64(O),
                                 _p(np.ascontiguousarray(camera_index, dtype=np.int32)),
                                 _p(np.ascontiguousarray(point_index, dtype=np.int32)), _p(_f64(observations)),
                                 _p(np.ascontiguousarray(order, dtype=np.int64)), _p(_f64(state)), int(camera_model),
                                 int(ltype), ctypes.c_double(la), ctypes.c_double(lb), ctypes.byref(cost), _p(res),
                                 _p(grad), _p(vals))
    return cost.value, res, grad, vals


def minimize_bal(num_cameras, num_points, camera_index, point_index, observations, state, solver_options,
                 options=None, loss=None, camera_model=0):
    state = np.array(state, dtype=np.float64).copy()
    options = options or minimizer_options()
    cap = options.max_num_iterations + 2
    its = (cx_iteration_summary * cap)()
    summ = cx_minimizer_summary()
    ltype, la, lb = loss if loss is not None else (0, 0.0, 0.0)
    O = int(np.asarray(camera_index).shape[0])
    lib().orc_minimize_bal(int(num_cameras), int(num_points), ctypes.c_int64(O),
                           _p(np.ascontiguousarray(camera_index, dtype=np.int32)),
                           _p(np.ascontiguousarray(point_index, dtype=np.int32)), _p(_f64(observations)),
                           int(camera_model), int(ltype), ctypes.c_double(la), ctypes.c_double(lb), ctypes.byref(solver_options),
                           ctypes.byref(options), _p(state), ctypes.byref(summ), its, cap)
    k = min(cap, summ.num_iterations)
    return state, _summary_dict(summ), [_summary_dict(its[i]) for i in range(k)]
