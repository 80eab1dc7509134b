"""ctypes binding of libmiba.so (include/miba.h) — plumbing for tests and bench.py.

The product is the C-ABI library; this module only marshals numpy arrays into it.  There is no
CPU fallback: if libmiba.so is missing, or no GPU is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIBA_LIB", os.path.join(_HERE, "libmiba.so"))   # MIBA_LIB: kernel-variant experiments

(DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR,
 ITERATIVE_SCHUR, CGNR) = range(7)
PRECOND_JACOBI, PRECOND_SCHUR_JACOBI = 0, 1
MODEL_BAL, MODEL_REFERENCE = 0, 1   # miba_problem.camera_model
LOSS_NONE, LOSS_HUBER, LOSS_CAUCHY = 0, 1, 2   # options.loss_type (ceres::HuberLoss / CauchyLoss with options.loss_scale)
CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE = range(5)
K_EVAL_JAC, K_EVAL_COST, K_SCHUR_INIT, K_SCHUR_MATVEC, K_BACKSUB = range(5)
K_SCHUR_MATVEC_SKIPPED = 5   # timed K3 launches that found the PCG finished and returned at once
K_SCHUR_INIT_CAM = 6         # K2b camera side of the Schur init (K_SCHUR_INIT is the point side K2a)
K_CAM_REDUCE = 7             # camera-side sum of a product, sampled like K3
K_COUNT = 8

EXPORTS = [
    "miba_default_options", "miba_create", "miba_create_multi", "miba_destroy", "miba_last_error", "miba_comm_unique_id",
    "miba_set_problem", "miba_set_state", "miba_get_state", "miba_evaluate", "miba_solve",
    "miba_get_iterations", "miba_linear_step", "miba_get_kernel_times", "miba_reset_kernel_times",
    "miba_plan_shards", "miba_plan_point_owner", "miba_plan_product_stats", "miba_plan_tile_order", "miba_time_matvec", "miba_set_constant_blocks",
    "miba_get_reduced_system", "miba_dense_spd_solve", "miba_dense_spd_solve_static",
]
STATIC_SOLVE_MAX_N = 1024   # MIBA_STATIC_SOLVE_MAX_N of include/miba.h


class MibaError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
                ("log_level", C.c_int32), ("comm_unique_id", C.c_void_p)]


class Problem(C.Structure):
    _fields_ = [("n_cam", C.c_int32), ("n_pt", C.c_int32), ("n_obs", C.c_int64),
                ("R", C.POINTER(C.c_double)), ("T", C.POINTER(C.c_double)),
                ("K", C.POINTER(C.c_double)), ("X", C.POINTER(C.c_double)),
                ("obs_cam", C.POINTER(C.c_int32)), ("obs_pt", C.POINTER(C.c_int32)),
                ("obs_xy", C.POINTER(C.c_double)), ("camera_model", C.c_int32), ("reserved", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("linear_solver_type", C.c_int32),
                ("jacobi_scaling", C.c_int32), ("max_linear_solver_iterations", C.c_int32),
                ("min_linear_solver_iterations", C.c_int32),
                ("max_num_consecutive_invalid_steps", C.c_int32),
                ("update_state_every_iteration", C.c_int32),
                ("minimizer_progress_to_stdout", C.c_int32), ("pcg_check_every", C.c_int32),
                ("preconditioner_type", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("eta", C.c_double), ("eta_exact", C.c_double),
                ("max_solver_time_in_seconds", C.c_double),
                ("loss_type", C.c_int32), ("async_callbacks", C.c_int32), ("loss_scale", C.c_double)]


class Iteration(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("step_is_valid", C.c_int32),
                ("step_is_successful", C.c_int32), ("linear_solver_iterations", C.c_int32),
                ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
                ("gradient_norm", C.c_double), ("step_norm", C.c_double),
                ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double),
                ("eta", C.c_double), ("iteration_time_in_seconds", C.c_double),
                ("cumulative_time_in_seconds", C.c_double), ("model_cost_change", C.c_double)]


class Summary(C.Structure):
    _fields_ = [("termination_type", C.c_int32), ("num_iterations", C.c_int32),
                ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
                ("num_linear_solver_iterations", C.c_int32), ("num_residual_blocks", C.c_int32),
                ("num_active_cameras", C.c_int32), ("num_active_points", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("total_time_in_seconds", C.c_double), ("linear_solver_time_in_seconds", C.c_double),
                ("jacobian_time_in_seconds", C.c_double), ("residual_time_in_seconds", C.c_double),
                ("message", C.c_char * 128), ("exact_linear_solver", C.c_int32), ("reserved", C.c_int32)]


class KernelTimes(C.Structure):
    _fields_ = [("launches", C.c_int64 * K_COUNT), ("total_ms", C.c_double * K_COUNT),
                ("n_slots", C.c_int64), ("n_obs_local", C.c_int64), ("n_pt_local", C.c_int64),
                ("n_tiles", C.c_int64), ("n_cam", C.c_int32), ("product_layout", C.c_int32),
                ("product_clock_mhz", C.c_double), ("pcg_launches", C.c_int32), ("pcg_exchange", C.c_int32)]


CALLBACK = C.CFUNCTYPE(C.c_int, C.POINTER(Iteration), C.c_void_p)

_lib = None


def lib():
    """Loads libmiba.so; raises MibaError when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MibaError("libmiba.so is missing (%s): run __graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.miba_last_error.restype = C.c_char_p
        _lib.miba_last_error.argtypes = [C.c_void_p]
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def default_options(**kw):
    o = Options()
    lib().miba_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().miba_comm_unique_id(buf)
    if rc != 0:
        raise MibaError("miba_comm_unique_id rc=%d: %s" % (rc, (lib().miba_last_error(None) or b"").decode()))
    return buf.raw


def split_cams(cams):
    """(n,9) [w t f k1 k2] -> reference arrays R (n,3), T (n,3), K (n,6) (src/BalDataloader.cpp:143-148)."""
    cams = np.asarray(cams, dtype=np.float64).reshape(-1, 9)
    R = np.ascontiguousarray(cams[:, 0:3])
    T = np.ascontiguousarray(cams[:, 3:6])
    K = np.zeros((cams.shape[0], 6))
    K[:, 0] = cams[:, 6]; K[:, 1] = cams[:, 6]; K[:, 4] = cams[:, 7]; K[:, 5] = cams[:, 8]
    return R, T, K


def join_cams(R, T, K):
    return np.concatenate([R, T, K[:, 0:1], K[:, 4:5], K[:, 5:6]], axis=1)


def plan_shards(n_cam, n_pt, obs_cam, obs_pt, obs_xy, world_size, X=None):
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
    obs_pt = np.ascontiguousarray(obs_pt, dtype=np.int32)
    obs_xy = np.ascontiguousarray(obs_xy, dtype=np.float64)
    p = Problem(n_cam, n_pt, obs_cam.size, None, None, None, _dp(X) if X is not None else None,
                _ip(obs_cam), _ip(obs_pt), _dp(obs_xy))
    nv, nt, ns = C.c_int64(), C.c_int64(), C.c_int64()
    tiles = (C.c_int64 * world_size)()
    obs = (C.c_int64 * world_size)()
    rc = lib().miba_plan_shards(C.byref(p), C.c_int32(world_size), C.byref(nv), C.byref(nt), C.byref(ns), tiles, obs)
    if rc != 0:
        raise MibaError("miba_plan_shards rc=%d: %s" % (rc, (lib().miba_last_error(None) or b"").decode()))
    return dict(n_valid=nv.value, n_tiles=nt.value, n_slots=ns.value, tiles_per_rank=list(tiles), obs_per_rank=list(obs))


def plan_point_owner(n_cam, n_pt, obs_cam, obs_pt, obs_xy, world_size):
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
    obs_pt = np.ascontiguousarray(obs_pt, dtype=np.int32)
    obs_xy = np.ascontiguousarray(obs_xy, dtype=np.float64)
    p = Problem(n_cam, n_pt, obs_cam.size, None, None, None, None, _ip(obs_cam), _ip(obs_pt), _dp(obs_xy))
    owner = np.zeros(n_pt, dtype=np.int32)
    rc = lib().miba_plan_point_owner(C.byref(p), C.c_int32(world_size), _ip(owner))
    if rc != 0:
        raise MibaError("miba_plan_point_owner rc=%d: %s" % (rc, (lib().miba_last_error(None) or b"").decode()))
    return owner


def plan_tile_order(n_rows, edges_u, edges_v, n_arcs, nested=True):
    """Host-only: elimination order of a tile graph as the exact solve's symbolic phase computes it (miba_plan_tile_order)."""
    eu = np.ascontiguousarray(edges_u, dtype=np.int32)
    ev = np.ascontiguousarray(edges_v, dtype=np.int32)
    pos = np.zeros(n_rows, dtype=np.int32)
    arc = np.zeros(n_arcs + 1, dtype=np.int32)
    gs, gl = np.zeros(2 * n_arcs + 1, dtype=np.int32), np.zeros(2 * n_arcs, dtype=np.int32)
    na, fill, cost, ng = C.c_int32(0), C.c_int64(0), C.c_double(0.0), C.c_int32(0)
    rc = lib().miba_plan_tile_order(C.c_int32(n_rows), _ip(eu), _ip(ev), C.c_int64(eu.size), C.c_int32(n_arcs), _ip(pos), _ip(arc),
                                    C.byref(na), C.byref(fill), C.byref(cost), C.c_int32(1 if nested else 0), _ip(gs), _ip(gl), C.byref(ng))
    if rc != 0:
        raise MibaError("miba_plan_tile_order rc=%d: %s" % (rc, (lib().miba_last_error(None) or b"").decode()))
    n_g = min(ng.value, 2 * n_arcs)
    return dict(pos=pos, arc_start=[int(a) for a in arc[:na.value + 1]], n_arcs=na.value, fill_tiles=fill.value, cost_us=cost.value,
                group_start=[int(a) for a in gs[:n_g + 1]], group_level=[int(a) for a in gl[:n_g]])


def plan_product_stats(n_cam, n_pt, obs_cam, obs_pt, obs_xy):
    """Host-only: the product kernel's camera-major layout of one shard, checked and priced (miba_plan_product_stats)."""
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
    obs_pt = np.ascontiguousarray(obs_pt, dtype=np.int32)
    obs_xy = np.ascontiguousarray(obs_xy, dtype=np.float64)
    p = Problem(n_cam, n_pt, obs_cam.size, None, None, None, None, _ip(obs_cam), _ip(obs_pt), _dp(obs_xy))
    out = np.zeros(10)
    rc = lib().miba_plan_product_stats(C.byref(p), _dp(out))
    if rc != 0:
        raise MibaError("miba_plan_product_stats rc=%d: %s" % (rc, (lib().miba_last_error(None) or b"").decode()))
    return dict(ok=bool(out[0]), n_obs=int(out[1]), n_entries=int(out[2]), n_sub=int(out[3]), n_seg=int(out[4]),
                pt_mult=float(out[5]), cam_mult=float(out[6]), n_giant_obs=int(out[7]),
                rd_extra=float(out[8]), at_extra=float(out[9]))


class Solver:
    """One handle = one GPU (one shard), or with devices=[...] one handle for several GPUs.  Mirrors the miba_* entry
    points one to one."""

    def __init__(self, device_id=-1, rank=0, world_size=1, unique_id=None, log_level=0, devices=None):
        """devices: list of HIP device ordinals for ONE handle that drives several GPUs from this process
        (miba_create_multi; a repeated ordinal puts several ranks on one GPU); otherwise one handle = one GPU = one rank"""
        self._h = C.c_void_p()
        self.n_cam = self.n_pt = self.n_obs = 0
        if devices is not None:
            ids = (C.c_int32 * len(devices))(*devices)
            rc = lib().miba_create_multi(ids, C.c_int32(len(devices)), C.c_int32(log_level), C.byref(self._h))
            if rc != 0:
                raise MibaError("miba_create_multi rc=%d: %s" % (rc, (lib().miba_last_error(None) or b"").decode()))
            return
        self._uid = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        cfg = Config(device_id, rank, world_size, log_level,
                     C.cast(self._uid, C.c_void_p) if self._uid is not None else None)
        rc = lib().miba_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise MibaError("miba_create rc=%d: %s" % (rc, (lib().miba_last_error(None) or b"").decode()))
        self.n_cam = self.n_pt = self.n_obs = 0

    def _chk(self, rc, what):
        if rc != 0:
            raise MibaError("%s rc=%d: %s" % (what, rc, (lib().miba_last_error(self._h) or b"").decode()))

    def close(self):
        if self._h:
            lib().miba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_problem(self, cams, pts, obs_cam, obs_pt, obs_xy, K=None, camera_model=0):
        """K: optional (n_cam, 6) intrinsics [fx fy ox oy k1 k2] handed over as they are instead of the BAL-shaped ones
        built from cams; camera_model: MODEL_BAL or MODEL_REFERENCE (the reference's live functor: K carries fx fy ox oy)"""
        R, T, Kb = split_cams(cams)
        K = Kb if K is None else np.ascontiguousarray(K, dtype=np.float64).reshape(-1, 6)
        X = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
        obs_pt = np.ascontiguousarray(obs_pt, dtype=np.int32)
        obs_xy = np.ascontiguousarray(obs_xy, dtype=np.float64)
        self.n_cam, self.n_pt, self.n_obs = R.shape[0], X.shape[0], obs_cam.size
        p = Problem(self.n_cam, self.n_pt, self.n_obs, _dp(R), _dp(T), _dp(K), _dp(X),
                    _ip(obs_cam), _ip(obs_pt), _dp(obs_xy), camera_model, 0)
        self._chk(lib().miba_set_problem(self._h, C.byref(p)), "miba_set_problem")

    def set_state(self, cams, pts, K=None):
        R, T, Kb = split_cams(cams)
        K = Kb if K is None else np.ascontiguousarray(K, dtype=np.float64).reshape(-1, 6)
        X = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        self._chk(lib().miba_set_state(self._h, _dp(R), _dp(T), _dp(K), _dp(X)), "miba_set_state")

    def set_constant_blocks(self, cam_const):
        """cam_const: (n_cam, 3) flags for the rotation / translation / intrinsics block of every camera
        (non-zero = constant), or None to free everything."""
        if cam_const is None:
            self._chk(lib().miba_set_constant_blocks(self._h, None), "miba_set_constant_blocks")
            return
        m = np.ascontiguousarray(np.asarray(cam_const).astype(np.uint8)).reshape(self.n_cam, 3)
        self._chk(lib().miba_set_constant_blocks(self._h, m.ctypes.data_as(C.c_void_p)), "miba_set_constant_blocks")

    def get_state(self, cams_in=None, pts_in=None):
        """Returns (cams (n,9), pts (m,3)); blocks not in the problem keep the given input values."""
        if cams_in is not None:
            R, T, K = split_cams(cams_in)
        else:
            R, T, K = np.zeros((self.n_cam, 3)), np.zeros((self.n_cam, 3)), np.zeros((self.n_cam, 6))
        X = np.array(pts_in, dtype=np.float64).reshape(-1, 3) if pts_in is not None else np.zeros((self.n_pt, 3))
        self._chk(lib().miba_get_state(self._h, _dp(R), _dp(T), _dp(K), _dp(X)), "miba_get_state")
        return join_cams(R, T, K), X

    def evaluate(self, jac=True, grad=True):
        cost = C.c_double()
        r = np.zeros((self.n_obs, 2))
        jc = np.zeros((self.n_obs, 2, 9)) if jac else None
        jp = np.zeros((self.n_obs, 2, 3)) if jac else None
        gc = np.zeros((self.n_cam, 9)) if grad else None
        gp = np.zeros((self.n_pt, 3)) if grad else None
        self._chk(lib().miba_evaluate(self._h, C.byref(cost), _dp(r), _dp(jc), _dp(jp), _dp(gc), _dp(gp)), "miba_evaluate")
        return dict(cost=cost.value, r=r, Jc=jc, Jp=jp, grad_cam=gc, grad_pt=gp)

    def cost(self):
        cost = C.c_double()
        self._chk(lib().miba_evaluate(self._h, C.byref(cost), None, None, None, None, None), "miba_evaluate")
        return cost.value

    def linear_step(self, radius, options=None, probe=None):
        o = options if options is not None else default_options()
        yc, yp, rhs = np.zeros((self.n_cam, 9)), np.zeros((self.n_pt, 3)), np.zeros((self.n_cam, 9))
        v = None if probe is None else np.ascontiguousarray(probe, dtype=np.float64)
        Sv = None if probe is None else np.zeros((self.n_cam, 9))
        ncg, mcc = C.c_int32(), C.c_double()
        self._chk(lib().miba_linear_step(self._h, C.byref(o), C.c_double(radius), _dp(yc), _dp(yp), _dp(rhs),
                                         _dp(v), _dp(Sv), C.byref(ncg), C.byref(mcc)), "miba_linear_step")
        return dict(y_cam=yc, y_pt=yp, rhs=rhs, Sv=Sv, n_cg=ncg.value, model_cost_change=mcc.value)

    def solve(self, options=None, callback=None):
        o = options if options is not None else default_options()
        s = Summary()
        cb = CALLBACK(lambda it, user: int(callback(it.contents) or 0)) if callback else C.cast(None, CALLBACK)
        self._chk(lib().miba_solve(self._h, C.byref(o), cb, None, C.byref(s)), "miba_solve")
        n = C.c_int32()
        lib().miba_get_iterations(self._h, None, 0, C.byref(n))
        its = (Iteration * max(1, n.value))()
        lib().miba_get_iterations(self._h, its, n.value, C.byref(n))
        recs = [{f: getattr(its[i], f) for f, _ in Iteration._fields_} for i in range(n.value)]
        summ = {f: getattr(s, f) for f, _ in Summary._fields_}
        summ["message"] = summ["message"].decode(errors="replace")
        return recs, summ

    def time_matvec(self, probe, reps=20):
        v = np.ascontiguousarray(probe, dtype=np.float64)
        ms = C.c_double()
        self._chk(lib().miba_time_matvec(self._h, _dp(v), C.c_int32(reps), C.byref(ms)), "miba_time_matvec")
        return ms.value

    def reduced_system(self):
        """Explicit Schur complement S (9 n_cam square) of the state prepared by the last linear_step."""
        n = 9 * self.n_cam
        S = np.empty((n, n), dtype=np.float64)
        self._chk(lib().miba_get_reduced_system(self._h, _dp(S)), "miba_get_reduced_system")
        return S

    def dense_spd_solve(self, A, b):
        """x = A^-1 b through the device Cholesky; returns (x, status) with status 2 = not positive definite."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        n = A.shape[0]
        x = np.empty(n, dtype=np.float64)
        st = C.c_int32()
        self._chk(lib().miba_dense_spd_solve(self._h, C.c_int32(n), _dp(A), _dp(b), _dp(x), C.byref(st)), "miba_dense_spd_solve")
        return x, st.value

    def dense_spd_solve_static(self, n, d_A, d_b, d_x, d_status):
        """miba_dense_spd_solve_static: the four arguments are DEVICE addresses (ints) on this handle's GPU; the library
        allocates nothing (static scratch).  n <= STATIC_SOLVE_MAX_N."""
        self._chk(lib().miba_dense_spd_solve_static(self._h, C.c_int32(n), C.c_void_p(d_A), C.c_void_p(d_b), C.c_void_p(d_x),
                                                    C.c_void_p(d_status)), "miba_dense_spd_solve_static")

    def kernel_times(self):
        kt = KernelTimes()
        self._chk(lib().miba_get_kernel_times(self._h, C.byref(kt)), "miba_get_kernel_times")
        return dict(launches=list(kt.launches), total_ms=list(kt.total_ms), n_slots=kt.n_slots,
                    n_obs_local=kt.n_obs_local, n_pt_local=kt.n_pt_local, n_tiles=kt.n_tiles, n_cam=kt.n_cam,
                    product_layout=kt.product_layout, product_clock_mhz=kt.product_clock_mhz,
                    pcg_launches=kt.pcg_launches, pcg_exchange=kt.pcg_exchange)

    def reset_kernel_times(self):
        self._chk(lib().miba_reset_kernel_times(self._h), "miba_reset_kernel_times")
