"""ctypes binding of the CPU oracle (oracle/libba_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product (bundle_adjustment_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libba_oracle.so")

ITERATIVE_SCHUR, DENSE_SCHUR, DENSE_NORMAL = 0, 1, 2
JACOBI, SCHUR_JACOBI = 0, 1
LOSS_NONE, LOSS_HUBER, LOSS_CAUCHY = 0, 1, 2
MODEL_BAL, MODEL_REFERENCE = 0, 1   # options.camera_model (MODEL_REFERENCE: src/BAConstraint.cpp:15-80, bao_project2)


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int),
        ("linear_solver", C.c_int),
        ("jacobi_scaling", C.c_int),
        ("max_linear_solver_iterations", C.c_int),
        ("min_linear_solver_iterations", C.c_int),
        ("max_num_consecutive_invalid_steps", C.c_int),
        ("num_threads", C.c_int),
        ("preconditioner", C.c_int),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("eta", C.c_double),
        ("cam_const", C.c_void_p),
        ("loss_type", C.c_int),
        ("loss_scale", C.c_double),
        ("camera_model", C.c_int),
        ("cam_k4", C.c_void_p),
    ]


class Iteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int),
        ("step_is_valid", C.c_int),
        ("step_is_successful", C.c_int),
        ("linear_solver_iterations", C.c_int),
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("gradient_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
        ("eta", C.c_double),
        ("iteration_time_s", C.c_double),
        ("model_cost_change", C.c_double),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("termination", C.c_int),
        ("num_iterations", C.c_int),
        ("num_successful", C.c_int),
        ("num_unsuccessful", C.c_int),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("total_time_s", C.c_double),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.bao_evaluate.restype = C.c_double
    return _lib


def _p(a, t=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def default_options(**kw):
    o = Options()
    lib().bao_default_options(C.byref(o))
    cc = kw.pop("cam_const", None)
    k4 = kw.pop("cam_k4", None)
    if k4 is not None:
        # the reference's live camera model (bao_project2): fx fy ox oy per camera, intrinsics constant
        o._cam_k4 = np.ascontiguousarray(k4, dtype=np.float64).reshape(-1, 4)
        o.cam_k4 = o._cam_k4.ctypes.data
        o.camera_model = MODEL_REFERENCE
        if cc is None:
            cc = np.zeros((o._cam_k4.shape[0], 3), np.uint8)
        cc = np.array(cc).astype(np.uint8).reshape(-1, 3)
        cc[:, 2] = 1
    if cc is not None:
        # [n_cam][3] flags (rotation, translation, intrinsics); the array is kept alive on the options object
        o._cam_const = np.ascontiguousarray(np.asarray(cc).astype(np.uint8))
        o.cam_const = o._cam_const.ctypes.data
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def project(cam9, X, obs2, jac=True):
    cam9, X, obs2 = _f64(cam9), _f64(X), _f64(obs2)
    r = np.zeros(2)
    jc = np.zeros((2, 9)) if jac else None
    jp = np.zeros((2, 3)) if jac else None
    lib().bao_project(_p(cam9), _p(X), _p(obs2), _p(r), _p(jc), _p(jp))
    return r, jc, jp


def project2(cam6, k4, X, obs2, jac=True):
    """the reference's live functor (src/BAConstraint.cpp:15-80): cam6 = rot, tr; k4 = fx fy ox oy"""
    cam6, k4, X, obs2 = _f64(cam6), _f64(k4), _f64(X), _f64(obs2)
    r = np.zeros(2)
    jc = np.zeros((2, 9)) if jac else None
    jp = np.zeros((2, 3)) if jac else None
    lib().bao_project2(_p(cam6), _p(k4), _p(X), _p(obs2), _p(r), _p(jc), _p(jp))
    return r, jc, jp


def evaluate(cams, pts, obs_cam, obs_pt, obs_xy, jac=True, grad=True, cam_k4=None):
    if cam_k4 is not None:
        return _evaluate2(cams, cam_k4, pts, obs_cam, obs_pt, obs_xy, jac, grad)
    cams, pts, obs_xy = _f64(cams), _f64(pts), _f64(obs_xy)
    obs_cam, obs_pt = _i32(obs_cam), _i32(obs_pt)
    n_cam, n_pt, n_obs = cams.size // 9, pts.size // 3, obs_cam.size
    r = np.zeros((n_obs, 2))
    jc = np.zeros((n_obs, 2, 9)) if jac else None
    jp = np.zeros((n_obs, 2, 3)) if jac else None
    gc = np.zeros((n_cam, 9)) if grad else None
    gp = np.zeros((n_pt, 3)) if grad else None
    cost = lib().bao_evaluate(C.c_int(n_cam), C.c_int(n_pt), C.c_int64(n_obs), _p(cams), _p(pts),
                              _p(obs_cam, C.c_int32), _p(obs_pt, C.c_int32), _p(obs_xy),
                              _p(r), _p(jc), _p(jp), _p(gc), _p(gp))
    return dict(cost=cost, r=r, Jc=jc, Jp=jp, grad_cam=gc, grad_pt=gp)


def _evaluate2(cams, k4, pts, obs_cam, obs_pt, obs_xy, jac, grad):
    cams, k4, pts, obs_xy = _f64(cams), _f64(k4), _f64(pts), _f64(obs_xy)
    obs_cam, obs_pt = _i32(obs_cam), _i32(obs_pt)
    n_cam, n_pt, n_obs = cams.size // 9, pts.size // 3, obs_cam.size
    r = np.zeros((n_obs, 2))
    jc = np.zeros((n_obs, 2, 9)) if jac else None
    jp = np.zeros((n_obs, 2, 3)) if jac else None
    gc = np.zeros((n_cam, 9)) if grad else None
    gp = np.zeros((n_pt, 3)) if grad else None
    lib().bao_evaluate2.restype = C.c_double
    cost = lib().bao_evaluate2(C.c_int(n_cam), C.c_int(n_pt), C.c_int64(n_obs), _p(cams), _p(k4), _p(pts),
                               _p(obs_cam, C.c_int32), _p(obs_pt, C.c_int32), _p(obs_xy),
                               _p(r), _p(jc), _p(jp), _p(gc), _p(gp))
    return dict(cost=cost, r=r, Jc=jc, Jp=jp, grad_cam=gc, grad_pt=gp)


def solve(cams, pts, obs_cam, obs_pt, obs_xy, options=None, max_records=4096):
    """Returns (cams_out, pts_out, iterations(list of dict), summary(dict))."""
    o = options if options is not None else default_options()
    cams, pts, obs_xy = _f64(cams).copy(), _f64(pts).copy(), _f64(obs_xy)
    obs_cam, obs_pt = _i32(obs_cam), _i32(obs_pt)
    n_cam, n_pt, n_obs = cams.size // 9, pts.size // 3, obs_cam.size
    its = (Iteration * max_records)()
    s = Summary()
    rc = lib().bao_solve(C.c_int(n_cam), C.c_int(n_pt), C.c_int64(n_obs), _p(cams), _p(pts),
                         _p(obs_cam, C.c_int32), _p(obs_pt, C.c_int32), _p(obs_xy),
                         C.byref(o), its, C.c_int(max_records), C.byref(s))
    if rc != 0:
        raise RuntimeError("bao_solve failed rc=%d" % rc)
    recs = [{f: getattr(its[i], f) for f, _ in Iteration._fields_}
            for i in range(min(s.num_iterations, max_records))]
    summ = {f: getattr(s, f) for f, _ in Summary._fields_}
    return cams.reshape(n_cam, 9), pts.reshape(n_pt, 3), recs, summ


def jacobi_scale(cams, pts, obs_cam, obs_pt, obs_xy):
    cams, pts, obs_xy = _f64(cams), _f64(pts), _f64(obs_xy)
    obs_cam, obs_pt = _i32(obs_cam), _i32(obs_pt)
    n_cam, n_pt, n_obs = cams.size // 9, pts.size // 3, obs_cam.size
    sc, sp = np.zeros((n_cam, 9)), np.zeros((n_pt, 3))
    lib().bao_jacobi_scale(C.c_int(n_cam), C.c_int(n_pt), C.c_int64(n_obs), _p(cams), _p(pts),
                           _p(obs_cam, C.c_int32), _p(obs_pt, C.c_int32), _p(obs_xy), _p(sc), _p(sp))
    return sc, sp


def linear_step(cams, pts, obs_cam, obs_pt, obs_xy, radius, scale_cam=None, scale_pt=None,
                options=None, probe=None):
    """One damped linear solve; returns dict(y_cam, y_pt, rhs, Sv, n_cg, model_cost_change, rc)."""
    o = options if options is not None else default_options()
    cams, pts, obs_xy = _f64(cams), _f64(pts), _f64(obs_xy)
    obs_cam, obs_pt = _i32(obs_cam), _i32(obs_pt)
    n_cam, n_pt, n_obs = cams.size // 9, pts.size // 3, obs_cam.size
    sc = None if scale_cam is None else _f64(scale_cam)
    sp = None if scale_pt is None else _f64(scale_pt)
    yc, yp, rhs = np.zeros((n_cam, 9)), np.zeros((n_pt, 3)), np.zeros((n_cam, 9))
    v = None if probe is None else _f64(probe)
    Sv = None if probe is None else np.zeros((n_cam, 9))
    ncg = C.c_int(0)
    mcc = C.c_double(0)
    rc = lib().bao_linear_step(C.c_int(n_cam), C.c_int(n_pt), C.c_int64(n_obs), _p(cams), _p(pts),
                               _p(obs_cam, C.c_int32), _p(obs_pt, C.c_int32), _p(obs_xy),
                               C.byref(o), C.c_double(radius), _p(sc), _p(sp), _p(yc), _p(yp),
                               _p(rhs), _p(v), _p(Sv), C.byref(ncg), C.byref(mcc))
    return dict(y_cam=yc, y_pt=yp, rhs=rhs, Sv=Sv, n_cg=ncg.value,
                model_cost_change=mcc.value, rc=rc)


def bal_initialize(cameras_f, points_f, sigma_cam=0.01, sigma_x=2.0, seed=1):
    cameras_f = np.ascontiguousarray(cameras_f, dtype=np.float32)
    points_f = np.ascontiguousarray(points_f, dtype=np.float32)
    n_cam, n_pt = cameras_f.size // 9, points_f.size // 3
    R, T = np.zeros((n_cam, 3)), np.zeros((n_cam, 3))
    K, X = np.zeros((n_cam, 6)), np.zeros((n_pt, 3))
    lib().bao_bal_initialize(C.c_int(n_cam), C.c_int(n_pt), _p(cameras_f, C.c_float),
                             _p(points_f, C.c_float), C.c_double(sigma_cam), C.c_double(sigma_x),
                             C.c_long(seed), _p(R), _p(T), _p(K), _p(X))
    return R, T, K, X
