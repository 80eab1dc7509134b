"""Independent numpy restatement of the reference's solve path for TINY problems.

TEST INFRASTRUCTURE ONLY (see oracle/ba_oracle.h).  It shares no code with ba_oracle.c:
  * residuals: vectorised numpy BAL model;
  * Jacobian: complex-step differentiation of that model (h = 1e-30), no analytic formulas;
  * linear solve: the FULL damped normal equations (J'J + D'D) y = J'r with numpy's LAPACK;
  * LM loop: SURVEY.md App. B.1/B.2 (Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy,
    reached in the reference through ceres::Solve, /root/reference/src/BundleAdjustment.cpp:108
    with the options of /root/reference/app/main.cpp:21-28).
It pins the C oracle's LM trajectory (tests/golden/lm_tiny_*.json, tools/gen_lm_golden.py).
"""
import numpy as np

DBL_EPS = np.finfo(np.float64).eps


def residuals(cams, pts, obs_cam, obs_pt, obs_xy):
    """cams (n,9) [w t f k1 k2], pts (m,3); complex inputs allowed. Returns (o,2)."""
    c = cams[obs_cam]
    X = pts[obs_pt]
    w, t, f, k1, k2 = c[:, 0:3], c[:, 3:6], c[:, 6], c[:, 7], c[:, 8]
    th2 = (w * w).sum(1)
    small = ~(th2.real > DBL_EPS)
    th2s = np.where(small, 1.0, th2)
    th = np.sqrt(th2s)
    k = w / th[:, None]
    co, si = np.cos(th), np.sin(th)
    kxX = np.cross(k, X)
    kX = (k * X).sum(1)
    RX_big = X * co[:, None] + kxX * si[:, None] + k * (kX * (1 - co))[:, None]
    RX_small = X + np.cross(w, X)
    RX = np.where(small[:, None], RX_small, RX_big)
    P = RX + t
    px, py = -P[:, 0] / P[:, 2], -P[:, 1] / P[:, 2]
    n = px * px + py * py
    d = 1 + k1 * n + k2 * n * n
    return np.stack([f * d * px - obs_xy[:, 0], f * d * py - obs_xy[:, 1]], 1)


def jacobian(cams, pts, obs_cam, obs_pt, obs_xy):
    """Dense J (2o x (3m + 9n)), columns: points first then cameras (Ceres' Schur ordering)."""
    n_cam, n_pt, n_obs = cams.shape[0], pts.shape[0], obs_cam.shape[0]
    J = np.zeros((2 * n_obs, 3 * n_pt + 9 * n_cam))
    h = 1e-30
    rows = np.arange(n_obs)
    for k in range(9):
        cc = cams.astype(complex)
        cc[:, k] += 1j * h
        d = residuals(cc, pts.astype(complex), obs_cam, obs_pt, obs_xy).imag / h
        J[2 * rows, 3 * n_pt + 9 * obs_cam + k] = d[:, 0]
        J[2 * rows + 1, 3 * n_pt + 9 * obs_cam + k] = d[:, 1]
    for k in range(3):
        pc = pts.astype(complex)
        pc[:, k] += 1j * h
        d = residuals(cams.astype(complex), pc, obs_cam, obs_pt, obs_xy).imag / h
        J[2 * rows, 3 * obs_pt + k] = d[:, 0]
        J[2 * rows + 1, 3 * obs_pt + k] = d[:, 1]
    return J


def default_options():
    return dict(max_num_iterations=50, jacobi_scaling=True, initial_trust_region_radius=1e4,
                max_trust_region_radius=1e16, min_trust_region_radius=1e-32,
                min_relative_decrease=1e-3, min_lm_diagonal=1e-6, max_lm_diagonal=1e32,
                function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8,
                max_num_consecutive_invalid_steps=5, cam_const=None, loss=None)


def solve(cams, pts, obs_cam, obs_pt, obs_xy, **kw):
    o = default_options()
    o.update(kw)
    cams, pts = np.array(cams, float).reshape(-1, 9), np.array(pts, float).reshape(-1, 3)
    obs_cam, obs_pt = np.asarray(obs_cam), np.asarray(obs_pt)
    obs_xy = np.asarray(obs_xy, float).reshape(-1, 2)
    n_cam, n_pt = cams.shape[0], pts.shape[0]
    np3 = 3 * n_pt
    # every block must be referenced (tiny fixtures are built that way)
    assert set(obs_cam.tolist()) == set(range(n_cam)) and set(obs_pt.tolist()) == set(range(n_pt))

    def pack(c, p):
        return np.concatenate([p.ravel(), c.ravel()])

    def unpack(x):
        return x[np3:].reshape(n_cam, 9), x[:np3].reshape(n_pt, 3)

    # robust loss (ceres::LossFunction on every residual block): loss = ("huber" | "cauchy", a).  Both have
    # rho'' <= 0, for which Ceres' Corrector multiplies the block's residual and Jacobian rows by sqrt(rho'(s)),
    # s = |r|^2, and the cost is 1/2 sum rho(s).
    loss = o.pop("loss", None)
    weights = [np.ones(2 * obs_cam.shape[0])]   # row weights of the last evaluate()

    def rho(s):
        if loss is None:
            return s, np.ones_like(s)
        kind, a = loss
        b = a * a
        if kind == "huber":
            rt = np.sqrt(np.maximum(s, b))
            return np.where(s > b, 2 * a * rt - b, s), np.where(s > b, a / rt, 1.0)
        if kind == "cauchy":
            return b * np.log1p(s / b), 1.0 / (1.0 + s / b)
        raise ValueError(kind)

    def evaluate(x):
        c, p = unpack(x)
        r = residuals(c, p, obs_cam, obs_pt, obs_xy)
        val, d1 = rho((r * r).sum(1))
        weights[0] = np.repeat(np.sqrt(d1), 2)
        return 0.5 * val.sum(), r.ravel() * weights[0]

    # constant parameter blocks (ceres::Problem::SetParameterBlockConstant, src/BundleAdjustment.cpp:77-79,88-90):
    # cam_const[c] = (rotation, translation, intrinsics) flags; constant columns are removed from the program
    free = np.ones(np3 + 9 * n_cam, bool)
    cc = o.pop("cam_const", None)
    if cc is not None:
        cc = np.asarray(cc, bool).reshape(n_cam, 3)
        free[np3:] = ~np.repeat(cc, 3, axis=1).ravel()
    jac_full = jacobian

    def jacobian_free(c, p, oc, op, oxy):   # always called right after evaluate() of the same state
        return jac_full(c, p, oc, op, oxy)[:, free] * weights[0][:, None]

    x = pack(cams, pts)
    radius, decrease_factor, reuse_diag = o["initial_trust_region_radius"], 2.0, False
    x_cost, r = evaluate(x)
    J = jacobian_free(*unpack(x), obs_cam, obs_pt, obs_xy)
    g = J.T @ r
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0))) if o["jacobi_scaling"] else np.ones(J.shape[1])
    J = J * scale
    x_norm = np.linalg.norm(x[free])
    best_x, min_cost = x.copy(), x_cost
    it = dict(iteration=0, cost=x_cost, cost_change=0.0, gradient_max_norm=np.abs(g).max(),
              gradient_norm=np.linalg.norm(g), step_norm=0.0, relative_decrease=0.0,
              trust_region_radius=radius, step_is_valid=False, step_is_successful=False)
    its = []
    invalid = 0
    diag = None
    termination = "NO_CONVERGENCE"
    while True:
        if it["step_is_successful"] and x_cost < min_cost:
            min_cost, best_x = x_cost, x.copy()
        it["trust_region_radius"] = radius
        its.append(dict(it))
        if it["iteration"] >= o["max_num_iterations"]:
            break
        if it["gradient_max_norm"] <= o["gradient_tolerance"]:
            termination = "CONVERGENCE"; break
        if radius < o["min_trust_region_radius"]:
            termination = "CONVERGENCE"; break
        prev = it
        it = dict(iteration=prev["iteration"] + 1, cost=x_cost, cost_change=0.0,
                  gradient_max_norm=prev["gradient_max_norm"], gradient_norm=prev["gradient_norm"],
                  step_norm=0.0, relative_decrease=0.0, trust_region_radius=radius,
                  step_is_valid=False, step_is_successful=False)
        if not reuse_diag:
            diag = np.clip((J * J).sum(0), o["min_lm_diagonal"], o["max_lm_diagonal"])
        D2 = diag / radius
        reuse_diag = True
        ok = True
        try:
            L = np.linalg.cholesky(J.T @ J + np.diag(D2))
            y = np.linalg.solve(L.T, np.linalg.solve(L, J.T @ r))
        except np.linalg.LinAlgError:
            ok = False
        mcc = 0.0
        if ok and np.all(np.isfinite(y)):
            step = -y
            Jd = J @ step
            mcc = -Jd @ (r + Jd / 2.0)
            it["step_is_valid"] = bool(mcc > 0)
        it["model_cost_change"] = mcc
        if not it["step_is_valid"]:
            invalid += 1
            if invalid >= o["max_num_consecutive_invalid_steps"]:
                termination = "FAILURE"; break
            radius *= 0.5
            continue
        invalid = 0
        delta = np.zeros_like(x)
        delta[free] = step * scale
        x_cand = x + delta
        cand_cost, _ = evaluate(x_cand)
        if not np.isfinite(cand_cost):
            cand_cost = np.finfo(float).max
        it["step_norm"] = np.linalg.norm(x - x_cand)
        it["cost_change"] = x_cost - cand_cost
        it["relative_decrease"] = it["cost_change"] / mcc
        if it["step_norm"] <= o["parameter_tolerance"] * (x_norm + o["parameter_tolerance"]):
            termination = "CONVERGENCE"; break
        if abs(it["cost_change"]) <= o["function_tolerance"] * x_cost:
            termination = "CONVERGENCE"; break
        if it["relative_decrease"] > o["min_relative_decrease"]:
            x = x_cand
            x_norm = np.linalg.norm(x[free])
            x_cost, r = evaluate(x)
            J = jacobian_free(*unpack(x), obs_cam, obs_pt, obs_xy)
            g = J.T @ r
            J = J * scale
            it.update(cost=x_cost, gradient_max_norm=np.abs(g).max(), gradient_norm=np.linalg.norm(g),
                      step_is_successful=True)
            radius = radius / max(1.0 / 3.0, 1.0 - (2.0 * it["relative_decrease"] - 1.0) ** 3)
            radius = min(o["max_trust_region_radius"], radius)
            decrease_factor, reuse_diag = 2.0, False
        else:
            it["cost"] = cand_cost
            radius /= decrease_factor
            decrease_factor *= 2.0
            reuse_diag = True
    c, p = unpack(best_x)
    return c.copy(), p.copy(), its, dict(termination=termination, final_cost=min_cost,
                                          initial_cost=its[0]["cost"])
