#!/usr/bin/env python3
"""Generate tests/golden/proj2_golden.json: 80-digit mpmath residuals and Jacobians of the REFERENCE's live cost functor
(/root/reference/src/BAConstraint.cpp:15-80; camera model #2 of libmiba) for random and edge-case (camera, point,
observation) triples.  The functor is restated here from its text, expression by expression: AngleAxisToRotationMatrix with
Ceres' two branches (:35), the 4x4 extrinsic and its inverse (:41-46; evaluated as R^-1 (X - t) with an explicit 3x3 inverse,
so the small-angle branch inverts I + [w]x as the reference does), division by z without a sign flip (:67-68), no distortion
(:69-70 computes it and drops it), K * Pimg (:72), observed - projected (:73).  Jacobians are central differences of that
expression at 80 digits (h = 1e-25) with respect to rot, tr and the point: they share no algebra with oracle/ba_oracle.c or
the HIP kernels.  Run here (container) only:  python tools/gen_golden2.py"""
import json, os, random, struct, sys
import mpmath as mp

mp.mp.dps = 80
DBL_EPS = mp.mpf(2) ** -52


def rotmat(w, small):
    if small:
        return mp.matrix([[1, -w[2], w[1]], [w[2], 1, -w[0]], [-w[1], w[0], 1]])
    th = mp.sqrt(w[0] ** 2 + w[1] ** 2 + w[2] ** 2)
    c, s = mp.cos(th), mp.sin(th)
    k = [wi / th for wi in w]
    K = mp.matrix([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    kk = mp.matrix(3, 3)
    for i in range(3):
        for j in range(3):
            kk[i, j] = k[i] * k[j]
    return c * mp.eye(3) + s * K + (1 - c) * kk


def residual(params, k4, obs, small):
    """params = rot[3], tr[3], point[3]"""
    w, t, X = params[0:3], params[3:6], params[6:9]
    R = rotmat(w, small)
    P = mp.inverse(R) * mp.matrix([X[i] - t[i] for i in range(3)])
    px, py = P[0] / P[2], P[1] / P[2]
    return [obs[0] - (k4[0] * px + k4[2]), obs[1] - (k4[1] * py + k4[3])]


def case(cam6, k4, X, obs):
    params = [mp.mpf(v) for v in cam6] + [mp.mpf(v) for v in X]
    kk, o = [mp.mpf(v) for v in k4], [mp.mpf(v) for v in obs]
    th2_d = float(cam6[0]) * float(cam6[0]) + float(cam6[1]) * float(cam6[1]) + float(cam6[2]) * float(cam6[2])
    small = not (th2_d > float(DBL_EPS))
    r = residual(params, kk, o, small)
    h = mp.mpf(10) ** -25
    J = [[None] * 9 for _ in range(2)]
    for j in range(9):
        pp, pm = list(params), list(params)
        pp[j] += h; pm[j] -= h
        rp, rm = residual(pp, kk, o, small), residual(pm, kk, o, small)
        for i in range(2):
            J[i][j] = (rp[i] - rm[i]) / (2 * h)
    fmt = lambda v: mp.nstr(v, 30)
    return {"cam_hex": [float(v).hex() for v in cam6], "k4_hex": [float(v).hex() for v in k4], "X_hex": [float(v).hex() for v in X],
            "obs_hex": [float(v).hex() for v in obs], "small_angle": bool(small), "r": [fmt(v) for v in r],
            "Jc": [[fmt(J[i][j]) for j in range(6)] for i in range(2)], "Jp": [[fmt(J[i][6 + j]) for j in range(3)] for i in range(2)]}


def main():
    rng = random.Random(20261111)
    cases = []

    def rand_case(theta=None, zmin=0.5, zmax=8.0, same_f=True):
        axis = [rng.gauss(0, 1) for _ in range(3)]
        nrm = sum(a * a for a in axis) ** 0.5
        th = rng.uniform(0.01, 3.0) if theta is None else theta
        w = [a / nrm * th for a in axis]
        small = not (w[0] * w[0] + w[1] * w[1] + w[2] * w[2] > float(DBL_EPS))
        # camera-frame point first (the Kinect camera looks down +z), then the world point: X = R P + t
        Pc = [rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(zmin, zmax)]
        t = [rng.uniform(-3, 3) for _ in range(3)]
        R = rotmat([mp.mpf(v) for v in w], small)
        Xm = R * mp.matrix(Pc)
        X = [float(Xm[i]) + t[i] for i in range(3)]
        fx = rng.uniform(300, 900)
        fy = fx if same_f else fx * rng.uniform(0.9, 1.1)
        k4 = [fx, fy, rng.uniform(200, 400), rng.uniform(150, 300)]
        r0 = residual([mp.mpf(v) for v in w + t + X], [mp.mpf(v) for v in k4], [mp.mpf(0), mp.mpf(0)], small)
        obs = [struct.unpack("f", struct.pack("f", -float(r0[i]) + rng.gauss(0, 0.7)))[0] for i in range(2)]
        return case(w + t, k4, X, obs)

    for _ in range(160):
        cases.append(rand_case())
    for _ in range(40):
        cases.append(rand_case(same_f=False))
    for th in [0.0, 1e-12, 1e-9, 1.4e-8, 1.5e-8, 1.6e-8, 1e-7, 1e-5, 1e-3, 3.0, 3.141592653589793, 4.5, 6.283185307179586]:
        for _ in range(3):
            cases.append(rand_case(theta=th))
    for _ in range(10):
        cases.append(rand_case(zmin=0.02, zmax=0.1))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "proj2_golden.json")
    with open(out, "w") as fh:
        json.dump({"generator": "tools/gen_golden2.py", "dps": mp.mp.dps, "n": len(cases), "cases": cases}, fh)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    sys.exit(main())
