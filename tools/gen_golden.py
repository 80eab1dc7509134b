#!/usr/bin/env python3
"""Generate tests/golden/proj_golden.json: 50-digit mpmath residuals and Jacobians of the
9-parameter BAL camera for random and edge-case (camera, point, obs)
triples.  Jacobians come from high-precision central differences of the *model expression*
(h = 1e-25 at 80 digits), i.e. they do not share any code or algebra with oracle/ba_oracle.c
or the HIP kernels.

Branch rule restated from ceres/rotation.h (AngleAxisRotatePoint; Ceres is not in the
reference tree): theta^2 > DBL_EPSILON uses Rodrigues, otherwise RX = X + w x X, and the
derivative is the derivative of the branch expression actually taken.

Run here (container) only:  python tools/gen_golden.py
"""
import json
import os
import random
import sys

import mpmath as mp

mp.mp.dps = 80
DBL_EPS = mp.mpf(2) ** -52


def rotate(w, X, small):
    if small:
        return [X[0] + (w[1] * X[2] - w[2] * X[1]),
                X[1] + (w[2] * X[0] - w[0] * X[2]),
                X[2] + (w[0] * X[1] - w[1] * X[0])]
    th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2]
    th = mp.sqrt(th2)
    c, s = mp.cos(th), mp.sin(th)
    k = [wi / th for wi in w]
    kxX = [k[1] * X[2] - k[2] * X[1], k[2] * X[0] - k[0] * X[2], k[0] * X[1] - k[1] * X[0]]
    kX = k[0] * X[0] + k[1] * X[1] + k[2] * X[2]
    return [X[i] * c + kxX[i] * s + k[i] * kX * (1 - c) for i in range(3)]


def residual(params, obs, small):
    """params = 9 camera scalars + 3 point scalars (mp.mpf)."""
    w, t, f, k1, k2, X = params[0:3], params[3:6], params[6], params[7], params[8], params[9:12]
    RX = rotate(w, X, small)
    P = [RX[i] + t[i] for i in range(3)]
    px, py = -P[0] / P[2], -P[1] / P[2]
    n = px * px + py * py
    d = 1 + k1 * n + k2 * n * n
    return [f * d * px - obs[0], f * d * py - obs[1]]


def case(cam, X, obs):
    """cam, X, obs are python floats (exact doubles); returns dict with hex inputs and
    decimal-string outputs."""
    params = [mp.mpf(v) for v in cam] + [mp.mpf(v) for v in X]
    o = [mp.mpf(v) for v in obs]
    th2 = params[0] ** 2 + params[1] ** 2 + params[2] ** 2
    # the branch test is evaluated in double precision by the implementation
    th2_d = float(cam[0]) * float(cam[0]) + float(cam[1]) * float(cam[1]) + float(cam[2]) * float(cam[2])
    small = not (th2_d > float(DBL_EPS))
    r = residual(params, o, small)
    h = mp.mpf(10) ** -25
    J = [[None] * 12 for _ in range(2)]
    for j in range(12):
        pp = list(params); pm = list(params)
        pp[j] += h; pm[j] -= h
        rp, rm = residual(pp, o, small), residual(pm, o, small)
        for i in range(2):
            J[i][j] = (rp[i] - rm[i]) / (2 * h)
    fmt = lambda v: mp.nstr(v, 30)
    return {
        "cam_hex": [float(v).hex() for v in cam],
        "X_hex": [float(v).hex() for v in X],
        "obs_hex": [float(v).hex() for v in obs],
        "small_angle": bool(small),
        "r": [fmt(v) for v in r],
        "Jc": [[fmt(J[i][j]) for j in range(9)] for i in range(2)],
        "Jp": [[fmt(J[i][9 + j]) for j in range(3)] for i in range(2)],
    }


def main():
    rng = random.Random(20201113)
    cases = []

    def rand_case(theta=None, zmin=-30.0, zmax=-1.0, k1=None, k2=None):
        axis = [rng.gauss(0, 1) for _ in range(3)]
        nrm = sum(a * a for a in axis) ** 0.5
        th = rng.uniform(0.01, 3.0) if theta is None else theta
        w = [a / nrm * th for a in axis]
        # choose the camera-frame point first (BAL cameras look down -z), then back out t
        Pc = [rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(zmin, zmax)]
        X = [rng.uniform(-10, 10) for _ in range(3)]
        small = not (w[0] * w[0] + w[1] * w[1] + w[2] * w[2] > float(DBL_EPS))
        RX = [float(v) for v in rotate([mp.mpf(v) for v in w], [mp.mpf(v) for v in X], small)]
        t = [Pc[i] - RX[i] for i in range(3)]
        f = rng.uniform(400, 2000)
        kk1 = rng.uniform(-0.3, 0.05) if k1 is None else k1
        kk2 = rng.uniform(-0.02, 0.1) if k2 is None else k2
        cam = w + t + [f, kk1, kk2]
        r0 = residual([mp.mpf(v) for v in cam + X], [mp.mpf(0), mp.mpf(0)], small)
        # observations are float32-rounded values widened to double
        import struct
        obs = [struct.unpack("f", struct.pack("f", float(r0[i]) + rng.gauss(0, 0.7)))[0] for i in range(2)]
        return case(cam, X, obs)

    for _ in range(256):
        cases.append(rand_case())
    # edge cases: rotation magnitude around the branch switch and around pi
    for th in [0.0, 1e-12, 1e-9, 1.4e-8, 1.49e-8, 1.5e-8, 1.6e-8, 3e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3,
               3.0, 3.14159, 3.141592653589793, 3.2, 4.5, 6.2, 6.283185307179586, 9.0]:
        for _ in range(3):
            cases.append(rand_case(theta=th))
    # strong distortion
    for _ in range(12):
        cases.append(rand_case(k1=rng.uniform(-2, 2), k2=rng.uniform(-1, 1)))
    # depth close to zero (large |p|)
    for _ in range(12):
        cases.append(rand_case(zmin=-0.2, zmax=-0.02))
    # far away
    for _ in range(6):
        cases.append(rand_case(zmin=-5000.0, zmax=-500.0))

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "proj_golden.json")
    with open(out, "w") as fh:
        json.dump({"generator": "tools/gen_golden.py", "dps": mp.mp.dps, "n": len(cases), "cases": cases}, fh)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    sys.exit(main())
