"""Deterministic synthetic BAL-format problems (SURVEY.md §8d "Concrete inputs").

No BAL file exists in the build container or on the GPU box, so every "BAL <name>" config of
BASELINE.json is stood in for by a seeded synthetic problem with the same
(cameras, points, observations) triple.  All file scalars are rounded through float32 exactly
as the reference loader parses them (/root/reference/src/BalDataloader.cpp:32-33,44-45,55-56)
and widened to double afterwards (:131-157).

Scene: points uniform in a 20 x 20 x 4 slab, cameras on a closed loop of radius 15 around it
looking at a jittered target near the slab centre, BAL convention (camera looks down -z,
p = -P.xy / P.z), f ~ U(500,1500), k1 ~ U(-0.1,0), k2 ~ U(0,0.02).  Each point is seen by a
contiguous window of cameras on the far side of the loop; track length 2 + Geometric.
Observation = exact projection + N(0, 0.5^2) px.
"""
import numpy as np

# (n_cam, n_pt, n_obs, seed) for the BASELINE.json configs
CONFIGS = {
    "ladybug-49": (49, 7776, 31843, 49),
    "trafalgar-257": (257, 65132, 225911, 257),
    "dubrovnik-356": (356, 226730, 1255268, 356),
    "final-13682": (13682, 4456117, 28987644, 13682),
    "synthetic-50k": (50000, 20000000, 150000000, 50000),
}


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def _rotmat_to_angle_axis(R):
    """R (n,3,3) -> angle-axis (n,3), robust for angles in [0, pi)."""
    q = np.empty((R.shape[0], 4))
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    q[:, 0] = np.sqrt(np.maximum(0.0, 1.0 + tr)) / 2.0
    q[:, 1] = np.sqrt(np.maximum(0.0, 1.0 + R[:, 0, 0] - R[:, 1, 1] - R[:, 2, 2])) / 2.0
    q[:, 2] = np.sqrt(np.maximum(0.0, 1.0 - R[:, 0, 0] + R[:, 1, 1] - R[:, 2, 2])) / 2.0
    q[:, 3] = np.sqrt(np.maximum(0.0, 1.0 - R[:, 0, 0] - R[:, 1, 1] + R[:, 2, 2])) / 2.0
    q[:, 1] = np.copysign(q[:, 1], R[:, 2, 1] - R[:, 1, 2])
    q[:, 2] = np.copysign(q[:, 2], R[:, 0, 2] - R[:, 2, 0])
    q[:, 3] = np.copysign(q[:, 3], R[:, 1, 0] - R[:, 0, 1])
    s = np.linalg.norm(q[:, 1:], axis=1)
    ang = 2.0 * np.arctan2(s, q[:, 0])
    k = np.where(s[:, None] > 1e-12, q[:, 1:] / np.maximum(s, 1e-300)[:, None], 0.0)
    return k * ang[:, None]


def _rotate(w, X):
    th = np.linalg.norm(w, axis=1)
    k = w / np.maximum(th, 1e-300)[:, None]
    c, s = np.cos(th)[:, None], np.sin(th)[:, None]
    return X * c + np.cross(k, X) * s + k * ((k * X).sum(1)[:, None]) * (1 - c)


def project(cams, pts, obs_cam, obs_pt):
    c = cams[obs_cam]
    P = _rotate(c[:, 0:3], pts[obs_pt]) + c[:, 3:6]
    p = -P[:, 0:2] / P[:, 2:3]
    n = (p * p).sum(1)
    d = 1.0 + c[:, 7] * n + c[:, 8] * n * n
    return (c[:, 6] * d)[:, None] * p, P[:, 2]


def make_problem(n_cam, n_pt, n_obs, seed=0, obs_sigma=0.5, shuffle_obs=False):
    """Returns dict(cams (n_cam,9), pts (n_pt,3), obs_cam, obs_pt (int32), obs_xy (n_obs,2)),
    all float-rounded ground truth as a BAL file would hold it."""
    lmax = max(2, min(n_cam, n_cam // 3 + 2))
    if n_cam < 2 or n_obs < 2 * n_pt or n_obs > n_pt * lmax:
        raise ValueError("need n_cam >= 2 and 2*n_pt <= n_obs <= n_pt*%d (track cap)" % lmax)
    rng = np.random.Generator(np.random.PCG64(seed))
    # cameras on the loop
    phi = 2 * np.pi * (np.arange(n_cam) + rng.uniform(-0.3, 0.3, n_cam)) / n_cam
    C = np.stack([15 * np.cos(phi), 15 * np.sin(phi), rng.uniform(1, 3, n_cam)], 1)
    target = np.stack([rng.uniform(-1, 1, n_cam), rng.uniform(-1, 1, n_cam), rng.uniform(-2, 2, n_cam)], 1)
    d = target - C
    d /= np.linalg.norm(d, axis=1)[:, None]
    zc = -d                                   # camera looks down -z
    up = np.array([0.0, 0.0, 1.0])
    xc = np.cross(up, zc)
    xc /= np.linalg.norm(xc, axis=1)[:, None]
    yc = np.cross(zc, xc)
    R = np.stack([xc, yc, zc], 1)             # rows = camera axes
    w = _rotmat_to_angle_axis(R)
    t = -np.einsum("nij,nj->ni", R, C)
    cams = np.concatenate([w, t, rng.uniform(500, 1500, (n_cam, 1)), rng.uniform(-0.1, 0.0, (n_cam, 1)),
                           rng.uniform(0.0, 0.02, (n_cam, 1))], 1)
    cams = _f32(cams)
    pts = _f32(np.stack([rng.uniform(-10, 10, n_pt), rng.uniform(-10, 10, n_pt), rng.uniform(-2, 2, n_pt)], 1))

    # track lengths: 2 + Geometric(mean = n_obs/n_pt - 2), capped, then fixed up to sum n_obs
    mean_extra = n_obs / n_pt - 2.0
    if mean_extra > 0:
        L = 2 + rng.geometric(1.0 / (1.0 + mean_extra), n_pt) - 1
    else:
        L = np.full(n_pt, 2, dtype=np.int64)
    L = np.minimum(L, lmax).astype(np.int64)
    diff = int(n_obs - L.sum())
    while diff != 0:
        if diff > 0:
            cand = np.flatnonzero(L < lmax)
            if cand.size == 0:
                raise RuntimeError("cannot reach n_obs")
            take = cand[rng.permutation(cand.size)[:min(diff, cand.size)]]
            L[take] += 1
        else:
            cand = np.flatnonzero(L > 2)
            if cand.size == 0:
                raise RuntimeError("cannot reach n_obs")
            take = cand[rng.permutation(cand.size)[:min(-diff, cand.size)]]
            L[take] -= 1
        diff = int(n_obs - L.sum())
    # window of cameras on the far side of the loop
    alpha = np.arctan2(pts[:, 1], pts[:, 0])
    c0 = np.rint((alpha + np.pi + rng.uniform(-0.4, 0.4, n_pt)) / (2 * np.pi) * n_cam).astype(np.int64)
    off = np.zeros(n_pt + 1, dtype=np.int64)
    np.cumsum(L, out=off[1:])
    obs_pt = np.repeat(np.arange(n_pt, dtype=np.int64), L)
    kk = np.arange(n_obs, dtype=np.int64) - off[obs_pt]
    obs_cam = np.mod(c0[obs_pt] - L[obs_pt] // 2 + kk, n_cam)
    obs_xy = np.empty((n_obs, 2))
    chunk = 1 << 22
    for a in range(0, n_obs, chunk):
        b = min(n_obs, a + chunk)
        xy, z = project(cams, pts, obs_cam[a:b], obs_pt[a:b])
        if not np.all(z < 0):
            raise RuntimeError("synthetic scene put a point behind a camera")
        obs_xy[a:b] = xy + rng.normal(0.0, obs_sigma, xy.shape)
    obs_xy = _f32(obs_xy)
    obs_cam, obs_pt = obs_cam.astype(np.int32), obs_pt.astype(np.int32)
    if shuffle_obs:
        perm = rng.permutation(n_obs)
        obs_cam, obs_pt, obs_xy = obs_cam[perm], obs_pt[perm], obs_xy[perm]
    return dict(cams=cams, pts=pts, obs_cam=obs_cam, obs_pt=obs_pt, obs_xy=np.ascontiguousarray(obs_xy),
                n_cam=n_cam, n_pt=n_pt, n_obs=n_obs, seed=seed)


def perturb(problem, sigma_cam=0.01, sigma_x=2.0, seed=1):
    """Initial state = ground truth + Gaussian noise on camera (rot, tr) and points, the
    reference loader's policy (/root/reference/src/BalDataloader.cpp:132-156) with a seeded
    numpy stream instead of glibc rand().  Intrinsics (f, k1, k2) are not perturbed."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cams = problem["cams"].copy()
    pts = problem["pts"].copy()
    cams[:, 0:6] += rng.normal(0.0, 1.0, (cams.shape[0], 6)) * sigma_cam
    pts += rng.normal(0.0, 1.0, pts.shape) * sigma_x
    return cams, pts


def write_bal(path, problem, fmt="%.16e"):
    """BAL text: header, obs lines 'cam pt x y', 9 scalars per camera, 3 per point."""
    with open(path, "w") as fh:
        fh.write("%d %d %d\n" % (problem["n_cam"], problem["n_pt"], problem["n_obs"]))
        xy = problem["obs_xy"]
        for i in range(problem["n_obs"]):
            fh.write(("%d %d " + fmt + " " + fmt + "\n") % (problem["obs_cam"][i], problem["obs_pt"][i], xy[i, 0], xy[i, 1]))
        for v in problem["cams"].ravel():
            fh.write((fmt + "\n") % v)
        for v in problem["pts"].ravel():
            fh.write((fmt + "\n") % v)
