"""TEST INFRASTRUCTURE -- CPU oracle (numpy + scipy.spatial.cKDTree) for the pose-label point-cloud path.

PARITY UNPINNED: the reference delegates all of this arithmetic to open3d==0.9.0.0 (README.md:44), which is neither
vendored in the reference tree nor installed here, and the reference holds no test or fixture for it.  The functions
below restate open3d 0.9's published algorithms as called from pc_reconstruction/open3d_utils.py (cited per function);
they are the spec the HIP path is checked against, together with recover-a-known-transform self-consistency tests.
Only tests/ may import this module."""
import math

import numpy as np
from scipy.spatial import cKDTree


def surface_points(label, depth, intr, robot2cam):
    """open3d_utils.py:171-192, the per-pixel loop, vectorised in the same float64 op order."""
    ys, xs = np.where(label != 0)
    d = depth[ys, xs].astype(np.float64)
    keep = d != 0
    ys, xs, d = ys[keep], xs[keep], d[keep]
    p0 = (xs - intr["ppx"]) * d / intr["fx"]
    p1 = (ys - intr["ppy"]) * d / intr["fy"]
    T = np.asarray(robot2cam, np.float64)
    return np.stack([((T[r, 0] * p0 + T[r, 1] * p1) + T[r, 2] * d) + T[r, 3] for r in range(3)], 1)


def voxel_down_sample(pts, voxel):
    """PointCloud::VoxelDownSample: voxel index floor((p - (min_bound - voxel/2)) / voxel), mean per voxel; voxels
    returned in lexicographic (x, y, z) index order (open3d's unordered_map order is unspecified)."""
    origin = pts.min(0) - voxel * 0.5
    idx = np.floor((pts - origin) / voxel).astype(np.int64)
    key = (idx[:, 0] << 42) | (idx[:, 1] << 21) | idx[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    out = np.empty((len(starts), 3))
    ends = np.r_[starts[1:], len(ks)]
    for i, (a, b) in enumerate(zip(starts, ends)):
        s = np.zeros(3)
        for j in order[a:b]:          # same summation order as the device (stable sort => ascending original index)
            s += pts[j]
        out[i] = s / (b - a)
    return out


def radius_outlier_mask(pts, nb_points, radius):
    """RemoveRadiusOutliers: keep if #neighbours with d^2 < radius^2 (FLANN radius search, self included) > nb_points"""
    tree = cKDTree(pts)
    cnt = np.array([(np.sum((pts[nb] - pts[i]) ** 2, 1) < radius ** 2).sum()
                    for i, nb in enumerate(tree.query_ball_point(pts, radius * 1.0000001))])
    return cnt > nb_points


def mahalanobis(pts):
    mu = pts.mean(0)
    cov = (pts - mu).T @ (pts - mu) / len(pts)
    ci = np.linalg.inv(cov)
    e = pts - mu
    return np.sqrt(np.einsum("ij,jk,ik->i", e, ci, e))


def statistical_outlier_mask(pts, nb_neighbors, std_ratio):
    """RemoveStatisticalOutliers: mean distance to the nb_neighbors nearest (self included)"""
    tree = cKDTree(pts)
    d, _ = tree.query(pts, k=min(nb_neighbors, len(pts)))
    d = d.reshape(len(pts), -1)
    mean = d.sum(1) / d.shape[1]
    cm = mean.sum() / len(pts)
    sd = math.sqrt(((mean - cm) ** 2).sum() / (len(pts) - 1))
    return (mean > 0) & (mean < cm + std_ratio * sd), mean


def estimate_normals(pts, radius, max_nn):
    """EstimateNormals(KDTreeSearchParamHybrid): covariance of the <= max_nn nearest neighbours with d < radius, smallest
    eigenvector, oriented towards +z; (0,0,1) with fewer than 3 neighbours"""
    tree = cKDTree(pts)
    out = np.zeros_like(pts)
    for i, p in enumerate(pts):
        nb = tree.query_ball_point(p, radius * 1.0000001)
        nb = np.array(nb)
        d2 = np.sum((pts[nb] - p) ** 2, 1)
        nb, d2 = nb[d2 < radius ** 2], d2[d2 < radius ** 2]
        if len(nb) > max_nn:
            nb = nb[np.argsort(d2, kind="stable")[:max_nn]]
        if len(nb) < 3:
            out[i] = (0, 0, 1)
            continue
        q = pts[nb]
        c = np.cov(q.T, bias=True)
        w, v = np.linalg.eigh(c)
        n = v[:, 0]
        out[i] = n if n[2] >= 0 else -n
    return out


def vec6_to_mat4(x):
    cx, sx, cy, sy, cz, sz = math.cos(x[0]), math.sin(x[0]), math.cos(x[1]), math.sin(x[1]), math.cos(x[2]), math.sin(x[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = x[3:]
    return T


def registration_icp(source, target, max_dist, init=None, point_to_plane=False, target_normals=None,
                     relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
    """registration::RegistrationICP as documented for open3d 0.9 (open3d_utils.py:98-117).
    Returns (T, fitness, inlier_rmse)."""
    T = np.eye(4) if init is None else np.array(init, float)
    tree = cKDTree(target)
    src = source @ T[:3, :3].T + T[:3, 3]

    def evaluate(src):
        d, j = tree.query(src, k=1)
        ok = d ** 2 < max_dist ** 2
        n = int(ok.sum())
        return ok, j, n / len(src), (math.sqrt((d[ok] ** 2).sum() / n) if n else 0.0)

    ok, j, fit, rmse = evaluate(src)
    for _ in range(max_iteration):
        s, t = src[ok], target[j[ok]]
        if len(s) < (6 if point_to_plane else 3):
            break
        if point_to_plane:
            nrm = target_normals[j[ok]]
            r = np.einsum("ij,ij->i", s - t, nrm)
            J = np.concatenate([np.cross(s, nrm), nrm], 1)
            x = np.linalg.solve(J.T @ J, -(J.T @ r))
            U = vec6_to_mat4(x)
        else:
            mu_s, mu_t = s.mean(0), t.mean(0)
            cov = (t - mu_t).T @ (s - mu_s) / len(s)
            Uu, _, Vt = np.linalg.svd(cov)
            S = np.eye(3)
            if np.linalg.det(Uu) * np.linalg.det(Vt) < 0:
                S[2, 2] = -1
            R = Uu @ S @ Vt
            U = np.eye(4)
            U[:3, :3] = R
            U[:3, 3] = mu_t - R @ mu_s
        T = U @ T
        src = src @ U[:3, :3].T + U[:3, 3]
        pf, pr = fit, rmse
        ok, j, fit, rmse = evaluate(src)
        if abs(pf - fit) < relative_fitness and abs(pr - rmse) < relative_rmse:
            break
    return T, fit, rmse
