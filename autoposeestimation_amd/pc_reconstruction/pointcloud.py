"""Device-resident stand-in for the slice of open3d 0.9 the pose-label path uses ("pc_reconstruction.
open3d_utils": duck-typed PointCloud surface + registration_icp), backed by the float64 kernels of
csrc/pointcloud.hip.  open3d is a third-party dependency absent from the reference tree, so its exact arithmetic is
UNPINNED; semantics follow open3d 0.9's documented behaviour and are checked against a
scipy/numpy restatement (oracle/pointcloud_oracle.py) plus recover-a-known-transform self-consistency tests.

Points live on the GPU as a contiguous float64 [n,3] tensor; `np.array(pcd.points)` / `np.asarray(pcd.points)` copies to
the host like open3d's Vector3dVector does; assigning `pcd.points = array` uploads.  No CPU fallback."""
import ctypes
import threading
import math
import os

import numpy as np
import torch

from autoposeestimation_amd import _lib

_D = torch.float64


def _st():
    return _lib.stream_ptr()


def _ws(n, device):
    nbytes = _lib.lib().ape_pc_workspace_bytes(int(max(n, 1)))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _host16(T):
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(16))
    return T, T.ctypes.data_as(ctypes.c_void_p)


class _Points:
    """What `pcd.points` returns: converts to numpy on demand, keeps the device tensor for kernels."""

    def __init__(self, t):
        self.t = t

    def __array__(self, dtype=None, copy=None):
        a = self.t.cpu().numpy()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.t.shape[0]


_EMPTY_POINTS = {}


def _empty_points(device):
    """the shared [0, 3] tensor of a device (an empty cloud has nothing to write to; the batched label path makes ~2 k clouds per step)"""
    t = _EMPTY_POINTS.get(device)
    if t is None:
        t = _EMPTY_POINTS[device] = torch.zeros(0, 3, dtype=_D, device=device)
    return t


class PointCloud:
    def __init__(self, points=None, device="cuda"):
        self.device = device if isinstance(device, torch.device) else torch.device(device)
        self._p = _empty_points(self.device)
        self._n = None   # normals
        self._epoch = 0  # bumped by every in-place change of the coordinates (transform): invalidates the cached search grid
        self._gcache = None
        if points is not None:
            self.points = points

    # -- open3d attribute surface ----------------------------------------------------------------------------------
    @property
    def points(self):
        return _Points(self._p)

    @points.setter
    def points(self, value):
        if isinstance(value, _Points):
            value = value.t
        if torch.is_tensor(value):
            t = value.to(device=self.device, dtype=_D)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(value, dtype=np.float64))).to(self.device)
        self._p = t.reshape(-1, 3).contiguous()
        self._n = None

    @property
    def normals(self):
        return None if self._n is None else _Points(self._n)

    def has_normals(self):
        return self._n is not None

    def __len__(self):
        return self._p.shape[0]

    def clone(self):
        c = PointCloud(device=self.device)
        c._p = self._p.clone()
        c._n = None if self._n is None else self._n.clone()
        return c

    __deepcopy__ = lambda self, memo: self.clone()   # noqa: E731  (copy.deepcopy(target), open3d_utils.py:72)

    # -- moments ---------------------------------------------------------------------------------------------------
    def _moments(self):
        n = len(self)
        out = torch.empty(9, dtype=_D, device=self.device)
        ws = torch.empty(512 * 29 * 8, dtype=torch.uint8, device=self.device)
        rc = _lib.lib().ape_icp_sums_f64(2, _lib.dptr(self._p, _D), None, None, None, None, n, _lib.dptr(out), _lib.dptr(ws),
                                         ws.numel(), _st())
        _lib.check(rc, "ape_icp_sums_f64")
        m = out.cpu().numpy()
        mean = m[:3] / n
        s2 = np.array([[m[3], m[4], m[5]], [m[4], m[6], m[7]], [m[5], m[7], m[8]]]) / n
        return mean, s2 - np.outer(mean, mean)      # open3d ComputeMeanAndCovariance: population covariance

    def get_center(self):
        if len(self) == 0:
            return np.zeros(3)
        return self._moments()[0]

    def compute_mahalanobis_distance(self):
        n = len(self)
        if n == 0:
            return np.zeros(0)
        mean, cov = self._moments()
        mc = np.ascontiguousarray(np.concatenate([mean, np.linalg.inv(cov).reshape(9)]))
        out = torch.empty(n, dtype=_D, device=self.device)
        rc = _lib.lib().ape_mahalanobis_f64(_lib.dptr(self._p, _D), n, mc.ctypes.data_as(ctypes.c_void_p), _lib.dptr(out), _st())
        _lib.check(rc, "ape_mahalanobis_f64")
        return out.cpu().numpy()

    # -- rigid motions (in place, return self like open3d) ------------------------------------------------------------
    def transform(self, T):
        if len(self):
            self._epoch += 1
            T, ptr = _host16(T)
            rc = _lib.lib().ape_transform_points_f64(_lib.dptr(self._p, _D), _lib.dptr(self._n), len(self), ptr, _st())
            _lib.check(rc, "ape_transform_points_f64")
        return self

    def translate(self, translation, relative=True):
        T = np.eye(4)
        t = np.asarray(translation, dtype=np.float64)
        T[:3, 3] = t if relative else t - self.get_center()
        return self.transform(T)

    def rotate(self, R, center=True):
        T = np.eye(4)
        T[:3, :3] = np.asarray(R, dtype=np.float64)
        if center:
            c = self.get_center()
            T[:3, 3] = c - T[:3, :3] @ c
        return self.transform(T)

    # -- filters ---------------------------------------------------------------------------------------------------
    def voxel_down_sample(self, voxel_size):
        n = len(self)
        out = PointCloud(device=self.device)
        if n == 0:
            return out
        buf = torch.empty(n, 3, dtype=_D, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        ws = _ws(n, self.device)
        rc = _lib.lib().ape_voxel_down_sample_f64(_lib.dptr(self._p, _D), n, float(voxel_size), _lib.dptr(buf), _lib.dptr(cnt),
                                                  _lib.dptr(ws), ws.numel(), _st())
        _lib.check(rc, "ape_voxel_down_sample_f64")
        out._p = buf[:int(cnt.item())].contiguous()
        return out

    def _grid(self, cell):
        """search grid of this cloud for cell size `cell`; the last one is kept while the coordinates stay the same tensor, unchanged
        (icp_regression registers two estimators against the same target)"""
        c = self._gcache
        if c is not None and c[0] is self._p and c[1] == float(cell) and c[2] == self._epoch:
            return c[3]
        g = self._build_grid(cell)
        self._gcache = (self._p, float(cell), self._epoch, g)
        return g

    def _build_grid(self, cell):
        n = len(self)
        g = {"sorted": torch.empty(n, 3, dtype=_D, device=self.device),
             "keys": torch.empty(n, dtype=torch.int64, device=self.device),
             "order": torch.empty(n, dtype=torch.int32, device=self.device),
             "origin": torch.empty(3, dtype=_D, device=self.device), "n": n, "cell": float(cell)}
        ws = _ws(n, self.device)
        rc = _lib.lib().ape_grid_build_f64(_lib.dptr(self._p, _D), n, float(cell), _lib.dptr(g["sorted"]), _lib.dptr(g["keys"]),
                                           _lib.dptr(g["order"]), _lib.dptr(g["origin"]), _lib.dptr(ws), ws.numel(), _st())
        _lib.check(rc, "ape_grid_build_f64")
        return g

    def _safe_cell(self, cell):
        """a k-NN cell size that keeps every cell coordinate inside the 21-bit key range (the shell bound needs unclamped cells)"""
        cell = float(cell)
        if len(self) * cell > 0 and cell < 1e-3:           # only tiny cells can overflow 2^21 cells per axis: check the extent then
            ext = float((self._p.max(0).values - self._p.min(0).values).max().item())
            cell = max(cell, ext / 1048576.0)
        return cell

    @staticmethod
    def _gargs(g):
        return (_lib.dptr(g["sorted"]), _lib.dptr(g["keys"]), _lib.dptr(g["order"]), _lib.dptr(g["origin"]), g["n"], g["cell"])

    def _select(self, keep):
        n = len(self)
        out = PointCloud(device=self.device)
        buf = torch.empty(n, 3, dtype=_D, device=self.device)
        sel = torch.empty(n, dtype=torch.int32, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        ws = _ws(n, self.device)
        rc = _lib.lib().ape_select_points_f64(_lib.dptr(self._p, _D), _lib.dptr(keep, torch.uint8), n, _lib.dptr(buf), _lib.dptr(sel),
                                              _lib.dptr(cnt), _lib.dptr(ws), ws.numel(), _st())
        _lib.check(rc, "ape_select_points_f64")
        k = int(cnt.item())
        out._p = buf[:k].contiguous()
        return out, sel[:k].cpu().numpy().tolist()

    def remove_radius_outlier(self, nb_points, radius):
        """keeps points with MORE than nb_points neighbours (self included) at distance < radius; -> (cloud, kept indices)"""
        n = len(self)
        if n == 0:
            return PointCloud(device=self.device), []
        g = self._grid(radius)
        count = torch.empty(n, dtype=torch.int32, device=self.device)
        rc = _lib.lib().ape_grid_radius_count_f64(*self._gargs(g), _lib.dptr(self._p, _D), n, float(radius), _lib.dptr(count), _st())
        _lib.check(rc, "ape_grid_radius_count_f64")
        return self._select((count > int(nb_points)).to(torch.uint8))

    def remove_statistical_outlier(self, nb_neighbors, std_ratio, cell_hint=None):
        """open3d 0.9 RemoveStatisticalOutliers: mean distance to the nb_neighbors nearest (self included) must be
        < cloud mean + std_ratio * sample std; -> (cloud, kept indices).  The k-NN search walks a uniform grid (exact for any cell
        size, ape_grid_knn_mean_dist_f64); `cell_hint` = a radius expected to hold the k neighbours (get_surface passes the radius of
        the radius-outlier filter that ran just before), otherwise it is derived from the cloud's extent and point count."""
        n = len(self)
        if n == 0:
            return PointCloud(device=self.device), []
        k = int(min(nb_neighbors, n))
        mean = torch.empty(n, dtype=_D, device=self.device)
        if cell_hint is None:
            ext = (self._p.max(0).values - self._p.min(0).values).cpu().numpy()
            area = ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2]       # a surface scan: ~n / area points per unit area
            cell_hint = math.sqrt(max(k * area / (3.0 * n), 1e-300))
            if not (cell_hint > 0 and math.isfinite(cell_hint)):
                cell_hint = 1.0
        g = self._grid(self._safe_cell(cell_hint))
        rc = _lib.lib().ape_grid_knn_mean_dist_f64(*self._gargs(g), k, _lib.dptr(mean), _st())
        _lib.check(rc, "ape_grid_knn_mean_dist_f64")
        m = mean.cpu().numpy()
        valid = m >= 0
        cloud_mean = m[valid].sum() / max(int(valid.sum()), 1)
        std = math.sqrt(((m[valid] - cloud_mean) ** 2).sum() / max(int(valid.sum()) - 1, 1))
        thr = cloud_mean + float(std_ratio) * std
        keep = torch.from_numpy(((m > 0) & (m < thr)).astype(np.uint8)).to(self.device)
        return self._select(keep)

    def estimate_normals(self, search_param=None, radius=None, max_nn=30):
        """KDTreeSearchParamHybrid(radius, max_nn) semantics (open3d_utils.py:25-27)"""
        if search_param is not None:
            radius, max_nn = search_param.radius, search_param.max_nn
        n = len(self)
        if n == 0:
            return self
        g = self._grid(radius)
        self._n = torch.empty(n, 3, dtype=_D, device=self.device)
        rc = _lib.lib().ape_grid_normals_f64(*self._gargs(g), _lib.dptr(self._p, _D), n, float(radius), int(max_nn), _lib.dptr(self._n), _st())
        _lib.check(rc, "ape_grid_normals_f64")
        return self


class KDTreeSearchParamHybrid:
    def __init__(self, radius, max_nn):
        self.radius, self.max_nn = radius, max_nn


class ICPConvergenceCriteria:
    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
        self.relative_fitness, self.relative_rmse, self.max_iteration = relative_fitness, relative_rmse, max_iteration


class TransformationEstimationPointToPoint:
    kind = 0

    def __init__(self, with_scaling=False):
        if with_scaling:
            raise NotImplementedError("with_scaling")


class TransformationEstimationPointToPlane:
    kind = 1


class RegistrationResult:
    def __init__(self, transformation, fitness, inlier_rmse, n_corr):
        self.transformation, self.fitness, self.inlier_rmse, self.correspondence_count = transformation, fitness, inlier_rmse, n_corr

    def __repr__(self):
        return "RegistrationResult(fitness=%.6f, inlier_rmse=%.6f, correspondences=%d)" % (self.fitness, self.inlier_rmse,
                                                                                           self.correspondence_count)


def _umeyama(s):
    """Eigen::umeyama(src, dst, with_scaling=false) from the one-pass sums s[17]."""
    n = s[0]
    mu_s, mu_t = s[2:5] / n, s[5:8] / n
    cov = s[8:17].reshape(3, 3).T / n - np.outer(mu_t, mu_s)        # (1/n) sum (t - mu_t)(s - mu_s)^T
    U, d, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = mu_t - R @ mu_s
    return T


def _vec6_to_mat4(x):
    """open3d TransformVector6dToMatrix4d: R = Rz(x[2]) Ry(x[1]) Rx(x[0]), t = x[3:6]"""
    cx, sx, cy, sy, cz, sz = math.cos(x[0]), math.sin(x[0]), math.cos(x[1]), math.sin(x[1]), math.cos(x[2]), math.sin(x[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = x[3:6]
    return T


def _point_to_plane(s):
    JTJ = np.zeros((6, 6))
    k = 2
    for a in range(6):
        for b in range(a, 6):
            JTJ[a, b] = JTJ[b, a] = s[k]
            k += 1
    JTr = s[23:29]
    try:
        x = np.linalg.solve(JTJ, -JTr)
    except np.linalg.LinAlgError:
        return np.eye(4)
    return _vec6_to_mat4(x)


ICP_STATS = None          # bench.py --workload label sets a dict here: registrations, evaluations, point pairs, (start, end) events
_ICP_STATS_LOCK = threading.Lock()     # registrations of different chains run on different host threads (sharding.run_side_by_side)
_ICP_CHUNK = 5            # iterations enqueued per device round trip (3 launches each); the reference's criteria (1e-2) stop after 2-4


def registration_icp(source, target, max_correspondence_distance, init=None, estimation_method=None, criteria=None, host_solve=False):
    """open3d 0.9 registration::RegistrationICP (call sites open3d_utils.py:56-58,98-117).  The source cloud is NOT
    modified (open3d works on a transformed copy).  The whole iteration -- correspondence search, the 17/29 reduced sums, the
    3x3 SVD / 6x6 solve, T <- update . T and the convergence test -- runs on the device (ape_icp_run_f64): a chunk of iterations
    is enqueued at once, launches after convergence are no-ops, and the 40-double state comes back once per chunk (normally once
    per registration).  `host_solve=True` keeps the round-1 loop (numpy SVD / solve, one D2H per iteration) for the parity tests."""
    estimation_method = estimation_method or TransformationEstimationPointToPoint()
    criteria = criteria or ICPConvergenceCriteria()
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64)
    ns, nt = len(source), len(target)
    if ns == 0 or nt == 0:
        return RegistrationResult(T, 0.0, 0.0, 0)
    if estimation_method.kind == 1 and not target.has_normals():
        raise RuntimeError("TransformationEstimationPointToPlane requires target normals")
    dev = source.device
    pcd = source.clone().transform(T)
    grid = target._grid(max_correspondence_distance)
    corr = torch.empty(ns, dtype=torch.int32, device=dev)
    d2 = torch.empty(ns, dtype=_D, device=dev)
    sums = torch.empty(29, dtype=_D, device=dev)
    ws = torch.empty(512 * 29 * 8, dtype=torch.uint8, device=dev)
    L = _lib.lib()
    if not host_solve:
        st0 = np.zeros(40)
        st0[5:21] = T.reshape(-1)
        state = torch.from_numpy(st0).to(dev)
        left, first, chunk = int(criteria.max_iteration), 1, _ICP_CHUNK         # the first call's block already holds one step
        if ICP_STATS is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        while True:
            n_it = min(chunk, left)
            _lib.check(L.ape_icp_run_f64(estimation_method.kind, *PointCloud._gargs(grid), _lib.dptr(pcd._p, _D), ns, _lib.dptr(target._p, _D),
                                         _lib.dptr(target._n), float(max_correspondence_distance), float(criteria.relative_fitness),
                                         float(criteria.relative_rmse), int(criteria.max_iteration), n_it, first, _lib.dptr(corr), _lib.dptr(d2),
                                         _lib.dptr(sums), _lib.dptr(state, _D), _lib.dptr(ws), ws.numel(), _st()), "ape_icp_run_f64")
            out = state.cpu().numpy()
            left -= n_it
            first = 0
            if out[0] != 0.0 or left <= 0:
                break
            chunk *= 2
        if ICP_STATS is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            with _ICP_STATS_LOCK:
                ICP_STATS["registrations"] += 1
                ICP_STATS["evaluations"] += int(out[1]) + 1
                ICP_STATS["pairs"] += (int(out[1]) + 1) * ns
                ICP_STATS["kind%d" % estimation_method.kind] = ICP_STATS.get("kind%d" % estimation_method.kind, 0) + (int(out[1]) + 1) * ns
                ICP_STATS["events"].append((ev0, ev1))
        return RegistrationResult(out[5:21].reshape(4, 4).copy(), float(out[2]), float(out[3]), int(out[4]))

    def evaluate():
        _lib.check(L.ape_grid_nn1_f64(*PointCloud._gargs(grid), _lib.dptr(pcd._p, _D), ns, float(max_correspondence_distance),
                                      _lib.dptr(corr), _lib.dptr(d2), _st()), "ape_grid_nn1_f64")
        _lib.check(L.ape_icp_sums_f64(estimation_method.kind, _lib.dptr(pcd._p, _D), _lib.dptr(target._p, _D), _lib.dptr(target._n),
                                      _lib.dptr(corr), _lib.dptr(d2), ns, _lib.dptr(sums), _lib.dptr(ws), ws.numel(), _st()),
                   "ape_icp_sums_f64")
        s = sums.cpu().numpy()
        n_corr = int(round(s[0]))
        fitness = n_corr / ns
        rmse = math.sqrt(s[1] / n_corr) if n_corr else 0.0
        return s, n_corr, fitness, rmse

    s, n_corr, fitness, rmse = evaluate()
    for _ in range(criteria.max_iteration):
        if n_corr < (3 if estimation_method.kind == 0 else 6):
            break
        update = _umeyama(s) if estimation_method.kind == 0 else _point_to_plane(s)
        T = update @ T
        pcd.transform(update)
        prev_f, prev_r = fitness, rmse
        s, n_corr, fitness, rmse = evaluate()
        if abs(prev_f - fitness) < criteria.relative_fitness and abs(prev_r - rmse) < criteria.relative_rmse:
            break
    return RegistrationResult(T, fitness, rmse, n_corr)


def surface_points(label, depth, intr, robot2cam, device="cuda"):
    """label u8[H,W], depth (integer sensor units) [H,W] -> PointCloud of the valid pixels in the robot frame (mm)."""
    dev = torch.device(device)
    if torch.is_tensor(label) and torch.is_tensor(depth) and label.is_cuda and depth.is_cuda:
        # views already resident in HBM
        if label.dtype != torch.uint8 or depth.dtype != torch.uint16:
            raise ValueError("resident views must be uint8 labels and uint16 depth")
        lab, dep, dev = label.contiguous(), depth.contiguous(), label.device
    else:
        lab = torch.as_tensor(np.ascontiguousarray(np.asarray(label, dtype=np.uint8))).to(dev)
        d = np.asarray(depth)
        if d.dtype != np.uint16:
            if (d < 0).any() or (d > 65535).any() or (d != np.floor(d)).any():
                raise ValueError("depth must hold integer sensor units in 0..65535")
            d = d.astype(np.uint16)
        dep = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    h, w = lab.shape
    buf = torch.empty(h * w, 3, dtype=_D, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _ws(h * w, dev)
    T, ptr = _host16(robot2cam)
    rc = _lib.lib().ape_surface_points_f64(_lib.dptr(lab, torch.uint8), _lib.dptr(dep, torch.uint16), h, w, float(intr.get("fx")),
                                           float(intr.get("fy")), float(intr.get("ppx")), float(intr.get("ppy")), ptr, _lib.dptr(buf),
                                           _lib.dptr(cnt), _lib.dptr(ws), ws.numel(), _st())
    _lib.check(rc, "ape_surface_points_f64")
    out = PointCloud(device=dev)
    out._p = buf[:int(cnt.item())].contiguous()
    return out


# ---- o3d.io stand-ins: .ply / .pcd point clouds, xyz only -----------------------------------------------------------------------
# The reference writes its clouds with o3d.io.write_point_cloud's defaults (create_pointcloud.py:324-344: write_ascii=False,
# compressed=False), i.e. `format binary_little_endian 1.0` PLY with double x/y/z (+ optional normals / colours) and `DATA binary`
# PCD; both, their ASCII forms and extra per-vertex properties are read here, and the writer produces open3d's default layout.
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
_PCD_TYPES = {("F", 4): "f4", ("F", 8): "f8", ("I", 1): "i1", ("I", 2): "i2", ("I", 4): "i4", ("I", 8): "i8",
              ("U", 1): "u1", ("U", 2): "u2", ("U", 4): "u4", ("U", 8): "u8"}


def write_point_cloud(path, pcd, write_ascii=False, compressed=False):
    if compressed:
        raise NotImplementedError("compressed point-cloud files are not written")
    pts = np.ascontiguousarray(np.array(pcd.points, dtype=np.float64).reshape(-1, 3))
    ext = os.path.splitext(path)[1].lower()
    if ext == ".ply":
        head = ("ply\nformat %s 1.0\ncomment Created by autoposeestimation_amd\nelement vertex %d\nproperty double x\nproperty double y\n"
                "property double z\nend_header\n" % ("ascii" if write_ascii else "binary_little_endian", len(pts)))
    elif ext == ".pcd":
        head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\n"
                "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (len(pts), len(pts), "ascii" if write_ascii else "binary"))
    else:
        raise ValueError("unsupported point-cloud format %r" % ext)
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        if write_ascii:
            f.write("".join("%.17g %.17g %.17g\n" % (p[0], p[1], p[2]) for p in pts).encode("ascii"))
        else:
            f.write(pts.astype("<f8").tobytes())
    return True


def _read_ply(raw):
    end = raw.index(b"end_header")
    end = raw.index(b"\n", end) + 1
    head = raw[:end].decode("ascii", "replace").split("\n")
    fmt = [ln.split()[1] for ln in head if ln.startswith("format")][0]
    props, n, in_vertex, before = [], 0, False, 0
    for ln in head:
        t = ln.split()
        if not t:
            continue
        if t[0] == "element":
            if in_vertex:
                in_vertex = False
            elif t[1] == "vertex":
                in_vertex, n = True, int(t[2])
            elif not props:
                before += 1
        elif t[0] == "property" and in_vertex:
            if t[1] == "list":
                raise ValueError("list properties on PLY vertices are not supported")
            props.append((t[-1], _PLY_TYPES[t[1]]))
    if before:
        raise ValueError("PLY files with elements in front of `vertex` are not supported")
    names = [p[0] for p in props]
    if not all(k in names for k in "xyz"):
        raise ValueError("PLY vertex element lacks x / y / z")
    if fmt == "ascii":
        rows = raw[end:].decode("ascii", "replace").split("\n")[:n]
        tab = np.array([[float(v) for v in r.split()[:len(props)]] for r in rows], dtype=np.float64).reshape(n, len(props))
        return np.stack([tab[:, names.index(k)] for k in "xyz"], 1)
    order = "<" if fmt == "binary_little_endian" else ">"
    dt = np.dtype([(nm, order + ty) for nm, ty in props])
    rec = np.frombuffer(raw, dtype=dt, count=n, offset=end)
    return np.stack([rec[k].astype(np.float64) for k in "xyz"], 1)


def _read_pcd(raw):
    pos, hdr = 0, {}
    while True:
        nl = raw.index(b"\n", pos)
        line = raw[pos:nl].decode("ascii", "replace").strip()
        pos = nl + 1
        if line.startswith("#") or not line:
            continue
        key, _, val = line.partition(" ")
        hdr[key] = val.split()
        if key == "DATA":
            break
    fields = hdr["FIELDS"]
    sizes = [int(v) for v in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(v) for v in hdr.get("COUNT", ["1"] * len(fields))]
    n = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
    if not all(k in fields for k in "xyz"):
        raise ValueError("PCD file lacks x / y / z fields")
    mode = hdr["DATA"][0]
    if mode == "ascii":
        cols = np.cumsum([0] + counts)
        rows = raw[pos:].decode("ascii", "replace").split("\n")[:n]
        tab = np.array([[float(v) for v in r.split()] for r in rows], dtype=np.float64).reshape(n, -1)
        return np.stack([tab[:, cols[fields.index(k)]] for k in "xyz"], 1)
    if mode != "binary":
        raise NotImplementedError("PCD DATA %s is not supported (open3d writes `binary` unless compressed=True)" % mode)
    dt = np.dtype([(f, "<" + _PCD_TYPES[(t, s)], (c,)) if c > 1 else (f, "<" + _PCD_TYPES[(t, s)]) for f, t, s, c in zip(fields, types, sizes, counts)])
    rec = np.frombuffer(raw, dtype=dt, count=n, offset=pos)
    return np.stack([rec[k].astype(np.float64).reshape(n) for k in "xyz"], 1)


def read_point_cloud(path, device="cuda"):
    ext = os.path.splitext(path)[1].lower()
    with open(path, "rb") as f:
        raw = f.read()
    if ext == ".ply":
        pts = _read_ply(raw)
    elif ext == ".pcd":
        pts = _read_pcd(raw)
    else:
        raise ValueError("unsupported point-cloud format %r" % ext)
    return PointCloud(np.ascontiguousarray(pts, dtype=np.float64), device=device)
