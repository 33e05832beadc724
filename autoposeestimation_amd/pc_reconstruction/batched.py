"""Lock-step batched form of the pose-label point-cloud path (the (object, direction) chains are the independent units).

A chain by itself is a string of tiny dependent kernels (10^3..10^4 points, float64): round 2 drove three chains side by side from three
host threads and still spent its time in launch latency (38.6 k launches and 13.6 k small copies per 200-view step, the threads fighting
over the GIL).  Here ONE host thread advances up to 16 clouds per call: every primitive below takes a LIST of clouds and issues ONE launch
(`blockIdx.y` = cloud, csrc/pointcloud.hip "BATCHED forms") with ONE device-to-host copy of the counts it needs -- the same `*_body`
device code with the same per-cloud grid sizes as the one-cloud methods of `PointCloud`, and the same host float64 arithmetic between
the launches, so every cloud comes out BIT-IDENTICAL to the one-cloud path (tests/test_gpu_pointcloud.py compares them).

    get_surface_batch(views, ...)          = [open3d_utils.get_surface(v) for v in views]              (reference open3d_utils.py:171-213)
    fuse_surfaces_batch(chains, ...)       = [open3d_utils.fuse_surfaces(c) for c in chains]           (create_pointcloud.py:288-312)
"""
import ctypes
import math

import numpy as np
import torch

from autoposeestimation_amd import _lib
from autoposeestimation_amd.pc_reconstruction import pointcloud as PC

_D = torch.float64
MAX_BATCH = 16


def _st():
    return _lib.stream_ptr()


def _ptrs(tensors):
    """(c_void_p * n) of the tensors' device addresses (None -> NULL)"""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _ints(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def _dbls(values):
    return (ctypes.c_double * len(values))(*[float(v) for v in values])


def _ws(nb, n_total, device):
    return torch.empty(_lib.lib().ape_pc_batch_workspace_bytes(int(nb), int(max(n_total, 1))), dtype=torch.uint8, device=device)


def _chunks(seq, size=MAX_BATCH):
    for i in range(0, len(seq), size):
        yield list(range(i, min(i + size, len(seq))))


def _cloud(points, device):
    c = PC.PointCloud(device=device)
    c._p = points
    return c


# ---- primitives over lists of clouds ----------------------------------------------------------------------------------------------------
def surface_points(views, intr_default, device="cuda"):
    """[pointcloud.surface_points(label, depth, intr, robot2cam)] for resident views (label u8 / depth u16 CUDA tensors of ONE image size)"""
    out = [None] * len(views)
    for idx in _chunks(views):
        labs, deps, intrs, Ts = [], [], [], []
        for i in idx:
            v = views[i]
            label, depth, cam = v[:3]
            intr = v[3] if len(v) > 3 and v[3] is not None else intr_default
            if not (torch.is_tensor(label) and label.is_cuda and label.dtype == torch.uint8 and torch.is_tensor(depth) and depth.dtype == torch.uint16):
                lab = torch.as_tensor(np.ascontiguousarray(np.asarray(label.cpu() if torch.is_tensor(label) else label, dtype=np.uint8))).to(device)
                d = np.asarray(depth.cpu() if torch.is_tensor(depth) else depth)
                if d.dtype != np.uint16:
                    if (d < 0).any() or (d > 65535).any() or (d != np.floor(d)).any():
                        raise ValueError("depth must hold integer sensor units in 0..65535")
                    d = d.astype(np.uint16)
                label, depth = lab, torch.from_numpy(np.ascontiguousarray(d)).to(device)
            labs.append(label.contiguous())
            deps.append(depth.contiguous())
            intrs += [float(intr.get("fx")), float(intr.get("fy")), float(intr.get("ppx")), float(intr.get("ppy"))]
            Ts.append(np.asarray(cam, dtype=np.float64).reshape(16))
        h, w = labs[0].shape
        if any(tuple(x.shape) != (h, w) for x in labs + deps):
            raise ValueError("the views of one batch must share one image size")
        dev = labs[0].device
        nb = len(idx)
        bufs = [torch.empty(h * w, 3, dtype=_D, device=dev) for _ in idx]
        cnt = torch.zeros(nb, dtype=torch.int32, device=dev)
        pix = torch.empty(nb * h * w, dtype=torch.int32, device=dev)
        rc = _lib.lib().ape_surface_points_batch_f64(nb, _ptrs(labs), _ptrs(deps), h, w, _dbls(intrs), _dbls(np.stack(Ts).reshape(-1)), _ptrs(bufs), _lib.dptr(cnt),
                                                     _lib.dptr(pix), _st())
        _lib.check(rc, "ape_surface_points_batch_f64")
        counts = cnt.cpu().numpy()
        for k, i in enumerate(idx):
            out[i] = _cloud(bufs[k][:int(counts[k])].contiguous(), dev)
    return out


def voxel_down_sample(clouds, voxel_size):
    out = [None] * len(clouds)
    for idx in _chunks(clouds):
        ns = [len(clouds[i]) for i in idx]
        dev = clouds[idx[0]].device
        if sum(ns) == 0:
            for i in idx:
                out[i] = PC.PointCloud(device=dev)
            continue
        nb = len(idx)
        bufs = [torch.empty(max(n, 1), 3, dtype=_D, device=dev) for n in ns]
        cnt = torch.zeros(nb, dtype=torch.int32, device=dev)
        ws = _ws(nb, sum(ns), dev)
        rc = _lib.lib().ape_voxel_down_sample_batch_f64(nb, _ptrs([clouds[i]._p for i in idx]), _ints(ns), float(voxel_size), _ptrs(bufs), _lib.dptr(cnt),
                                                        _lib.dptr(ws), ws.numel(), _st())
        _lib.check(rc, "ape_voxel_down_sample_batch_f64")
        counts = cnt.cpu().numpy()
        for k, i in enumerate(idx):
            out[i] = _cloud(bufs[k][:int(counts[k])].contiguous(), dev) if ns[k] else PC.PointCloud(device=dev)
    return out


def build_grids(clouds, cell):
    """the search grid of every cloud for ONE cell size (PointCloud._grid, cached on the cloud like the one-cloud path does); empty clouds get None"""
    grids = [None] * len(clouds)
    todo = []
    for i, c in enumerate(clouds):
        if len(c) == 0:
            continue
        g = c._gcache
        if g is not None and g[0] is c._p and g[1] == float(cell) and g[2] == c._epoch:
            grids[i] = g[3]
        else:
            todo.append(i)
    for idx in _chunks(todo):
        sel = [todo[j] for j in idx]
        ns = [len(clouds[i]) for i in sel]
        dev = clouds[sel[0]].device
        gs = [{"sorted": torch.empty(n, 3, dtype=_D, device=dev), "keys": torch.empty(n, dtype=torch.int64, device=dev),
               "order": torch.empty(n, dtype=torch.int32, device=dev), "origin": torch.empty(3, dtype=_D, device=dev), "n": n, "cell": float(cell)} for n in ns]
        ws = _ws(len(sel), sum(ns), dev)
        rc = _lib.lib().ape_grid_build_batch_f64(len(sel), _ptrs([clouds[i]._p for i in sel]), _ints(ns), float(cell), _ptrs([g["sorted"] for g in gs]),
                                                 _ptrs([g["keys"] for g in gs]), _ptrs([g["order"] for g in gs]), _ptrs([g["origin"] for g in gs]),
                                                 _lib.dptr(ws), ws.numel(), _st())
        _lib.check(rc, "ape_grid_build_batch_f64")
        for g, i in zip(gs, sel):
            clouds[i]._gcache = (clouds[i]._p, float(cell), clouds[i]._epoch, g)
            grids[i] = g
    return grids


def _grid_args(grids):
    """BGRID_ARGS for a list of grids (None -> an empty grid)"""
    return (_ptrs([None if g is None else g["sorted"] for g in grids]), _ptrs([None if g is None else g["keys"] for g in grids]),
            _ptrs([None if g is None else g["order"] for g in grids]), _ptrs([None if g is None else g["origin"] for g in grids]),
            _ints([0 if g is None else g["n"] for g in grids]))


def _select(clouds, mode, counts=None, thr_count=0, means=None, thr_means=None):
    """ordered row selection with the keep rule on the device -> new clouds"""
    out = [None] * len(clouds)
    for idx in _chunks(clouds):
        ns = [len(clouds[i]) for i in idx]
        dev = clouds[idx[0]].device
        nb = len(idx)
        bufs = [torch.empty(max(n, 1), 3, dtype=_D, device=dev) for n in ns]
        cnt = torch.zeros(nb, dtype=torch.int32, device=dev)
        sel = torch.empty(max(sum(ns), 1), dtype=torch.int32, device=dev)
        rc = _lib.lib().ape_select_points_batch_f64(mode, nb, _ptrs([clouds[i]._p for i in idx]), _ints(ns),
                                                    _ptrs([counts[i] for i in idx]) if mode == 0 else None, int(thr_count),
                                                    _ptrs([means[i] for i in idx]) if mode == 1 else None,
                                                    _dbls([thr_means[i] for i in idx]) if mode == 1 else None, _ptrs(bufs), _lib.dptr(cnt), _lib.dptr(sel), _st())
        _lib.check(rc, "ape_select_points_batch_f64")
        kept = cnt.cpu().numpy()
        for k, i in enumerate(idx):
            out[i] = _cloud(bufs[k][:int(kept[k])].contiguous(), dev) if ns[k] else PC.PointCloud(device=dev)
    return out


def remove_radius_outlier(clouds, nb_points, radius):
    grids = build_grids(clouds, radius)
    counts = [None] * len(clouds)
    for idx in _chunks(clouds):
        dev = clouds[idx[0]].device
        cs = [torch.empty(max(len(clouds[i]), 1), dtype=torch.int32, device=dev) for i in idx]
        rc = _lib.lib().ape_grid_query_batch_f64(0, len(idx), *_grid_args([grids[i] for i in idx]), float(radius), _ptrs([clouds[i]._p for i in idx]),
                                                 _ints([len(clouds[i]) for i in idx]), float(radius), 0, _ptrs(cs), None, None, _st())
        _lib.check(rc, "ape_grid_query_batch_f64")
        for k, i in enumerate(idx):
            counts[i] = cs[k]
    return _select(clouds, 0, counts=counts, thr_count=int(nb_points))


def moments(clouds):
    """[PointCloud._moments()] -> (mean[3], covariance[3,3]) per cloud (None for an empty one)"""
    out = [None] * len(clouds)
    for idx in _chunks(clouds):
        dev = clouds[idx[0]].device
        ns = [len(clouds[i]) for i in idx]
        o9 = torch.zeros(len(idx), 9, dtype=_D, device=dev)
        ws = torch.empty(len(idx) * 512 * 9 * 8, dtype=torch.uint8, device=dev)
        rc = _lib.lib().ape_moments_batch_f64(len(idx), _ptrs([clouds[i]._p for i in idx]), _ints(ns), _lib.dptr(o9), _lib.dptr(ws), ws.numel(), _st())
        _lib.check(rc, "ape_moments_batch_f64")
        m_all = o9.cpu().numpy()
        for k, i in enumerate(idx):
            if ns[k] == 0:
                continue
            m, n = m_all[k], ns[k]
            mean = m[:3] / n
            s2 = np.array([[m[3], m[4], m[5]], [m[4], m[6], m[7]], [m[5], m[7], m[8]]]) / n
            out[i] = (mean, s2 - np.outer(mean, mean))
    return out


def mahalanobis(clouds):
    """[np.array(c.compute_mahalanobis_distance())] (host arrays, like the one-cloud method)"""
    mom = moments(clouds)
    out = [np.zeros(0)] * len(clouds)
    for idx in _chunks(clouds):
        dev = clouds[idx[0]].device
        ns = [len(clouds[i]) for i in idx]
        mc = np.zeros((len(idx), 12))
        for k, i in enumerate(idx):
            if ns[k]:
                mean, cov = mom[i]
                mc[k] = np.concatenate([mean, np.linalg.inv(cov).reshape(9)])
        total = sum(ns)
        if total == 0:
            continue
        flat = torch.empty(total, dtype=_D, device=dev)
        offs = np.concatenate([[0], np.cumsum(ns)])
        rc = _lib.lib().ape_mahalanobis_batch_f64(len(idx), _ptrs([clouds[i]._p for i in idx]), _ints(ns), _dbls(mc.reshape(-1)),
                                                  _ptrs([flat[offs[k]:offs[k + 1]] for k in range(len(idx))]), _st())
        _lib.check(rc, "ape_mahalanobis_batch_f64")
        host = flat.cpu().numpy()                    # ONE device-to-host copy for the whole batch
        for k, i in enumerate(idx):
            out[i] = host[offs[k]:offs[k + 1]].copy()
    return out


def remove_statistical_outlier(clouds, nb_neighbors, std_ratios, cell_hint):
    """PointCloud.remove_statistical_outlier per cloud (its own std_ratio each), one k-NN launch, one copy of the means, the float64 threshold
    statistics on the host exactly as the one-cloud method computes them, the keep rule evaluated on the device"""
    out = [None] * len(clouds)
    live = [i for i, c in enumerate(clouds) if len(c) > 0]
    for i in range(len(clouds)):
        if i not in live:
            out[i] = PC.PointCloud(device=clouds[i].device)
    # the k-NN grid's cell: _safe_cell(cell_hint) per cloud; clouds that need another cell than the common one go through the one-cloud method
    common = float(cell_hint)
    odd = [i for i in live if clouds[i]._safe_cell(common) != common or min(nb_neighbors, len(clouds[i])) != nb_neighbors]
    for i in odd:
        out[i] = clouds[i].remove_statistical_outlier(nb_neighbors, std_ratios[i], cell_hint=cell_hint)[0]
    live = [i for i in live if i not in odd]
    if not live:
        return out
    sub = [clouds[i] for i in live]
    grids = build_grids(sub, common)
    means = [None] * len(sub)
    for idx in _chunks(sub):
        dev = sub[idx[0]].device
        ns = [len(sub[j]) for j in idx]
        offs = np.concatenate([[0], np.cumsum(ns)])
        flat = torch.empty(int(offs[-1]), dtype=_D, device=dev)
        views = [flat[offs[k]:offs[k + 1]] for k in range(len(idx))]
        rc = _lib.lib().ape_grid_query_batch_f64(2, len(idx), *_grid_args([grids[j] for j in idx]), common, None, None, 0.0, int(nb_neighbors), None, None,
                                                 _ptrs(views), _st())
        _lib.check(rc, "ape_grid_query_batch_f64")
        host = flat.cpu().numpy()
        for k, j in enumerate(idx):
            means[j] = (views[k], host[offs[k]:offs[k + 1]])
    thr = []
    for j, (_, m) in enumerate(means):
        valid = m >= 0
        cloud_mean = m[valid].sum() / max(int(valid.sum()), 1)
        std = math.sqrt(((m[valid] - cloud_mean) ** 2).sum() / max(int(valid.sum()) - 1, 1))
        thr.append(cloud_mean + float(std_ratios[live[j]]) * std)
    kept = _select(sub, 1, means=[mv[0] for mv in means], thr_means=thr)
    for j, i in enumerate(live):
        out[i] = kept[j]
    return out


def estimate_normals(clouds, radius, max_nn):
    grids = build_grids(clouds, radius)
    for idx in _chunks(clouds):
        dev = clouds[idx[0]].device
        nrm = [torch.empty(max(len(clouds[i]), 1), 3, dtype=_D, device=dev) for i in idx]
        rc = _lib.lib().ape_grid_query_batch_f64(1, len(idx), *_grid_args([grids[i] for i in idx]), float(radius), _ptrs([clouds[i]._p for i in idx]),
                                                 _ints([len(clouds[i]) for i in idx]), float(radius), int(max_nn), None, _ptrs(nrm), None, _st())
        _lib.check(rc, "ape_grid_query_batch_f64")
        for k, i in enumerate(idx):
            if len(clouds[i]):
                clouds[i]._n = nrm[k][:len(clouds[i])]
    return clouds


def transform(clouds, Ts):
    """in place, like PointCloud.transform"""
    for idx in _chunks(clouds):
        live = [i for i in idx if len(clouds[i])]
        if not live:
            continue
        T_host = np.stack([np.asarray(Ts[i], dtype=np.float64).reshape(16) for i in live]).reshape(-1)
        for i in live:
            clouds[i]._epoch += 1
        rc = _lib.lib().ape_transform_points_batch_f64(len(live), _ptrs([clouds[i]._p for i in live]), _ptrs([clouds[i]._n for i in live]),
                                                       _ints([len(clouds[i]) for i in live]), _dbls(T_host), _st())
        _lib.check(rc, "ape_transform_points_batch_f64")
    return clouds


def concat(a_list, b_list=None):
    """[cat(a, b)] as new clouds (b_list None: copies of a)"""
    out = []
    for idx in _chunks(a_list):
        dev = a_list[idx[0]].device
        na = [len(a_list[i]) for i in idx]
        nb_ = [len(b_list[i]) if b_list is not None else 0 for i in idx]
        bufs = [torch.empty(x + y, 3, dtype=_D, device=dev) for x, y in zip(na, nb_)]
        rc = _lib.lib().ape_concat_points_batch_f64(len(idx), _ptrs([a_list[i]._p for i in idx]), _ints(na),
                                                    _ptrs([b_list[i]._p for i in idx]) if b_list is not None else None,
                                                    _ints(nb_) if b_list is not None else None, _ptrs(bufs), _st())
        _lib.check(rc, "ape_concat_points_batch_f64")
        out += [_cloud(b, dev) for b in bufs]
    return out


def registration_icp(sources, targets, max_correspondence_distance, inits, kind, criteria):
    """[pointcloud.registration_icp(s, t, dist, init, estimator(kind), criteria).transformation] for pairs advancing together: one launch triple
    per iteration for all pairs, one copy of all 40-double states per chunk of iterations; a pair that has converged turns its launches
    into no-ops (its device `done` word), exactly as in the one-pair loop."""
    n = len(sources)
    Ts = [np.eye(4) if inits[i] is None else np.array(inits[i], dtype=np.float64) for i in range(n)]
    live = [i for i in range(n) if len(sources[i]) and len(targets[i])]
    if not live:
        return Ts
    moved = transform(concat([sources[i] for i in live]), [Ts[i] for i in live])      # open3d works on a transformed copy
    tg = [targets[i] for i in live]
    grids = build_grids(tg, max_correspondence_distance)
    L = _lib.lib()
    for idx in _chunks(live):
        dev = moved[idx[0]].device
        nb = len(idx)
        ns = [len(moved[j]) for j in idx]
        corr = [torch.empty(x, dtype=torch.int32, device=dev) for x in ns]
        d2 = [torch.empty(x, dtype=_D, device=dev) for x in ns]
        sums = torch.empty(nb, 29, dtype=_D, device=dev)
        st0 = np.zeros((nb, 40))
        for k, j in enumerate(idx):
            st0[k, 5:21] = Ts[live[j]].reshape(-1)
        state = torch.from_numpy(st0).to(dev)
        ws = torch.empty(nb * 512 * 29 * 8, dtype=torch.uint8, device=dev)
        left, first, chunk = int(criteria.max_iteration), 1, PC._ICP_CHUNK
        if PC.ICP_STATS is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        while True:
            n_it = min(chunk, left)
            rc = L.ape_icp_run_batch_f64(kind, nb, *_grid_args([grids[j] for j in idx]), float(max_correspondence_distance), _ptrs([moved[j]._p for j in idx]),
                                         _ints(ns), _ptrs([tg[j]._p for j in idx]), _ptrs([tg[j]._n for j in idx]) if kind == 1 else None,
                                         float(max_correspondence_distance), float(criteria.relative_fitness), float(criteria.relative_rmse),
                                         int(criteria.max_iteration), n_it, first, _ptrs(corr), _ptrs(d2), _ptrs([sums[k] for k in range(nb)]),
                                         _ptrs([state[k] for k in range(nb)]), _lib.dptr(ws), ws.numel(), _st())
            _lib.check(rc, "ape_icp_run_batch_f64")
            out = state.cpu().numpy()
            left -= n_it
            first = 0
            if (out[:, 0] != 0.0).all() or left <= 0:
                break
            chunk *= 2
        if PC.ICP_STATS is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            with PC._ICP_STATS_LOCK:
                for k in range(nb):
                    ev = int(out[k, 1]) + 1
                    PC.ICP_STATS["registrations"] += 1
                    PC.ICP_STATS["evaluations"] += ev
                    PC.ICP_STATS["pairs"] += ev * ns[k]
                    PC.ICP_STATS["kind%d" % kind] = PC.ICP_STATS.get("kind%d" % kind, 0) + ev * ns[k]
                PC.ICP_STATS["events"].append((ev0, ev1))
        for k, j in enumerate(idx):
            Ts[live[j]] = out[k, 5:21].reshape(4, 4).copy()
    return Ts


# ---- the two stages of the label path, in lock step over chains -------------------------------------------------------------------------
def get_surface_batch(views, intr, min_friends, min_dist, nb_neighbors, voxel_size, device="cuda"):
    """[open3d_utils.get_surface(label, depth, intr, robot2cam, ...)] for many views at once (reference open3d_utils.py:171-213)"""
    clouds = voxel_down_sample(surface_points(views, intr, device), voxel_size)
    clouds = remove_radius_outlier(clouds, min_friends, min_dist)
    std_ratios = [np.abs(np.std(np.abs(m))) if len(m) else 0.0 for m in mahalanobis(clouds)]
    hint = float(min_dist)
    if voxel_size:
        hint = max(hint, 1.5 * float(voxel_size) * float(np.sqrt(nb_neighbors / np.pi)))
    return remove_statistical_outlier(clouds, nb_neighbors, std_ratios, hint)


def icp_regression_batch(targets, sources, voxel_size, threshold, icp_point2point=True, icp_point2plane=True):
    """[open3d_utils.icp_regression(t, s, ...)[2]] (reference :63-122): down-sample + normals of both clouds, p2p then p2plane ICP"""
    n = len(targets)
    tg = estimate_normals(voxel_down_sample(targets, voxel_size), voxel_size * 2, 30)        # preprocess_point_cloud(target.clone(), voxel)
    sr = estimate_normals(voxel_down_sample(sources, voxel_size), voxel_size * 2, 30)
    criteria = PC.ICPConvergenceCriteria(relative_fitness=1e-2, relative_rmse=1e-2, max_iteration=100)
    Ts = [np.identity(4) for _ in range(n)]
    if icp_point2point:
        Ts = registration_icp(sr, tg, threshold, Ts, 0, criteria)
    if icp_point2plane:
        Ts = registration_icp(sr, tg, threshold, Ts, 1, criteria)
    return Ts


def fuse_surfaces_batch(chains, voxel_size=2, threshold=10, voxel_size_out=None, icp_point2point=True, icp_point2plane=False):
    """[open3d_utils.fuse_surfaces(surfaces, ...)] for several chains in lock step: step v registers every chain's v-th surface to that chain's
    accumulating cloud (create_pointcloud.py:288-312).  Empty surfaces are skipped, a chain's first surface starts its cloud -- as in the
    one-chain loop.  -> [(cloud or None, [T per surface])]"""
    n = len(chains)
    acc = [None] * n
    tfs = [[] for _ in range(n)]
    for v in range(max((len(c) for c in chains), default=0)):
        work = []
        for ci, ch in enumerate(chains):
            if v >= len(ch):
                continue
            s = ch[v]
            if len(s) == 0:
                tfs[ci].append(None)
            elif acc[ci] is None:
                acc[ci] = s
                tfs[ci].append(np.identity(4))
            else:
                work.append(ci)
        if not work:
            continue
        srcs = [chains[ci][v] for ci in work]
        Ts = icp_regression_batch([acc[ci] for ci in work], srcs, voxel_size, threshold, icp_point2point, icp_point2plane)
        moved = transform(concat(srcs), Ts)                                               # source.clone().transform(T)
        merged = voxel_down_sample(concat(moved, [acc[ci] for ci in work]), voxel_size)   # cat([source, acc]) -> voxel_down_sample
        for k, ci in enumerate(work):
            tfs[ci].append(Ts[k])
            acc[ci] = merged[k]
    if voxel_size_out:
        live = [ci for ci in range(n) if acc[ci] is not None]
        down = voxel_down_sample([acc[ci] for ci in live], voxel_size_out)
        for k, ci in enumerate(live):
            acc[ci] = down[k]
    return [(acc[ci], tfs[ci]) for ci in range(n)]
