"""Drop-in surface of pc_reconstruction/open3d_utils.py.

Host numpy helpers the live path touches (visualisation / bookkeeping): `points2pixel` :233-243, `pixels2points`
:215-231, `pointcloud2image` :246-270, `get_my_source_center` :273-292.
The open3d-backed half -- `get_surface` :171-213, `preprocess_point_cloud` :19-33, `icp_regression` :63-122,
`refine_registration` :51-59, `align_point_clouds` :125-168 -- runs on the gfx950 float64 point-cloud kernels through
`pc_reconstruction.pointcloud.PointCloud` (open3d itself is a third-party dependency absent from the reference tree:
parity UNPINNED, semantics restated from open3d 0.9).  `fuse_views` is the array-level core of
create_pointcloud.load_point_cloud :276-312 (its PNG/JSON directory walking is SURVEY.md 8f "next")."""
import numpy as np


def pixels2points(pixels, depth, intr):
    pixels = np.asarray(pixels).reshape(-1, 2)
    p2 = depth[pixels[:, 0], pixels[:, 1]]
    keep = p2 != 0
    py, px, p2 = pixels[keep, 0], pixels[keep, 1], p2[keep]
    p0 = (px - intr.get("ppx")) * p2 / intr.get("fx")
    p1 = (py - intr.get("ppy")) * p2 / intr.get("fy")
    return [[a, b, c] for a, b, c in zip(p0, p1, p2)]


def points2pixel(points, intr):
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    p1 = (x / (z / intr.get("fx")) + intr.get("ppx")).astype(np.int64)     # int() truncates toward zero
    p0 = (y / (z / intr.get("fy")) + intr.get("ppy")).astype(np.int64)
    return [[int(a), int(b)] for a, b in zip(p0, p1)]


def pointcloud2image(image, point_cloud, point_size, intr, color=None):
    step = int((point_size - 1) / 2)
    mark = np.zeros((point_size, point_size, 3))
    if not color:
        mark[:, :, 0] = 255
    else:
        mark[:, :, :] = np.asarray(color)[:3]
    points = point_cloud if isinstance(point_cloud, np.ndarray) else np.array(point_cloud.points)
    for r, c in points2pixel(points, intr):
        patch = image[r - step:r + step + 1, c - step:c + step + 1, :]
        if r - step < 0 or c - step < 0 or patch.shape != mark.shape:
            continue                     # the reference swallows the broadcast error for stamps leaving the image
        image[r - step:r + step + 1, c - step:c + step + 1, :] = mark * 0.3 + patch * 0.7
    return image


def get_my_source_center(source):
    pts = np.array(source.points)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    return lo + (hi - lo) / 2


# ---- the open3d-backed half, on the gfx950 point-cloud kernels ----------------------------
from autoposeestimation_amd.pc_reconstruction import pointcloud as _pc  # noqa: E402


def preprocess_point_cloud(pcd, voxel_size):
    """reference :19-33.  Down-sample + normals (hybrid radius 2*voxel, max_nn 30).  The FPFH feature the reference also
    computes (:29-32) is only consumed by the global RANSAC registration, which every caller disables
    (main.py:177, create_labels.py:229) -- it is returned as None."""
    pcd_down = pcd.voxel_down_sample(voxel_size)
    pcd_down.estimate_normals(_pc.KDTreeSearchParamHybrid(radius=voxel_size * 2, max_nn=30))
    return pcd_down, None


def refine_registration(source, target, result_ransac, voxel_size):
    """reference :51-59: point-to-plane ICP at 0.4 * voxel from a previous result"""
    return _pc.registration_icp(source, target, voxel_size * 0.4, result_ransac.transformation,
                                _pc.TransformationEstimationPointToPlane())


def icp_regression(target, source, voxel_size=5, threshold=100, global_regression=False, icp_point2point=True,
                   icp_point2plane=True, plot=False):
    """reference :63-122.  Returns (target_down, source_down, T); like the reference, `target` is copied and `source` is not
    (both are only read).  global_regression (FPFH + RANSAC) is not provided."""
    if global_regression:
        raise NotImplementedError("global RANSAC registration is disabled by every caller of the reference and is not built")
    target, _ = preprocess_point_cloud(target.clone(), voxel_size)
    source, _ = preprocess_point_cloud(source, voxel_size)
    init_tf = np.identity(4)
    criteria = _pc.ICPConvergenceCriteria(relative_fitness=1e-2, relative_rmse=1e-2, max_iteration=100)
    if icp_point2point:
        init_tf = _pc.registration_icp(source, target, threshold, init_tf, _pc.TransformationEstimationPointToPoint(),
                                       criteria).transformation
    if icp_point2plane:
        init_tf = _pc.registration_icp(source, target, threshold, init_tf, _pc.TransformationEstimationPointToPlane(),
                                       criteria).transformation
    return target, source, init_tf


def _post_filter(cloud, min_friends, min_dist, nb_neighbors, voxel_size=None):
    cloud, _ = cloud.remove_radius_outlier(nb_points=min_friends, radius=min_dist)
    std_ratio = np.abs(np.std(np.abs(np.array(cloud.compute_mahalanobis_distance()))))
    # search-cell hint for the k-NN (speed only, any value is exact): after the radius filter every point has min_friends neighbours
    # within min_dist; on a surface voxelised at `voxel_size` the k nearest lie within ~voxel * sqrt(k / pi)
    hint = float(min_dist)
    if voxel_size:
        hint = max(hint, 1.5 * float(voxel_size) * float(np.sqrt(nb_neighbors / np.pi)))
    cloud, _ = cloud.remove_statistical_outlier(nb_neighbors=nb_neighbors, std_ratio=std_ratio, cell_hint=hint)
    return cloud


def align_point_clouds(point_clouds, min_friends, min_dist, nb_neighbors, plot=False, global_regression=False,
                       icp_point2point=True, icp_point2plane=False, voxel_size=5, threshold=50):
    """reference :125-168: register every further cloud to the growing target with p2p ICP, merge, down-sample, filter"""
    target = point_clouds[0]
    for source in point_clouds[1:]:
        diff = np.array(source.get_center()) - np.array(target.get_center())
        if diff[1] > -30:
            source.translate(translation=[0, -30 - diff[1], 0])
        target, source, init_tf = icp_regression(target, source, voxel_size=voxel_size, threshold=threshold,
                                                 global_regression=global_regression, icp_point2point=True,
                                                 icp_point2plane=False, plot=plot)
        source = source.transform(init_tf)
        target.points = np.concatenate((np.array(source.points), np.array(target.points)))
        target = target.voxel_down_sample(voxel_size=voxel_size)
        target = _post_filter(target, min_friends, min_dist, nb_neighbors, voxel_size)
    return target


def get_surface(label, depth_frame, intr, robot2Cam_ft, min_friends, min_dist, nb_neighbors, voxel_size):
    """reference :171-213: label & depth pixels -> robot-frame cloud (mm) -> voxel down-sample -> radius filter ->
    statistical filter with std_ratio = std(|mahalanobis|)."""
    surface = _pc.surface_points(label, depth_frame, intr, robot2Cam_ft)
    surface = surface.voxel_down_sample(voxel_size=voxel_size)
    return _post_filter(surface, min_friends, min_dist, nb_neighbors, voxel_size)


def fuse_surfaces(surfaces, voxel_size=2, threshold=10, voxel_size_out=None, icp_point2point=True, icp_point2plane=False):
    """The sequential accumulation at the heart of load_point_cloud (pc_reconstruction/create_pointcloud.py:288-312) over already
    pre-processed surfaces (PointClouds in view order, empty ones skipped): each is registered to the accumulating cloud, merged, and
    the union is voxel down-sampled.  Order-dependent: one rank runs a chain.  Returns (cloud, [T per surface])."""
    acc, tfs = None, []
    for source in surfaces:
        if len(source) == 0:
            tfs.append(None)
            continue
        if acc is None:
            acc = source
            tfs.append(np.identity(4))
            continue
        _, _, T = icp_regression(acc, source, voxel_size=voxel_size, threshold=threshold, global_regression=False,
                                 icp_point2point=icp_point2point, icp_point2plane=icp_point2plane)
        tfs.append(T)
        merged = _pc.PointCloud(device=acc.device)
        merged.points = torch.cat([source.clone().transform(T)._p, acc._p], 0)
        acc = merged.voxel_down_sample(voxel_size)
    if acc is not None and voxel_size_out:
        acc = acc.voxel_down_sample(voxel_size_out)
    return acc, tfs


def fuse_views(views, intr, voxel_size=2, threshold=10, min_friends=20, min_dist=5, nb_neighbors=20, voxel_size_out=None,
               icp_point2point=True, icp_point2plane=False, dist=None, owner=0):
    """`views` = sequence of (label u8[H,W], depth [H,W], robot2cam 4x4) of ONE (object, direction) chain (create_pointcloud.py:276-312):
    get_surface per view, then fuse_surfaces.  With a torch.distributed group (`dist`) the per-view get_surface work is sharded over the
    ranks and one padded all-gather hands the surfaces to the chain's `owner`, which fuses them (sharding.sharded_chain);
    the other ranks return (None, None).  The result on the owner is bit-identical to the single-rank call."""
    from autoposeestimation_amd import sharding
    make_set, fuse = _chain_workers(intr, voxel_size, threshold, min_friends, min_dist, nb_neighbors, voxel_size_out, icp_point2point,
                                    icp_point2plane)
    res = sharding.sharded_chain(list(views), make_set, fuse, owner, dist, load=_load_view)
    return res if res is not None else (None, None)


def _load_view(view):
    """a callable view decodes its files (host threads, ahead of the GPU: sharding.prefetched); arrays / tensors pass through"""
    return view() if callable(view) else view


def _chain_workers(intr, voxel_size, threshold, min_friends, min_dist, nb_neighbors, voxel_size_out, icp_point2point, icp_point2plane):
    def make_set(view):
        view = view() if callable(view) else view                           # (already decoded when it came through _load_view)
        label, depth, robot2cam = view[:3]
        view_intr = view[3] if len(view) > 3 and view[3] is not None else intr   # a view may carry its own meta['intr'] (reference create_pointcloud.py:55)
        return get_surface(label, depth, view_intr, robot2cam, min_friends, min_dist, nb_neighbors, voxel_size)._p

    def fuse(sets):
        dev = torch.device("cuda", torch.cuda.current_device())
        clouds = []
        for p in sets:
            c = _pc.PointCloud(device=dev)
            c.points = p.to(dev)
            clouds.append(c)
        return fuse_surfaces(clouds, voxel_size=voxel_size, threshold=threshold, voxel_size_out=voxel_size_out,
                             icp_point2point=icp_point2point, icp_point2plane=icp_point2plane)
    return make_set, fuse


def fuse_chains(chains, intr, voxel_size=2, threshold=10, min_friends=20, min_dist=5, nb_neighbors=20, voxel_size_out=None,
                icp_point2point=True, icp_point2plane=False, dist=None):
    """`fuse_views` for several (object, direction) chains at once (`chains` = list of view lists): the get_surface work of all views
    of all chains is spread over the ranks, one padded all-gather, then every rank runs the sequential fusion of the chains it owns
    (chain i -> rank i % world) -- the chains' ICP sequences proceed in parallel on different GPUs.
    Returns {chain index: (cloud, [T per view])} for this rank's chains; each is bit-identical to the single-rank fuse_views of it."""
    from autoposeestimation_amd import sharding
    make_set, fuse = _chain_workers(intr, voxel_size, threshold, min_friends, min_dist, nb_neighbors, voxel_size_out, icp_point2point,
                                    icp_point2plane)
    if not USE_BATCHED:
        return sharding.sharded_chains([list(v) for v in chains], make_set, fuse, dist, load=_load_view)
    # lock-step batched form (pc_reconstruction/batched.py): one host thread, one launch per step for all of this rank's views / chains
    from autoposeestimation_amd.pc_reconstruction import batched as B

    def make_sets(views):
        views = [v() if callable(v) else v for v in views]
        return [c._p for c in B.get_surface_batch(views, intr, min_friends, min_dist, nb_neighbors, voxel_size)]

    def fuse_many(parts):
        dev = torch.device("cuda", torch.cuda.current_device())
        clouds = []
        for sets in parts:
            row = []
            for p in sets:
                c = _pc.PointCloud(device=dev)
                c.points = p.to(dev)
                row.append(c)
            clouds.append(row)
        return B.fuse_surfaces_batch(clouds, voxel_size=voxel_size, threshold=threshold, voxel_size_out=voxel_size_out,
                                     icp_point2point=icp_point2point, icp_point2plane=icp_point2plane)

    return sharding.sharded_chains([list(v) for v in chains], make_set, fuse, dist, load=_load_view, make_sets=make_sets, fuse_many=fuse_many)


import os  # noqa: E402

import torch  # noqa: E402

# fuse_chains advances all of a rank's views / chains in lock step, one launch per step (batched.py); APE_LABEL_BATCHED=0 keeps the round-2 form
# (a host thread and a HIP stream per unit, sharding.run_side_by_side) -- both give the same clouds bit for bit
USE_BATCHED = os.environ.get("APE_LABEL_BATCHED", "1") != "0"
