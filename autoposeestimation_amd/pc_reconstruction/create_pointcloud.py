"""Array-level drop-in for pc_reconstruction/create_pointcloud.py: `get_view_distribution` (reference :46-174) and the
fusion / alignment / export tail of `load_point_cloud` (:181-378).  The reference walks directories of PNG/JSON samples;
that on-disk format is SURVEY.md 8f "next", so the entry points here take the decoded arrays.  Per-view geometry runs on
the GPU (open3d_utils.get_surface / icp_regression); view selection over <= ~200 camera positions is host numpy."""
import numpy as np

from autoposeestimation_amd.pc_reconstruction import open3d_utils as utils
from autoposeestimation_amd.pc_reconstruction import pointcloud as pc


def _voxel_mean_host(points, voxel):
    origin = points.min(0) - voxel * 0.5
    idx = np.floor((points - origin) / voxel).astype(np.int64)
    key = (idx[:, 0] << 42) | (idx[:, 1] << 21) | idx[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ends = np.r_[starts[1:], len(ks)]
    return np.array([points[order[a:b]].mean(0) for a, b in zip(starts, ends)])


def get_view_distribution(cam_positions, n_viewpoints, rng=None):
    """reference :46-174 on the camera positions robot2Cam[:3,3] of a directory's samples: thin them with a voxel grid
    whose size is searched in 1 mm steps until exactly `n_viewpoints` voxels remain (random subset when it overshoots),
    snap each voxel centroid to its nearest real view, then order the views greedily by nearest neighbour starting from
    the one closest to the robot origin.  Returns indices into `cam_positions`."""
    points = np.asarray(cam_positions, dtype=np.float64)
    n = len(points)
    if n_viewpoints > n:
        raise ValueError("more viewpoints requested than samples")
    rng = rng or np.random.default_rng()
    d = np.linalg.norm(points[:, None] - points[None], axis=2)
    np.fill_diagonal(d, np.inf)
    voxel_size = float(int(d.min()))                           # int(np.linalg.norm(...)) of the closest pair (:84-95)
    if voxel_size == 0:
        voxel_size = 1.0                                        # a zero voxel is meaningless (the reference would divide by it)
    while True:
        down = _voxel_mean_host(points, voxel_size)
        if len(down) == n_viewpoints:
            selected = down
            break
        if len(down) < n_viewpoints:
            voxel_size -= 1
            down = _voxel_mean_host(points, max(voxel_size, 1e-9))
            selected = down[rng.choice(len(down), replace=False, size=n_viewpoints)]
            break
        voxel_size += 1
    selection = np.array([int(np.argmin(np.linalg.norm(points - p, axis=1))) for p in selected])
    pts = points[selection]
    order = [int(np.argmin(np.linalg.norm(pts, axis=1)))]
    while len(order) != n_viewpoints:
        dist = np.linalg.norm(pts - pts[order[-1]], axis=1)
        dist[order] = np.inf
        order.append(int(np.argmin(dist)))
    return selection[order]


def fuse_direction(views, intr, point_cloud_tf=None, **kw):
    """One (object, rotation-directory) chain (reference :227-325): sequential registration of the selected views, then the
    directory's known object rotation applied about the cloud centre (:320).  `dist=` / `owner=` (keyword, passed on to
    open3d_utils.fuse_views) shard the per-view work over the ranks of a torch.distributed group; only `owner` gets the cloud."""
    cloud, tfs = utils.fuse_views(views, intr, **kw)
    if cloud is not None and point_cloud_tf is not None:
        cloud.rotate(R=np.asarray(point_cloud_tf, dtype=np.float64)[:3, :3], center=True)
    return cloud, tfs


def finish_object(point_clouds, min_friends=20, min_dist=5, nb_neighbors=20, voxel_size=2, voxel_size_out=5, threshold=10):
    """reference :331-376: align the per-directory clouds, then derive the three exported clouds: `<obj>_out` (robot
    frame), `<obj>` (centred, voxel_size_out) and the >= 1000-point `<obj>.xyz` model cloud used by DenseFusion (voxel grown
    in 0.1 steps until fewer than 1000 points would remain).  Returns (out, centred_down, xyz ndarray)."""
    out = utils.align_point_clouds(point_clouds, min_friends=min_friends, min_dist=min_dist, nb_neighbors=nb_neighbors,
                                   voxel_size=voxel_size, threshold=threshold)
    down = out.voxel_down_sample(voxel_size=voxel_size_out)
    down.translate(translation=-utils.get_my_source_center(down))
    big = out.clone()
    big.translate(translation=-utils.get_my_source_center(big))
    v = voxel_size
    while True:
        v += 0.1
        if len(big.voxel_down_sample(voxel_size=v)) < 1000:
            big = big.voxel_down_sample(voxel_size=v - 0.1)
            break
    return out, down, np.array(big.points)


def write_xyz(path, points):
    """`<obj>.xyz` exactly as the reference writes it (:373-376): one numpy-repr'd point per line, parsed back by
    pipeline/utils.py:667-684 (read_xyz_cloud)."""
    with open(path, "w") as f:
        for item in np.asarray(points):
            f.write("%s\n" % item)


def load_point_cloud(object_name, save_dir, root, reference_point=np.array([0, 0, 0]), mode="gen", n_viewpoints=10, min_friends=10,
                     voxel_size=5, voxel_size_out=10, threshold=50, min_dist=10, nb_neighbors=5, l_arrow=30,
                     global_regression=False, icp_point2point=True, icp_point2plane=True, plot=False, rng=None, dist=None):
    """Reference signature (pc_reconstruction/create_pointcloud.py:181-197) over the reference's directory layout:
    for every rotation directory of `label_generator/data/<obj>` select `n_viewpoints` views, fuse them sequentially on the
    GPU, rotate by the directory's object_pose, write `<d>.pcd/.ply`; then align the directories and export
    `<obj>_out.{pcd,ply}`, the centred `<obj>.{pcd,ply}` and the >= 1000-point `<obj>.xyz`.  Returns the `_out` cloud.

    `dist` (an initialised torch.distributed module; one process per GPU, shared file system): the rotation directories are the
    CHAINS -- directory i is fused by rank i % world -- the decode + get_surface work of all chains' views is spread over all ranks
    (each rank reads only the PNGs of its share; ONE padded all-gather), the owners then fuse their directories side by side
    (open3d_utils.fuse_chains), the per-directory clouds go to rank 0 with one more gather, and rank 0 aligns and exports (it alone
    returns the cloud; the others return None).  The view selection draws from `rng` on rank 0 ONLY and is broadcast, so the ranks
    cannot disagree about which views a flat index means (an unseeded or differently seeded `rng` per rank would otherwise mix
    different views into one cloud without any error).  Every view carries its own meta['intr'] like the reference loop (:55)."""
    from autoposeestimation_amd import sharding
    from autoposeestimation_amd.data_generation import sample_io as io
    dist_on = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if dist_on else 0
    world = dist.get_world_size() if dist_on else 1
    object_label_path = os.path.join(root, "label_generator/data", object_name)
    dirs = [d for d in sorted(os.listdir(object_label_path)) if d != "extra"]
    if not dirs:
        raise ValueError("no labels obtained yet")
    data_path = os.path.join(root, "data_generation/data", object_name)
    pcd_path = os.path.join(save_dir, object_name)
    os.makedirs(pcd_path, exist_ok=True)
    n = len([f for f in os.listdir(os.path.join(object_label_path, dirs[0])) if ".{}.label.png".format(mode) in f])
    chains, tfs, all_metas = [], [], []
    for di, d in enumerate(dirs):
        all_metas.append([io.read_meta(os.path.join(data_path, d), "{:06d}".format(i)) for i in range(n)])
    selection = None
    if rank == 0:
        selection = [[int(i) for i in get_view_distribution(np.array([io.robot2cam(m)[:3, 3] for m in metas]), n_viewpoints, rng)]
                     for metas in all_metas]
    if dist_on:
        box = [selection]
        dist.broadcast_object_list(box, src=0)
        selection = box[0]
    intr = None
    for di, d in enumerate(dirs):
        metas = all_metas[di]
        views, tf = [], None
        for idx in selection[di]:
            meta = metas[idx]
            tf = np.array(meta.get("object_pose"), dtype=np.float64).reshape(4, 4)[:3, :3]

            def decode(sid="{:06d}".format(idx), meta=meta, d=d):
                return (io.read_label(os.path.join(object_label_path, d), sid, mode), io.read_depth(os.path.join(data_path, d), sid),
                        io.robot2cam(meta), meta.get("intr"))
            views.append(decode)
        intr = metas[0].get("intr") if intr is None else intr         # default for views without their own (each view carries its own)
        chains.append(views)
        tfs.append(tf)
    # all directories at once: per-view work of every chain over all ranks, one all-gather, then the owners fuse side by side
    fused = utils.fuse_chains(chains, intr, voxel_size=voxel_size, threshold=threshold, min_friends=min_friends, min_dist=min_dist,
                              nb_neighbors=nb_neighbors, icp_point2point=icp_point2point, icp_point2plane=icp_point2plane,
                              dist=dist if dist_on else None)
    def export_mine():
        mine = []
        for di in sorted(fused):
            cloud, _ = fused[di]
            if cloud is None:
                raise ValueError("no valid surface in %s/%s" % (object_name, dirs[di]))
            if tfs[di] is not None:
                cloud.rotate(R=np.asarray(tfs[di], dtype=np.float64)[:3, :3], center=True)        # as fuse_direction (reference :320)
            pc.write_point_cloud(os.path.join(pcd_path, "{}.pcd".format(dirs[di])), cloud)
            pc.write_point_cloud(os.path.join(pcd_path, "{}.ply".format(dirs[di])), cloud)
            mine.append((di, cloud._p))
        return mine

    # an empty directory is an error on EVERY rank (the owner alone raising would leave the others in the gather below)
    mine = sharding.guarded(dist if dist_on else None, export_mine, "the per-directory export of load_point_cloud")
    sets = sharding.gather_point_sets(mine, len(dirs), dist if dist_on else None, owners=[0] * len(dirs) if dist_on else None)
    if rank != 0:
        return None
    point_clouds = [pc.PointCloud(p) for p in sets]
    out, down, xyz = finish_object(point_clouds, min_friends=min_friends, min_dist=min_dist, nb_neighbors=nb_neighbors,
                                   voxel_size=voxel_size, voxel_size_out=voxel_size_out, threshold=threshold)
    for ext in ("pcd", "ply"):
        pc.write_point_cloud(os.path.join(pcd_path, "{}_out.{}".format(object_name, ext)), out)
        pc.write_point_cloud(os.path.join(pcd_path, "{}.{}".format(object_name, ext)), down)
    write_xyz(os.path.join(pcd_path, "{}.xyz".format(object_name)), xyz)
    return out


import os  # noqa: E402

__all__ = ["get_view_distribution", "fuse_direction", "finish_object", "write_xyz", "load_point_cloud", "pc"]
