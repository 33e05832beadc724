"""Label generation sharded over two ranks (BASELINE configs[4]) with the REAL kernels: two processes (gloo rendezvous,
both computing on cuda:0 -- the box has one GPU) run `fuse_views` / `load_point_cloud` with the per-view work split between them; the
chain owner's cloud and every exported file must equal the single-rank run bit for bit."""
import hashlib
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from autoposeestimation_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _tree(root, obj, n_views):
    from autoposeestimation_amd.data_generation import sample_io as io
    cloud = S.bumpy_sphere(200000, 21)
    os.makedirs(os.path.join(root, "data_generation/data", obj, "background"), exist_ok=True)
    for di, d in enumerate(("foreground", "foreground180")):
        for i, (label, depth, cam) in enumerate(S.label_views(n_views, seed=3 + di, cloud=cloud)):
            rgb = np.full((480, 640, 3), 120, np.uint8)
            meta = {"intr": dict(S.LABEL_INTR), "depth_scale": 0.001, "hand_eye_calibration": list(np.eye(4).flatten()),
                    "robot2endEff_tf": list(cam.flatten()), "object_pose": list(np.eye(4).flatten()), "view_point_id": i}
            io.write_sample(os.path.join(root, "data_generation/data", obj, d), "{:06d}".format(i), rgb, depth, meta)
            io.write_label(os.path.join(root, "label_generator/data", obj, d), "{:06d}".format(i), "pred", label)


def _digests(save_dir, obj):
    out = {}
    for f in sorted(os.listdir(os.path.join(save_dir, obj))):
        out[f] = hashlib.sha256(open(os.path.join(save_dir, obj, f), "rb").read()).hexdigest()
    return out


def _worker(rank, world, port, root, q):
    from autoposeestimation_amd.pc_reconstruction import open3d_utils as U
    from autoposeestimation_amd.pc_reconstruction.create_pointcloud import load_point_cloud
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    views = S.label_views(5, seed=11, cloud=S.bumpy_sphere(200000, 21))
    cloud, tfs = U.fuse_views(views, S.LABEL_INTR, voxel_size=2, threshold=10, icp_point2point=True, icp_point2plane=True, dist=dist, owner=1)
    fused = None if cloud is None else np.array(cloud.points)
    out = load_point_cloud("ball", os.path.join(root, "pc2"), root, mode="pred", n_viewpoints=4, min_friends=20, min_dist=5, nb_neighbors=20,
                           threshold=10, voxel_size=2, voxel_size_out=5, icp_point2point=True, icp_point2plane=False,
                           rng=np.random.default_rng(1), dist=dist)
    q.put((rank, fused, None if out is None else np.array(out.points)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_label_generation_equals_single_rank(tmp_path):
    from autoposeestimation_amd.pc_reconstruction import open3d_utils as U
    from autoposeestimation_amd.pc_reconstruction.create_pointcloud import load_point_cloud
    root = str(tmp_path)
    _tree(root, "ball", 6)
    views = S.label_views(5, seed=11, cloud=S.bumpy_sphere(200000, 21))
    want_cloud, _ = U.fuse_views(views, S.LABEL_INTR, voxel_size=2, threshold=10, icp_point2point=True, icp_point2plane=True)
    want_fused = np.array(want_cloud.points)
    want_out = np.array(load_point_cloud("ball", os.path.join(root, "pc1"), root, mode="pred", n_viewpoints=4, min_friends=20, min_dist=5,
                                         nb_neighbors=20, threshold=10, voxel_size=2, voxel_size_out=5, icp_point2point=True,
                                         icp_point2plane=False, rng=np.random.default_rng(1)).points)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, fused, out = q.get()
        got[r] = (fused, out)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got[0][0] is None and np.array_equal(got[1][0], want_fused)           # the chain belongs to rank 1
    assert got[1][1] is None and np.array_equal(got[0][1], want_out)             # rank 0 aligns and exports
    assert _digests(os.path.join(root, "pc1"), "ball") == _digests(os.path.join(root, "pc2"), "ball")     # every exported file, byte for byte
    assert len(want_fused) > 500 and len(want_out) > 500


def test_fuse_chains_equals_fuse_views_per_chain_and_resident_views():
    """Several chains through one call (open3d_utils.fuse_chains, single rank) give each chain's fuse_views result bit for bit, and views
    handed over as resident device tensors (u8 label, u16 depth) give the same cloud as the host arrays."""
    from autoposeestimation_amd import synthetic as S
    from autoposeestimation_amd.pc_reconstruction import open3d_utils as U
    cloud = S.bumpy_sphere(120000, 21)
    chains = [S.label_views(4, seed=c, cloud=cloud) for c in range(3)]
    kw = dict(voxel_size=2, threshold=10, icp_point2point=True, icp_point2plane=True)
    multi = U.fuse_chains(chains, S.LABEL_INTR, **kw)
    assert sorted(multi) == [0, 1, 2]
    for c, views in enumerate(chains):
        one, tfs = U.fuse_views(views, S.LABEL_INTR, **kw)
        assert torch.equal(one._p, multi[c][0]._p)
        assert all(np.array_equal(a, b) for a, b in zip(tfs, multi[c][1]))
    resident = [(torch.from_numpy(l).cuda(), torch.from_numpy(d).cuda(), cam) for (l, d, cam) in chains[1]]
    assert resident[0][1].dtype == torch.uint16
    r, _ = U.fuse_views(resident, S.LABEL_INTR, **kw)
    assert torch.equal(r._p, multi[1][0]._p)
