"""End-to-end "Create Pose labels" on a synthetic dataset TREE in the reference's on-disk format:
PNG/JSON samples -> load_point_cloud -> exported clouds + .xyz -> create_pose_label -> per-sample pose-label JSON."""
import json
import math
import os

import numpy as np
import pytest

from oracle import pointcloud_oracle as PO
from test_gpu_pointcloud import INTR, _bumpy_sphere, _render, _rot

pytestmark = pytest.mark.gpu
CENTRE = np.array([400.0, -20.0, 150.0])


def _make_tree(root, obj, n_views):
    from autoposeestimation_amd.data_generation import sample_io as io
    cloud = _bumpy_sphere(300000, 21)
    data_dir = os.path.join(root, "data_generation/data", obj, "foreground")
    label_dir = os.path.join(root, "label_generator/data", obj, "foreground")
    os.makedirs(os.path.join(root, "data_generation/data", obj, "background"))
    rng = np.random.default_rng(0)
    cams = []
    for i in range(n_views):
        ang_y, ang_x = rng.uniform(-1.0, 1.0), rng.uniform(-0.5, 0.5)
        cam = _rot(math.pi, 0.0, 0.0, tuple(CENTRE + [0, 0, 500.0]))
        cam = _rot(0, 0, 0, tuple(CENTRE)) @ _rot(ang_x, ang_y, 0.0, (0, 0, 0)) @ _rot(0, 0, 0, tuple(-CENTRE)) @ cam
        depth = _render(cloud, cam)
        label = (depth != 0).astype(np.uint8) * 255
        rgb = np.full((480, 640, 3), 120, np.uint8)
        rgb[label != 0] = (230, 40, 40)
        meta = {"intr": dict(INTR), "depth_scale": 0.001, "hand_eye_calibration": list(np.eye(4).flatten()),
                "robot2endEff_tf": list(cam.flatten()), "object_pose": list(np.eye(4).flatten()), "view_point_id": i}
        io.write_sample(data_dir, "{:06d}".format(i), rgb, depth, meta)
        io.write_label(label_dir, "{:06d}".format(i), "pred", label)
        cams.append(cam)
    return cams


def test_load_point_cloud_and_pose_labels_on_disk(tmp_path):
    from autoposeestimation_amd.data_generation import sample_io as io
    from autoposeestimation_amd.label_generator.create_labels import create_pose_label
    from autoposeestimation_amd.pc_reconstruction import pointcloud as PC
    from autoposeestimation_amd.pc_reconstruction.create_pointcloud import load_point_cloud
    from autoposeestimation_amd.pipeline.utils import read_xyz_cloud
    root, obj = str(tmp_path), "ball"
    cams = _make_tree(root, obj, 12)
    # sample round trip
    d = os.path.join(root, "data_generation/data", obj, "foreground")
    assert io.list_samples(d) == ["{:06d}".format(i) for i in range(12)]
    assert io.read_depth(d, "000003").dtype == np.uint16 and io.read_color(d, "000003").shape == (480, 640, 3)
    np.testing.assert_allclose(io.robot2cam(io.read_meta(d, "000005")), cams[5])

    save_dir = os.path.join(root, "pc_reconstruction/data")
    out = load_point_cloud(obj, save_dir, root, mode="pred", n_viewpoints=8, min_friends=20, min_dist=5, nb_neighbors=20, threshold=10,
                           voxel_size=2, voxel_size_out=5, icp_point2point=True, icp_point2plane=False, rng=np.random.default_rng(1))
    for f in ("foreground.ply", "foreground.pcd", obj + "_out.ply", obj + "_out.pcd", obj + ".ply", obj + ".pcd", obj + ".xyz"):
        assert os.path.exists(os.path.join(save_dir, obj, f)), f
    # the fused cloud lies on the rendered surface (renders quantise depth to 1 mm)
    from scipy.spatial import cKDTree
    dist, _ = cKDTree(_bumpy_sphere(300000, 21)).query(np.array(out.points))
    assert np.quantile(dist, 0.99) < 4.0     # two 2 mm voxels: 1 mm depth quantisation + voxel means + sparse surface sample
    back = PC.read_point_cloud(os.path.join(save_dir, obj, obj + "_out.ply"))
    assert np.array_equal(np.array(back.points), np.array(out.points))          # %.17g round trip
    xyz = read_xyz_cloud(os.path.join(save_dir, obj, obj + ".xyz"), to_meter=False)
    assert len(xyz) >= 1000 and np.abs(xyz.max(0) + xyz.min(0)).max() < 8.0      # centred BEFORE the final voxel thinning, like the reference (:358-370)
    n = create_pose_label(root, obj, False, True, False)
    assert n == 12
    centre = np.array(out.points).min(0) + (np.array(out.points).max(0) - np.array(out.points).min(0)) / 2
    for i in (0, 7):
        with open(os.path.join(root, "label_generator/data", obj, "foreground", "{:06d}.meta.json".format(i))) as f:
            lab = json.load(f)
        want = np.linalg.inv(cams[i]) @ np.r_[centre, 1.0]
        np.testing.assert_allclose(lab["position"], want[:3], atol=1e-6)
        np.testing.assert_allclose(np.array(lab["rotation"]).reshape(3, 3), np.linalg.inv(cams[i])[:3, :3], atol=1e-9)
        assert np.linalg.norm(np.array(lab["position"]) - (np.linalg.inv(cams[i]) @ np.r_[CENTRE, 1.0])[:3]) < 15.0   # near the true centre


def test_create_pose_data_end_to_end(tmp_path):
    """main.py 'Create Pose labels' -> create_pose_data on the synthetic tree, with a PsPNet segmentor whose last layer is
    least-squares fitted to the rendered object (no checkpoint on disk: `model=` injection)."""
    import torch
    from autoposeestimation_amd import engine as E, synthetic as S
    from autoposeestimation_amd.data_generation import sample_io as io
    from autoposeestimation_amd.label_generator.create_labels import create_pose_data
    from autoposeestimation_amd.segmentation.utils import get_model
    root, obj = str(tmp_path), "ball"
    _make_tree(root, obj, 10)
    d = os.path.join(root, "data_generation/data", obj, "foreground")
    seg = get_model("PsPNet", {"encoder_name": "resnet18", "encoder_weights": None, "activation": "softmax", "in_channels": 3, "classes": 2})
    sd = S.pspnet_state_dict("resnet18", seed=5, stem_gain=1.0)
    seg.load_state_dict(sd)
    seg = seg.cuda().eval()
    feats, labels = [], []
    for sid in ("000000", "000001"):
        rgb = torch.from_numpy(io.read_color(d, sid)[None]).cuda()
        lab = io.read_label(os.path.join(root, "label_generator/data", obj, "foreground"), sid, "pred").reshape(-1) != 0
        f = seg.plan().features(E.preprocess_u8(rgb, torch.zeros(1, 3, dtype=torch.int32).cuda(), 480, 640, True))[0].reshape(-1, 64)
        feats.append(f)
        labels.append(torch.from_numpy(lab.astype(np.int64)))
    w, b = S.fit_final_layer(torch.cat(feats), torch.cat(labels), 2)
    fw, fb = sd["final.0.weight"].clone(), sd["final.0.bias"].clone()
    fw[:2, :, 0, 0], fb[:2] = w, b
    sd["final.0.weight"], sd["final.0.bias"] = fw, fb
    seg.load_state_dict(sd)
    stats, times = create_pose_data(root, [obj], "synthetic", reference_point=CENTRE, new_pred=True, model=seg.cuda().eval(), n_viewpoints=6,
                                    batch=5)
    assert stats["n_samples"] + stats["bs_copied"] >= 8, stats
    lab_dir = os.path.join(root, "label_generator/data", obj, "foreground")
    new = io.read_label(lab_dir, "000004", "new_pred")
    old = io.read_label(lab_dir, "000004", "pred")
    inter = ((new != 0) & (old != 0)).sum() / max(1, ((new != 0) | (old != 0)).sum())
    assert inter > 0.8, inter                                               # the relabelled mask is the rendered object
    assert os.path.exists(os.path.join(root, "pc_reconstruction/data", obj, obj + ".xyz"))
    assert os.path.exists(os.path.join(lab_dir, "000004.meta.json"))


def test_pose_dataset_and_addS_eval_harness(tmp_path):
    """PoseDataset (test mode) + experiments/eval.py on the synthetic tree; ADD-S of every sample must agree
    with the CPU oracle chain (posenet -> Loss -> 2 x (refiner -> Loss_refine)) within 1e-4 m."""
    import torch
    from autoposeestimation_amd import synthetic as S
    from autoposeestimation_amd.DenseFusion.datasets.myDatasetAugmented.dataset import PoseDataset, get_bbox
    from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet
    from autoposeestimation_amd.experiments.eval import eval as addS_eval
    from autoposeestimation_amd.label_generator.create_labels import create_pose_label
    from autoposeestimation_amd.pc_reconstruction.create_pointcloud import load_point_cloud
    from oracle import densefusion_oracle as O
    root, obj = str(tmp_path), "ball"
    _make_tree(root, obj, 8)
    # `symmetric` flag in the sample meta (getData.py:177-221) makes the class use ADD-S (k-NN) in Loss_refine
    d = os.path.join(root, "data_generation/data", obj, "foreground")
    for f in os.listdir(d):
        if f.endswith(".meta.json"):
            m = json.load(open(os.path.join(d, f)))
            m["symmetric"] = True
            json.dump(m, open(os.path.join(d, f), "w"))
    load_point_cloud(obj, os.path.join(root, "pc_reconstruction/data"), root, mode="pred", n_viewpoints=6, min_friends=20, min_dist=5,
                     nb_neighbors=20, threshold=10, voxel_size=2, voxel_size_out=5, icp_point2point=True, icp_point2plane=False,
                     rng=np.random.default_rng(1))
    create_pose_label(root, obj, False, True, False)
    ds_dir = os.path.join(root, "label_generator/data_sets/pose_estimation/synth")
    os.makedirs(ds_dir)
    open(os.path.join(ds_dir, "classes.txt"), "w").write(obj + "\n")
    open(os.path.join(ds_dir, "test_data_list.txt"), "w").write("".join("%s/foreground/%06d\n" % (obj, i) for i in (1, 4, 6)))
    ds = PoseDataset("test", 500, False, 0.0, True, "synth", root, label_mode="pred")
    assert len(ds) == 3 and ds.get_sym_list() == [0] and ds.get_num_points_mesh() == 1000
    pts, choose, img, target, model, idx, intr, np_img = ds[0]
    assert pts.shape == (500, 3) and choose.shape == (1, 500) and target.shape == (1000, 3) and model.shape == (1000, 3)
    lab = np.array(np_img)[:, :, 0] > 200
    assert img.shape[1:] == tuple(np.subtract(get_bbox(lab)[1::2], get_bbox(lab)[0::2]))
    assert get_bbox(lab) == O.get_bbox(lab)
    est_sd, ref_sd = S.posenet_state_dict(1, 0), S.refiner_state_dict(1, 0)
    est = PoseNet(500, 1)
    est.load_state_dict(est_sd)
    ref = PoseRefineNet(500, 1)
    ref.load_state_dict(ref_sd)
    res = addS_eval(500, True, "synth", False, "pred", 0.0, 1.0, est.cuda().eval(), 0.015, ref.cuda().eval(), 2, 0, [obj], root=root)
    got = res[obj]["dis_all"]
    assert len(got) == 3 and res[obj]["<2"] + res[obj][">=2"] == 3
    for j in range(3):   # oracle chain on the same sample tuples
        pts, choose, img, target, model, idx, _, _ = ds[j]
        with torch.no_grad():
            pr, pt, pc, emb = O.posenet_forward(est_sd, img[None], pts[None], choose[None], idx.view(1, 1), 1)
            _, dis, newp, newt, _ = O.loss_forward(pr, pt, pc, target[None], model[None], idx, pts[None], 0.015, True, 1000, [0])
            for _ in range(2):
                rr, rt = O.refiner_forward(ref_sd, newp, emb, idx.view(1, 1), 1)
                dis, newp, newt, _ = O.loss_refine_forward(rr, rt, newt, model[None], idx, newp, 1000, [0])
        assert abs(got[j] - float(dis)) <= 1e-4 * max(1.0, abs(float(dis))), (j, got[j], float(dis))
