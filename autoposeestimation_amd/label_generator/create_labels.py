"""Drop-in for `get_default_model` of label_generator/create_labels.py (reference :20-37).

The reference hard-codes smp's Unet-resnet34; that third-party model is unavailable (see segmentation/utils.py), so the
default here is the in-repo PSPNet ('PsPNet', resnet34 encoder) with the same config keys and checkpoint convention:
`<root>/segmentation/trained_models/<ds_name>/<name>_<encoder>.ckpt` holding {'state_dict': ...}."""
import os

import torch

from autoposeestimation_amd.segmentation.utils import get_model


def get_default_model(root, ds_name, n_classes, name="PsPNet", encoder_name="resnet34", load=True):
    segmentation_config = {"encoder_name": encoder_name,
                           "encoder_weights": None,
                           "activation": "softmax",
                           "in_channels": 3,
                           "classes": n_classes}
    model = get_model(name, segmentation_config)
    if load:
        ckpt_dir = os.path.join(root, "segmentation", "trained_models", ds_name)
        path = os.path.join(ckpt_dir, "{}_{}.ckpt".format(name, segmentation_config["encoder_name"]))
        if not os.path.exists(path):
            smp_ckpt = os.path.join(ckpt_dir, "Unet_{}.ckpt".format(segmentation_config["encoder_name"]))
            hint = ""
            if os.path.exists(smp_ckpt):
                hint = (" -- {} is the reference's segmentation_models_pytorch Unet checkpoint (create_labels.py:20-35): its state-dict "
                        "layout belongs to a third-party model that is not part of the reference tree and cannot be loaded here; train the "
                        "in-repo 'PsPNet' segmentor and save it as {}".format(smp_ckpt, os.path.basename(path)))
            raise FileNotFoundError("segmentor checkpoint {} not found{}".format(path, hint))
        cp = torch.load(path, map_location=torch.device("cpu"))
        model.load_state_dict(cp["state_dict"])
    return model


# ---- "Create Pose labels": the per-frame relabelling and the pose-label maths (reference :96-214, :395-429) -------------
import math  # noqa: E402

import numpy as np  # noqa: E402

from autoposeestimation_amd import _lib  # noqa: E402
from autoposeestimation_amd import engine as E  # noqa: E402


def relabel_frames(model, rgb, depth, robot2cam, reference_point, class_id, bs_labels=None, is_extra=False):
    """Device form of the seg-relabel loop body (reference :101-205) for a batch of frames of ONE object class.

    rgb[B,H,W,3] u8 cuda, depth[B,H,W] u16 cuda, robot2cam[B,4,4] (robot2endEff_tf . hand_eye_calibration, :104-106),
    reference_point[3] mm, class_id 0-based, bs_labels[B,H,W] u8 cuda = the background-subtraction `.pred.label.png`
    (None for the 'extra' directories).  Returns (labels[B,H,W] u8 {0,255} cuda, save[B] bool, stats dict): frame i's
    `.new_pred.label.png` is labels[i] when save[i], otherwise the reference deletes stale outputs (:206-214)."""
    b, h, w, _ = rgb.shape
    dev = rgb.device
    target = class_id + 1
    n_cls = model.classes
    rects = torch.zeros(b, 3, dtype=torch.int32)
    rects[:, 0] = torch.arange(b, dtype=torch.int32)
    x4 = E.preprocess_u8(rgb, rects.to(dev), h, w, div255=True)
    if hasattr(model, "label_score_nhwc"):                                              # predict's softmax + F.softmax (:121-122)
        label, score = model.label_score_nhwc(x4, double_softmax=True)
    else:
        label, score = E.seg_argmax(model.logits_nhwc(x4), n_cls, double_softmax=True)
    label = torch.where(label == target, label, torch.zeros_like(label))                # pred_arg[pred_arg != class_id+1] = 0 (:128)
    objmap, _ = E.seg_components(label, score, n_cls, min_pixels=0)                      # best mean-probability component (:131-141)
    pos = np.asarray(robot2cam, dtype=np.float64).reshape(b, 4, 4)[:, :3, 3]
    dist = np.linalg.norm(np.asarray(reference_point, dtype=np.float64)[None] - pos, axis=1)
    gate = torch.from_numpy(np.stack([dist - 150, dist + 150], 1).astype(np.float32)).to(dev)   # :107-109
    counts = torch.zeros(b, 6, dtype=torch.int32, device=dev)
    rc = _lib.lib().ape_label_trust_counts(_lib.dptr(objmap, torch.uint8), target, _lib.dptr(bs_labels), _lib.dptr(depth, torch.uint16),
                                           _lib.dptr(gate), b, h, w, 30, 50, _lib.dptr(counts), _lib.stream_ptr())
    _lib.check(rc, "ape_label_trust_counts")
    c = counts.cpu().numpy()
    labels = torch.where(objmap == target, torch.full_like(objmap, 255), torch.zeros_like(objmap))
    save = np.zeros(b, bool)
    stats = {"bs_copied": 0, "no_depth_overlap": 0, "not_in_center": 0}
    for i in range(b):
        if not is_extra and bs_labels is not None and not (c[i, 0] > 0 and c[i, 1] > 0):      # len(np.unique(pred[bs != 0])) <= 1 (:169)
            labels[i] = bs_labels[i]
            save[i] = True
            stats["bs_copied"] += 1
            continue
        if not (c[i, 2] > 0 and c[i, 3] > 0):                                                   # :182-185
            stats["no_depth_overlap"] += 1
            continue
        if c[i, 4] > 0 and c[i, 5] > 0:                                                         # :187-193
            save[i] = True
        else:
            stats["not_in_center"] += 1
    return labels, save, stats


_EPS4 = np.finfo(float).eps * 4.0


def mat2euler(M):
    """transforms3d.euler.mat2euler(M) with its default axes='sxyz' (reference :343,368; UNPINNED third-party maths)."""
    M = np.asarray(M, dtype=np.float64)[:3, :3]
    cy = math.sqrt(M[0, 0] * M[0, 0] + M[1, 0] * M[1, 0])
    if cy > _EPS4:
        return math.atan2(M[2, 1], M[2, 2]), math.atan2(-M[2, 0], cy), math.atan2(M[1, 0], M[0, 0])
    return math.atan2(-M[1, 2], M[1, 1]), math.atan2(-M[2, 0], cy), 0.0


def euler2mat(ai, aj, ak):
    """transforms3d.euler.euler2mat(ai, aj, ak) with axes='sxyz': R = Rz(ak) Ry(aj) Rx(ai)"""
    ci, si, cj, sj, ck, sk = math.cos(ai), math.sin(ai), math.cos(aj), math.sin(aj), math.cos(ak), math.sin(ak)
    return np.array([[cj * ck, sj * si * ck - ci * sk, sj * ci * ck + si * sk],
                     [cj * sk, sj * si * sk + ci * ck, sj * ci * sk - si * ck],
                     [-sj, cj * si, cj * ci]])


def constrain_rotation(pc_rotation, init_tf):
    """reference :366-377: fold the ICP correction into the requested object rotation and keep only the Euler axes that
    were requested non-zero."""
    old = np.rad2deg(mat2euler(pc_rotation))
    new = np.dot(np.asarray(pc_rotation, float), np.asarray(init_tf, float)[:3, :3])
    euler = np.array(mat2euler(new))
    euler[old == 0.0] = 0.0
    return euler2mat(*euler)


def pose_label(meta, pc_position, pc_rotation, object_name):
    """reference :395-429: cam2object = inv(hand_eye) . inv(robot2endEff) . [pc_rotation | pc_position] -> label dict"""
    hand_eye = np.array(meta.get("hand_eye_calibration"), dtype=np.float64).reshape(4, 4)
    robot2end = np.array(meta.get("robot2endEff_tf"), dtype=np.float64).reshape(4, 4)
    robot2object = np.identity(4)
    robot2object[:3, :3] = pc_rotation
    robot2object[:3, 3] = pc_position
    cam2robot = np.dot(np.linalg.inv(hand_eye), np.linalg.inv(robot2end))
    cam2object = np.dot(cam2robot, robot2object)
    return {"position": list(cam2object[:3, 3]), "rotation": list(cam2object[:3, :3].flatten()), "cls_name": object_name,
            "cam2robot": list(cam2robot.flatten()), "robot2object": list(robot2object.flatten())}


# ---- directory drivers with the reference's signatures (create_labels.py:40-289, :292-440) ----------------------------
import json  # noqa: E402
import time  # noqa: E402


def create_pose_label(root, object_name, global_regression, icp_point2point, icp_point2plane, plot=False, view_label=False,
                      with_extra=False):
    """reference :292-440: per rotation directory find the object centre (bbox centre of `<obj>_out.ply`) and rotation
    (object_pose of the directory's samples; for rotated directories refined by ICP of the fused cloud onto the directory's
    cloud and constrained to the requested Euler axes), then write `<id>.meta.json` pose labels for every sample."""
    from autoposeestimation_amd.data_generation import sample_io as io
    from autoposeestimation_amd.pc_reconstruction import open3d_utils as pc_utils
    from autoposeestimation_amd.pc_reconstruction import pointcloud as pc
    object_path = os.path.join(root, "data_generation/data", object_name)
    dirs = sorted(os.listdir(object_path))
    if "background" not in dirs:
        raise ValueError("background does not exist in object_path: {}".format(object_path))
    dirs.remove("background")
    if "extra" in dirs:
        dirs.remove("extra")
        if with_extra:
            dirs.append("extra")
    if len(dirs) < 1:
        raise ValueError("no foreground")
    pc_path = os.path.join(root, "pc_reconstruction/data", object_name, "{}_out.ply".format(object_name))
    remember = []
    n_written = 0
    for d in dirs:
        pc_position = pc_rotation = None
        data_path = os.path.join(object_path, d)
        label_path = os.path.join(root, "label_generator/data", object_name, d)
        samples = io.list_samples(data_path)
        if d != "extra":
            source = pc.read_point_cloud(pc_path)
            pc_position = pc_utils.get_my_source_center(source)
            pc_rotation = np.array(io.read_meta(data_path, samples[0]).get("object_pose"), dtype=np.float64).reshape(4, 4)[:3, :3]
            old_rotation = np.rad2deg(mat2euler(pc_rotation))
            if not np.array_equal(old_rotation, np.array([0.0, 0.0, 0.0])):
                target = pc.read_point_cloud(os.path.join(root, "pc_reconstruction/data", object_name, "{}.ply".format(d)))
                _, source, init_tf = pc_utils.icp_regression(target, source, voxel_size=5, threshold=10, global_regression=global_regression,
                                                             icp_point2point=icp_point2point, icp_point2plane=icp_point2plane)
                pc_rotation = constrain_rotation(pc_rotation, init_tf)
                pc_position = np.array(pc_utils.get_my_source_center(source))
            remember.append({"old_rotation": old_rotation, "pc_position": pc_position, "pc_rotation": pc_rotation})
        for sid in samples:
            meta = io.read_meta(data_path, sid)
            if d == "extra":
                rot = np.rad2deg(mat2euler(np.array(meta.get("object_pose"), dtype=np.float64).reshape(4, 4)[:3, :3]))
                for r in remember:
                    if np.array_equal(rot, r["old_rotation"]):
                        pc_position, pc_rotation = r["pc_position"], r["pc_rotation"]
                        break
            os.makedirs(label_path, exist_ok=True)
            with open(os.path.join(label_path, "{}.meta.json".format(sid)), "w") as f:
                json.dump(pose_label(meta, pc_position, pc_rotation, object_name), f)
            n_written += 1
    return n_written


def create_pose_data(root, classes, ds_name, reference_point=np.array([]), new_pred=True, get_extra_labels=False, plot=False,
                     use_cuda=True, model=None, n_viewpoints=30, batch=16, dist=None, rng_seed=None):
    """reference :40-289 ("Create Pose labels"): per class (1) re-label every frame with the segmentor + trust checks
    (`<id>.new_pred.label.png`), (2) fuse the selected views into the object's point cloud, (3) write the pose labels.
    `model` (an already loaded segmentor) and `n_viewpoints` / `batch` are additions; everything else keeps the reference
    names, defaults and hyper-parameters (:219-231).

    `dist` (an initialised torch.distributed module, one process per GPU on a shared file system): the frames of the
    relabel loop are independent -- every rank relabels its contiguous share of each directory's samples and writes their PNGs --
    then load_point_cloud shards as described there and rank 0 writes the pose labels.  `rng_seed` seeds the view selection (needed
    with `dist`: all ranks must draw the same views).  The returned stats are summed over the ranks."""
    from autoposeestimation_amd import sharding
    from autoposeestimation_amd.data_generation import sample_io as io
    dist_on = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if dist_on else 0
    world = dist.get_world_size() if dist_on else 1
    if dist_on and rng_seed is None:
        rng_seed = 0
    from autoposeestimation_amd.pc_reconstruction.create_pointcloud import load_point_cloud
    if not (torch.cuda.is_available() and use_cuda):
        raise RuntimeError("create_pose_data needs the GPU: the MI355X path has no CPU fallback")
    device = torch.device("cuda:0")
    mode = "new_pred" if new_pred else "pred"
    if model is None:
        model = get_default_model(root, ds_name, len(classes) + 1)
    model.to(device).eval()
    stats = {"n_samples": 0, "n_extra_samples": 0, "bs_copied": 0, "no_depth_overlap": 0, "not_in_center": 0}
    times = {"seg": [], "pc": [], "pose": []}
    for class_id, cls in enumerate(classes):
        data_path = os.path.join(root, "data_generation", "data", cls)
        dirs = [d for d in sorted(os.listdir(data_path)) if d != "background" and (get_extra_labels or d != "extra")]
        t0 = time.time()
        for d in dirs:
            if not (d == "extra" or new_pred):
                continue
            data_dir = os.path.join(data_path, d)
            label_path = os.path.join(root, "label_generator/data", cls, d)
            os.makedirs(label_path, exist_ok=True)
            samples = io.list_samples(data_dir)
            lo, hi = sharding.shard_range(len(samples), rank, world)
            samples = samples[lo:hi]
            for s0 in range(0, len(samples), batch):
                ids = samples[s0:s0 + batch]
                rgb = torch.from_numpy(np.stack([io.read_color(data_dir, i) for i in ids])).to(device)
                depth = torch.from_numpy(np.stack([io.read_depth(data_dir, i) for i in ids])).to(device)
                r2c = np.stack([io.robot2cam(io.read_meta(data_dir, i)) for i in ids])
                bs = None
                if d != "extra":
                    bs = torch.from_numpy(np.stack([io.read_label(label_path, i, "pred") for i in ids])).to(device)
                labels, save, st = relabel_frames(model, rgb, depth, r2c, reference_point, class_id, bs, is_extra=(d == "extra"))
                for k in ("bs_copied", "no_depth_overlap", "not_in_center"):
                    stats[k] += st[k]
                labels_h = labels.cpu().numpy()
                for j, sid in enumerate(ids):
                    new_png = os.path.join(label_path, "{}.new_pred.label.png".format(sid))
                    if save[j]:
                        stats["n_extra_samples" if d == "extra" else "n_samples"] += 1
                        io.write_label(label_path, sid, "new_pred", labels_h[j])
                    else:                                               # reference :206-214: drop stale outputs
                        for stale in (new_png, os.path.join(label_path, "{}.meta.json".format(sid))):
                            if os.path.exists(stale):
                                os.remove(stale)
        if dist_on:
            torch.cuda.synchronize()
            dist.barrier()                      # every rank's label PNGs are on the shared file system
        times["seg"].append(time.time() - t0)
        t0 = time.time()
        load_point_cloud(cls, os.path.join(root, "pc_reconstruction/data"), root, reference_point=reference_point, mode=mode,
                         n_viewpoints=n_viewpoints, min_friends=20, min_dist=5, nb_neighbors=20, threshold=10, voxel_size=2,
                         voxel_size_out=5, l_arrow=75, global_regression=False, icp_point2point=True, icp_point2plane=False,
                         rng=None if rng_seed is None else np.random.default_rng(rng_seed + class_id), dist=dist if dist_on else None)
        times["pc"].append(time.time() - t0)
        t0 = time.time()
        if rank == 0:
            create_pose_label(root, cls, False, True, False, plot=False, view_label=False, with_extra=get_extra_labels)
        if dist_on:
            dist.barrier()
        times["pose"].append(time.time() - t0)
        print("class {}: seg {:.2f} s, pc {:.2f} s, pose {:.2f} s; stats {}".format(cls, times["seg"][-1], times["pc"][-1],
                                                                                   times["pose"][-1], stats))
    if dist_on:
        keys = sorted(stats)
        t = torch.tensor([stats[k] for k in keys], dtype=torch.int64, device=sharding._dist_device(dist))
        dist.all_reduce(t)
        stats = {k: int(v) for k, v in zip(keys, t.cpu().tolist())}
    return stats, times
