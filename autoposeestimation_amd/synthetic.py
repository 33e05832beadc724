"""Deterministic synthetic weights and RGB-D frames (no datasets, no checkpoints, no network).

The reference ships no weights ("Checkpoint / resume"), so parity tests, the golden
generator (tools/gen_golden.py) and bench.py all need weights that

  * carry exactly the reference's state-dict key names and shapes
    (PoseNet: 77 tensors, PoseRefineNet: 24 tensors -- DenseFusion/lib/network.py:70-132,170-206,
    pspnet.py:40-62, extractors.py:78-112), so they load into the reference modules with strict=True
    (that is how gen_golden.py proves the key layout), and
  * can be regenerated bit-identically anywhere from (name, seed) alone -- an 86 MB PoseNet state
    dict cannot be committed as a fixture.  numpy's PCG64 `default_rng` stream is stable across
    platforms and numpy versions, torch's generator is not guaranteed to be.

Scales are chosen so that activations stay O(1..100) from the raw 0-255 ImageNet-"normalised" crop
(pipeline/utils.py:559-560 feeds (rgb-0.485)/0.229, i.e. values up to ~1100) down to O(0.1) pose
outputs, the way a trained network's would; He-style fan-in scaling elsewhere.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

_BLOCKS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}


def _rng(seed, key):
    return np.random.default_rng([int(seed), zlib.crc32(key.encode())])


def _normal(seed, key, shape, std):
    return torch.from_numpy((_rng(seed, key).standard_normal(shape, dtype=np.float32) * np.float32(std)))


def _uniform(seed, key, shape, bound):
    return torch.from_numpy(((_rng(seed, key).random(shape, dtype=np.float32) * 2 - 1) * np.float32(bound)))


def pspnet_spec(backend="resnet18", n_classes=21, in_channels=3):
    """[(key, shape)] of DenseFusion/lib/pspnet.py:PSPNet with a BasicBlock encoder, in state_dict order.
    in_channels != 3 only for the segmentor stand-ins (the background-subtraction network reads 7 channels)."""
    spec = [("feats.conv1.weight", (64, in_channels, 7, 7))]
    inplanes = 64
    for li, (planes, nblk, stride) in enumerate(zip((64, 128, 256, 512), _BLOCKS[backend], (1, 2, 1, 1)), start=1):
        for b in range(nblk):
            cin = inplanes if b == 0 else planes
            spec.append((f"feats.layer{li}.{b}.conv1.weight", (planes, cin, 3, 3)))
            spec.append((f"feats.layer{li}.{b}.conv2.weight", (planes, planes, 3, 3)))
            if b == 0 and (stride != 1 or inplanes != planes):
                spec.append((f"feats.layer{li}.{b}.downsample.0.weight", (planes, inplanes, 1, 1)))
        inplanes = planes
    for s in range(4):
        spec.append((f"psp.stages.{s}.1.weight", (512, 512, 1, 1)))
    spec += [("psp.bottleneck.weight", (1024, 2560, 1, 1)), ("psp.bottleneck.bias", (1024,))]
    for name, cin, cout in (("up_1", 1024, 256), ("up_2", 256, 64), ("up_3", 64, 64)):
        spec += [(f"{name}.conv.1.weight", (cout, cin, 3, 3)), (f"{name}.conv.1.bias", (cout,)),
                 (f"{name}.conv.2.weight", (1,))]
    spec += [("final.0.weight", (32, 64, 1, 1)), ("final.0.bias", (32,))]
    spec += [("classifier.0.weight", (256, 256)), ("classifier.0.bias", (256,)),
             ("classifier.2.weight", (n_classes, 256)), ("classifier.2.bias", (n_classes,))]
    return spec


def _fill(spec, seed, prefix="", overrides=None):
    """He-normal (fan-in) weights, small uniform biases, PReLU slope 0.25; `overrides[key]` scales std."""
    overrides = overrides or {}
    sd = OrderedDict()
    for key, shape in spec:
        full = prefix + key
        if len(shape) == 1 and key.endswith(".conv.2.weight"):  # nn.PReLU single slope (pspnet.py:33)
            sd[full] = torch.full(shape, 0.25, dtype=torch.float32)
            continue
        if key.endswith(".bias"):
            sd[full] = _uniform(seed, full, shape, 0.05 * overrides.get(key, 1.0))
            continue
        fan_in = int(np.prod(shape[1:]))
        std = (2.0 / fan_in) ** 0.5 * overrides.get(key, 1.0)
        sd[full] = _normal(seed, full, shape, std)
    return sd


def pspnet_state_dict(backend="resnet18", seed=0, prefix="", n_classes=21, stem_gain=1.0 / 256.0, in_channels=3):
    """stem_gain: the PoseNet crop encoder sees raw 0-255-scale pixels (pipeline/utils.py:559-560), so its stem is scaled
    by 1/256 to bring them to O(1) like a trained first layer would; a SEGMENTOR sees ToTensor'd [0,1] pixels
    (pipeline/utils.py:421-424) -> use stem_gain=1.0 there."""
    spec = pspnet_spec(backend, n_classes, in_channels)
    over = {"feats.conv1.weight": stem_gain, "final.0.weight": 0.35}
    # residual branches at half gain so 8..16 un-normalised blocks do not blow the variance up
    over.update({k: 0.5 for k, _ in spec if k.startswith("feats.layer") and k.endswith("conv2.weight")})
    return _fill(spec, seed, prefix, over)


def _pointnet_spec(refine):
    c5 = 384 if refine else 256
    return [("feat.conv1.weight", (64, 3, 1)), ("feat.conv1.bias", (64,)),
            ("feat.conv2.weight", (128, 64, 1)), ("feat.conv2.bias", (128,)),
            ("feat.e_conv1.weight", (64, 32, 1)), ("feat.e_conv1.bias", (64,)),
            ("feat.e_conv2.weight", (128, 64, 1)), ("feat.e_conv2.bias", (128,)),
            ("feat.conv5.weight", (512, c5, 1)), ("feat.conv5.bias", (512,)),
            ("feat.conv6.weight", (1024, 512, 1)), ("feat.conv6.bias", (1024,))]


def posenet_state_dict(num_obj, seed=0):
    """77 tensors with the key names of DenseFusion/lib/network.py:PoseNet (incl. `cnn.model.module.`)."""
    sd = pspnet_state_dict("resnet18", seed, prefix="cnn.model.module.")
    spec = _pointnet_spec(False)
    for i, (cin, cout) in enumerate(((1408, 640), (640, 256), (256, 128)), start=1):
        for h in "rtc":
            spec += [(f"conv{i}_{h}.weight", (cout, cin, 1)), (f"conv{i}_{h}.bias", (cout,))]
    for h, m in (("r", 4), ("t", 3), ("c", 1)):
        spec += [(f"conv4_{h}.weight", (num_obj * m, 128, 1)), (f"conv4_{h}.bias", (num_obj * m,))]
    # point coordinates are metres (O(0.5)): lift conv1 so geometry and colour features have equal weight;
    # keep translation offsets ~cm and confidence logits O(1)
    over = {"feat.conv1.weight": 2.0, "conv4_t.weight": 0.02, "conv4_t.bias": 0.2, "conv4_c.weight": 0.5}
    sd.update(_fill(spec, seed, "", over))
    return sd


def refiner_state_dict(num_obj, seed=0):
    """24 tensors with the key names of DenseFusion/lib/network.py:PoseRefineNet."""
    spec = _pointnet_spec(True)
    for i, (cin, cout) in enumerate(((1024, 512), (512, 128)), start=1):
        for h in "rt":
            spec += [(f"conv{i}_{h}.weight", (cout, cin)), (f"conv{i}_{h}.bias", (cout,))]
    for h, m in (("r", 4), ("t", 3)):
        spec += [(f"conv3_{h}.weight", (num_obj * m, 128)), (f"conv3_{h}.bias", (num_obj * m,))]
    over = {"feat.conv1.weight": 2.0, "conv3_t.weight": 0.02, "conv3_t.bias": 0.2}
    sd = _fill(spec, seed + 1, "", over)
    # a refiner predicts a *residual* rotation: bias the quaternion head towards identity (w dominant)
    b = sd["conv3_r.bias"].view(num_obj, 4)
    b[:, 0] += 1.0
    return sd


# ----------------------------------------------------------------------------------------------
# synthetic RGB-D frames
# ----------------------------------------------------------------------------------------------
REALSENSE_META = {"intr": {"fx": 615.0, "fy": 615.0, "ppx": 320.0, "ppy": 240.0}, "depth_scale": 0.001}
YCB_META = {"intr": {"fx": 1066.778, "fy": 1067.487, "ppx": 312.9869, "ppy": 241.3109}, "depth_scale": 1.0 / 10000}

CLASS_COLOURS = np.array([[230, 40, 40], [40, 200, 60], [50, 80, 230], [230, 210, 40], [200, 50, 200],
                          [40, 210, 210], [250, 140, 30], [130, 60, 20], [120, 120, 250], [20, 120, 70],
                          [250, 160, 200], [90, 90, 90]], dtype=np.uint8)


def synthetic_frame(frame_id, cls=1, box=(150, 250), size=(150, 150), h=480, w=640):
    """One 640x480 RGB u8 + depth u16 frame with a single painted box "object" of class `cls` (1-based).

    Returns (rgb[h,w,3] u8, depth[h,w] u16, label[h,w] u8 with {0, cls}).  The default 150x150 box
    makes `get_bbox` (myDatasetAugmented/dataset.py:342-380) round the crop up to 160x160.
    Background: uniform noise in [96,160); depth: plane at 600 mm, object 80 mm closer with a
    smooth bump, 2 % of all pixels invalid (0).
    """
    return synthetic_frame_multi(frame_id, [(cls, box, size)], h, w)


def synthetic_frame_multi(frame_id, objects, h=480, w=640):
    """synthetic_frame with several painted boxes: objects = [(cls, (r0, c0), (rh, rw)), ...] (non-overlapping, distinct classes -- the
    live path keeps one detection per class and frame, pipeline/utils.py:444-470).  One object gives exactly synthetic_frame's arrays
    (same generator, same draw order: background, per object its colour jitter, then the invalid-depth mask)."""
    rng = np.random.default_rng(int(frame_id))
    rgb = rng.integers(96, 160, size=(h, w, 3), dtype=np.uint8)
    label = np.zeros((h, w), np.uint8)
    depth = np.full((h, w), 600, np.uint16)
    for cls, (r0, c0), (rh, rw) in objects:
        label[r0:r0 + rh, c0:c0 + rw] = cls
        col = CLASS_COLOURS[(cls - 1) % len(CLASS_COLOURS)].astype(np.int16)
        jitter = rng.integers(-12, 13, size=(rh, rw, 3), dtype=np.int16)
        rgb[r0:r0 + rh, c0:c0 + rw] = np.clip(col + jitter, 0, 255).astype(np.uint8)
        yy, xx = np.mgrid[0:rh, 0:rw]
        bump = 40.0 * np.exp(-(((yy - rh / 2) / (rh / 3)) ** 2 + ((xx - rw / 2) / (rw / 3)) ** 2))
        depth[r0:r0 + rh, c0:c0 + rw] = (520 - bump).astype(np.uint16)
    depth[rng.random((h, w)) < 0.02] = 0
    return rgb, depth, label


# SURVEY.md 8d's crop sweep {80^2, 120x160, 160^2, 240^2, 320x400}: painted sizes that get_bbox (myDatasetAugmented/dataset.py:342-380) rounds
# up to those crops
MIXED_SIZES = ((70, 70), (110, 150), (150, 150), (230, 230), (310, 390))


def mixed_frame(frame_id, h=480, w=640, margin=24):
    """A frame of the `bench.py --mixed` sweep: 1-3 objects of distinct classes (1..3), sizes drawn from MIXED_SIZES, placed at random
    without overlap (a draw that does not fit after 40 tries is dropped, so a frame holds at least its first object).  Seeded by
    frame_id alone.  Returns (rgb, depth, label, objects)."""
    rng = np.random.default_rng([4242, int(frame_id)])
    n_obj = int(rng.integers(1, 4))
    classes = rng.permutation(3)[:n_obj] + 1
    placed = []
    for cls in classes.tolist():
        for _ in range(40):
            rh, rw = MIXED_SIZES[int(rng.integers(0, len(MIXED_SIZES)))]
            r0 = int(rng.integers(margin, h - margin - rh + 1)) if h - 2 * margin - rh >= 0 else -1
            c0 = int(rng.integers(margin, w - margin - rw + 1)) if w - 2 * margin - rw >= 0 else -1
            if r0 < 0 or c0 < 0:
                continue
            if all(r0 + rh + margin <= pr or pr + ph + margin <= r0 or c0 + rw + margin <= pc or pc + pw + margin <= c0
                   for _, (pr, pc), (ph, pw) in placed):
                placed.append((cls, (r0, c0), (rh, rw)))
                break
    rgb, depth, label = synthetic_frame_multi(900000 + int(frame_id), placed, h, w)
    return rgb, depth, label, placed


def model_cloud(cls, m=1000, seed=1234):
    """`m` model points uniform in a 0.1 m cube (metres), per class."""
    rng = np.random.default_rng([seed, int(cls)])
    return ((rng.random((m, 3), dtype=np.float32) - 0.5) * np.float32(0.1)).astype(np.float32)


def fit_final_layer(feats, labels, n_out, margin=8.0, ridge=1e-2):
    """Least-squares "training" of a segmentor's final 1x1 conv on frozen random features, so that synthetic frames
    segment into their painted object with real logit margins (random weights alone give a random label map).
    feats[P,64] float tensor (any device), labels[P] int64 in [0,n_out) -> (W[n_out,64], b[n_out]) float32 on CPU.
    Targets: +margin for the pixel's class, -margin for every other class."""
    f = feats.double()
    p = f.shape[0]
    a = torch.cat([f, torch.ones(p, 1, dtype=torch.float64, device=f.device)], 1)
    t = torch.full((p, n_out), -float(margin), dtype=torch.float64, device=f.device)
    t[torch.arange(p, device=f.device), labels.to(f.device)] = float(margin)
    ata = a.t() @ a + ridge * p * torch.eye(a.shape[1], dtype=torch.float64, device=f.device)
    sol = torch.linalg.solve(ata, a.t() @ t)          # [65, n_out]
    return sol[:64].t().float().cpu().contiguous(), sol[64].float().cpu().contiguous()


# ---- label path (BASELINE configs[4]): synthetic multi-view depth renders of a known object --------------------------------------
LABEL_INTR = {"fx": 615.0, "fy": 615.0, "ppx": 320.0, "ppy": 240.0}
LABEL_CENTRE = np.array([400.0, -20.0, 150.0])


def rigid(rx, ry, rz, t):
    """4x4 from Euler ZYX angles (R = Rz Ry Rx) and a translation"""
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    R = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]]) @ np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]]) @ np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


def bumpy_sphere(n, seed, radius=60.0, centre=LABEL_CENTRE):
    """a sphere with bumps (SURVEY.md 8d, config 5's known object), n surface samples in the robot frame (mm)"""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    r = radius * (1 + 0.15 * np.sin(3 * v[:, 0]) * np.cos(4 * v[:, 1]) + 0.1 * np.sin(5 * v[:, 2]))
    return v * r[:, None] + np.asarray(centre)


def render_depth(points, cam2robot, intr=LABEL_INTR, h=480, w=640):
    """z-buffer render of a robot-frame cloud into a uint16 depth image (mm) for a pin-hole camera at `cam2robot`"""
    Tinv = np.linalg.inv(cam2robot)
    pc = points @ Tinv[:3, :3].T + Tinv[:3, 3]
    pc = pc[pc[:, 2] > 50]
    u = np.round(pc[:, 0] * intr["fx"] / pc[:, 2] + intr["ppx"]).astype(int)
    v = np.round(pc[:, 1] * intr["fy"] / pc[:, 2] + intr["ppy"]).astype(int)
    ok = (u >= 0) & (u < w) & (v >= 0) & (v < h)
    depth = np.zeros((h, w), np.float64)
    order = np.argsort(-pc[ok, 2])
    depth[v[ok][order], u[ok][order]] = np.round(pc[ok, 2][order])
    return depth.astype(np.uint16)


def label_views(n_views, seed=0, cloud=None, distance=500.0, poses=None):
    """`n_views` (label u8, depth u16, robot2cam 4x4) of the bumpy sphere.  `poses` = robot2cam matrices to render from (the reference's
    own capture path: `capture_path()`); without them the cameras sit on a cap around the object (looking at it from `distance` mm,
    tilted by up to +-1 rad about y and +-0.5 rad about x, drawn from `seed`)."""
    cloud = bumpy_sphere(300000, 21) if cloud is None else cloud
    rng = np.random.default_rng(seed)
    views = []
    for i in range(n_views):
        if poses is not None:
            cam = np.asarray(poses[i % len(poses)], dtype=np.float64)
        else:
            ang_y, ang_x = rng.uniform(-1.0, 1.0), rng.uniform(-0.5, 0.5)
            cam = rigid(np.pi, 0.0, 0.0, tuple(LABEL_CENTRE + [0, 0, distance]))
            cam = rigid(0, 0, 0, tuple(LABEL_CENTRE)) @ rigid(ang_x, ang_y, 0.0, (0, 0, 0)) @ rigid(0, 0, 0, tuple(-LABEL_CENTRE)) @ cam
        depth = render_depth(cloud, cam)
        views.append(((depth != 0).astype(np.uint8) * 255, depth, cam))
    return views


def capture_path():
    """(robot2cam[164, 4, 4], focus[3]): the camera poses of the reference's capture run (robot_controller/robot_path/
    viewpointsPath2.json through the hand-eye calibration) and the point its optical axes meet in, from the committed fixture
    tests/golden/viewpoints_path2.npz (tools/gen_golden_viewpoints.py); None when the fixture is absent."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "viewpoints_path2.npz")
    if not os.path.exists(path):
        return None
    d = np.load(path)
    return d["robot2cam"], d["focus"]


def pose_dataset_tree(root, data_set_name="synth", n_view=5, n_extra=3, seed=3):
    """A small pose-estimation data set in the reference's on-disk layout (data_generation/getData.py:177-221,
    label_generator/create_labels.py:422-429, create_pointcloud.py:373-376, make_train_and_test_dataset): two classes (the second
    symmetric), `n_view` view-point samples each under `<cls>/foreground`, `n_extra` extra samples under `<cls>/extra`, label PNGs in
    all three label modes, pose-label JSONs, `.xyz` model clouds (1200 points, mm) and the list files.  Deterministic in `seed`;
    shared by tools/gen_golden_dataset.py (which runs the REFERENCE's PoseDataset on it) and tests/test_pose_dataset_golden.py."""
    import json
    import os
    from PIL import Image
    rng = np.random.default_rng(seed)
    classes = ["objA", "objB"]
    ds = os.path.join(root, "label_generator/data_sets/pose_estimation", data_set_name)
    os.makedirs(ds, exist_ok=True)
    train, test, extra = [], [], []
    for ci, cls in enumerate(classes):
        cloud = (rng.uniform(-40, 40, (1200, 3)) * [1.0, 0.7, 0.5]).round(3)
        os.makedirs(os.path.join(root, "pc_reconstruction/data", cls), exist_ok=True)
        with open(os.path.join(root, "pc_reconstruction/data", cls, cls + ".xyz"), "w") as f:
            for item in cloud:
                f.write("%s\n" % item)                          # numpy's repr of the point, as create_pointcloud.py:373-376 writes it
        for sub, n in (("foreground", n_view), ("extra", n_extra)):
            ddir = os.path.join(root, "data_generation/data", cls, sub)
            ldir = os.path.join(root, "label_generator/data", cls, sub)
            os.makedirs(ddir, exist_ok=True)
            os.makedirs(ldir, exist_ok=True)
            for i in range(n):
                sid = "%06d" % i
                h, w = 480, 640
                r0, c0 = int(rng.integers(60, 250)), int(rng.integers(80, 380))
                hh, ww = int(rng.integers(70, 150)), int(rng.integers(70, 170))
                yy, xx = np.mgrid[0:h, 0:w]
                inside = (((yy - r0 - hh / 2) / (hh / 2)) ** 2 + ((xx - c0 - ww / 2) / (ww / 2)) ** 2) < 1
                rgb = (rng.integers(0, 60, (h, w, 3)) + 90).astype(np.uint8)
                rgb[inside] = (CLASS_COLOURS[ci] * 0.8).astype(np.uint8) + rng.integers(0, 40, (int(inside.sum()), 3)).astype(np.uint8)
                depth = np.full((h, w), 620, np.uint16) + (xx // 40).astype(np.uint16)
                depth[inside] = (560 + 25 * np.sin(xx[inside] / 17.0) + 15 * np.cos(yy[inside] / 13.0)).astype(np.uint16)
                depth[rng.random((h, w)) < 0.03] = 0
                Image.fromarray(rgb).save(os.path.join(ddir, sid + ".color.png"))
                Image.fromarray(depth).save(os.path.join(ddir, sid + ".depth.png"))
                meta = {"intr": dict(REALSENSE_META["intr"]), "depth_scale": 0.001, "symmetric": bool(ci == 1), "view_point_id": i % n_view,
                        "robot2endEff_tf": np.eye(4).flatten().tolist(), "hand_eye_calibration": np.eye(4).flatten().tolist(),
                        "object_pose": np.eye(4).flatten().tolist()}
                with open(os.path.join(ddir, sid + ".meta.json"), "w") as f:
                    json.dump(meta, f)
                lab = inside.astype(np.uint8) * 255
                for mode in ("gen", "pred", "new_pred"):
                    Image.fromarray(lab).save(os.path.join(ldir, "%s.%s.label.png" % (sid, mode)))
                cam2robot = rigid(rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-3, 3), tuple(rng.uniform(-50, 50, 3)))
                robot2object = rigid(rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1), tuple(rng.uniform(-30, 30, 2)) + (float(rng.uniform(500, 650)),))
                with open(os.path.join(ldir, sid + ".meta.json"), "w") as f:
                    json.dump({"cls_name": cls, "cam2robot": cam2robot.flatten().tolist(), "robot2object": robot2object.flatten().tolist()}, f)
                rel = "%s/%s/%s" % (cls, sub, sid)
                if sub == "extra":
                    extra.append(rel)
                elif i == n - 1:
                    test.append(rel)
                else:
                    train.append(rel)
    with open(os.path.join(ds, "classes.txt"), "w") as f:
        f.write("".join(c + "\n" for c in classes))
    for name, items in (("train_data_list.txt", train), ("test_data_list.txt", test), ("extra_train_data_list.txt", extra)):
        with open(os.path.join(ds, name), "w") as f:
            f.write("".join(x + "\n" for x in items))
    return classes, train, test, extra
