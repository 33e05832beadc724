"""TEST INFRASTRUCTURE -- CPU oracle (plain PyTorch fp32 / numpy) for the seg -> DenseFusion slice.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; the
product path (autoposeestimation_amd) never does and fails loudly without its HIP library.

Every function restates the arithmetic of one reference function (cited file:line, relative to
KochPJ/AutoPoseEstimation) in functional form on a flat state_dict -- no nn.Module copies.  Parity is
pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which tools/gen_golden.py produced
by importing and running the reference's own modules in the build container.

Unpinned (third-party arithmetic absent from the reference tree):
`cv2.connectedComponents` label numbering -- restated as raster-order 8-connectivity labelling;
the smp segmentor -- replaced by the in-repo PSPNet behind `segmentation.utils.get_model('PsPNet')`.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

_LAYERS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}


# ----------------------------------------------------------------------------------------------
# PSPNet + dilated BN-free ResNet  (DenseFusion/lib/pspnet.py:7-77, extractors.py:18-124)
# ----------------------------------------------------------------------------------------------
def _basic_block(sd, p, x, stride, dilation, has_down):
    # extractors.py:29-43 -- conv3x3(stride, dilation) -> relu -> conv3x3(dilation) -> += residual -> relu
    out = F.conv2d(x, sd[p + "conv1.weight"], None, stride, dilation, dilation)
    out = F.relu(out)
    out = F.conv2d(out, sd[p + "conv2.weight"], None, 1, dilation, dilation)
    res = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride) if has_down else x
    return F.relu(out + res)


def resnet_features(sd, p, x, backend="resnet18", taps=None):
    # extractors.py:114-124; _make_layer (:99-112) forwards `dilation` only to blocks 1.. of a layer
    x = F.relu(F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3))
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["stem"] = x
    inplanes = 64
    for li, (planes, nblk, stride, dil) in enumerate(
            zip((64, 128, 256, 512), _LAYERS[backend], (1, 2, 1, 1), (1, 1, 2, 4)), start=1):
        for b in range(nblk):
            first = b == 0
            x = _basic_block(sd, f"{p}layer{li}.{b}.", x, stride if first else 1, 1 if first else dil,
                             first and (stride != 1 or inplanes != planes))
        inplanes = planes
        if taps is not None:
            taps[f"layer{li}"] = x
    return x


def psp_module(sd, p, f):
    # pspnet.py:20-24 -- AdaptiveAvgPool(s) -> 1x1 conv (no bias) -> bilinear up (align_corners=False) ; cat ; 1x1 ; relu
    h, w = f.shape[2], f.shape[3]
    priors = []
    for i, s in enumerate((1, 2, 3, 6)):
        y = F.conv2d(F.adaptive_avg_pool2d(f, (s, s)), sd[f"{p}stages.{i}.1.weight"])
        priors.append(F.interpolate(y, size=(h, w), mode="bilinear", align_corners=False))
    priors.append(f)
    return F.relu(F.conv2d(torch.cat(priors, 1), sd[p + "bottleneck.weight"], sd[p + "bottleneck.bias"]))


def psp_upsample(sd, p, x):
    # pspnet.py:27-37 -- bilinear x2 align_corners=True -> 3x3 conv pad 1 (+bias) -> PReLU (1 slope)
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.conv2d(x, sd[p + "conv.1.weight"], sd[p + "conv.1.bias"], 1, 1)
    return F.prelu(x, sd[p + "conv.2.weight"])


def pspnet_forward(sd, x, prefix="", backend="resnet18", taps=None, logits_only=False, drop=None):
    """pspnet.py:64-77.  Eval mode (Dropout2d = identity) unless `drop` gives the train-mode channel multipliers
    {'drop_1': [B,1024], 'drop_2a': [B,256], 'drop_2b': [B,64]} (0 or 1/(1-p), what nn.Dropout2d(p) multiplies by at :48,50).
    Returns log_softmax(final conv) [B,32,H,W]."""
    f = resnet_features(sd, prefix + "feats.", x, backend, taps)
    p = psp_module(sd, prefix + "psp.", f)
    if taps is not None:
        taps["feats"], taps["psp"] = f, p
    if drop is not None:
        p = p * drop["drop_1"][:, :, None, None]
    for name, dkey in (("up_1", "drop_2a"), ("up_2", "drop_2b"), ("up_3", None)):
        p = psp_upsample(sd, f"{prefix}{name}.", p)
        if drop is not None and dkey is not None:
            p = p * drop[dkey][:, :, None, None]
        if taps is not None:
            taps[name] = p
    logits = F.conv2d(p, sd[prefix + "final.0.weight"], sd[prefix + "final.0.bias"])
    if logits_only:
        return logits
    return F.log_softmax(logits, dim=1)  # nn.LogSoftmax() implicit dim -> 1 for 4-D input


# ----------------------------------------------------------------------------------------------
# PoseNet / PoseRefineNet  (DenseFusion/lib/network.py:39-206)
# ----------------------------------------------------------------------------------------------
def _c1(sd, name, x, relu=True):
    y = F.conv1d(x, sd[name + ".weight"], sd[name + ".bias"])
    return F.relu(y) if relu else y


def posenet_feat(sd, x, emb):
    # network.py:53-68
    x1 = _c1(sd, "feat.conv1", x)
    e1 = _c1(sd, "feat.e_conv1", emb)
    pf1 = torch.cat((x1, e1), 1)
    x2 = _c1(sd, "feat.conv2", x1)
    e2 = _c1(sd, "feat.e_conv2", e1)
    pf2 = torch.cat((x2, e2), 1)
    x5 = _c1(sd, "feat.conv5", pf2)
    x6 = _c1(sd, "feat.conv6", x5)
    n = x.shape[2]
    ap = F.avg_pool1d(x6, n).view(-1, 1024, 1).repeat(1, 1, n)
    return torch.cat([pf1, pf2, ap], 1)


def posenet_forward(sd, img, x, choose, obj, num_obj, taps=None, drop=None):
    """network.py:95-132.  img[1,3,H,W], x[1,N,3], choose[1,1,N] i64, obj[1,1] i64.  drop: train-mode Dropout2d multipliers."""
    out_img = pspnet_forward(sd, img, "cnn.model.module.", "resnet18", taps, drop=drop)
    bs, di = out_img.shape[:2]
    n = x.shape[1]
    emb = torch.gather(out_img.view(bs, di, -1), 2, choose.repeat(1, di, 1)).contiguous()
    ap_x = posenet_feat(sd, x.transpose(2, 1).contiguous(), emb)
    if taps is not None:
        taps["final"], taps["posenetfeat"] = out_img, ap_x
    outs = {}
    for h, m in (("r", 4), ("t", 3), ("c", 1)):
        y = _c1(sd, f"conv1_{h}", ap_x)
        y = _c1(sd, f"conv2_{h}", y)
        y = _c1(sd, f"conv3_{h}", y)
        y = _c1(sd, f"conv4_{h}", y, relu=False)
        if h == "c":
            y = torch.sigmoid(y)
        y = y.view(bs, num_obj, m, n)
        outs[h] = torch.index_select(y[0], 0, obj[0]).transpose(2, 1).contiguous()
    return outs["r"], outs["t"], outs["c"], emb


def refiner_feat(sd, x, emb):
    # network.py:151-168
    x1 = _c1(sd, "feat.conv1", x)
    e1 = _c1(sd, "feat.e_conv1", emb)
    x2 = _c1(sd, "feat.conv2", x1)
    e2 = _c1(sd, "feat.e_conv2", e1)
    pf3 = torch.cat([x1, e1, x2, e2], 1)
    x5 = _c1(sd, "feat.conv5", pf3)
    x6 = _c1(sd, "feat.conv6", x5)
    return F.avg_pool1d(x6, x.shape[2]).view(-1, 1024)


def refiner_forward(sd, x, emb, obj, num_obj):
    """network.py:187-206.  x[1,N,3], emb[1,32,N], obj[1,1] -> r[1,4], t[1,3]."""
    ap = refiner_feat(sd, x.transpose(2, 1).contiguous(), emb)
    outs = []
    for h, m in (("r", 4), ("t", 3)):
        y = F.relu(F.linear(ap, sd[f"conv1_{h}.weight"], sd[f"conv1_{h}.bias"]))
        y = F.relu(F.linear(y, sd[f"conv2_{h}.weight"], sd[f"conv2_{h}.bias"]))
        y = F.linear(y, sd[f"conv3_{h}.weight"], sd[f"conv3_{h}.bias"]).view(1, num_obj, m)
        outs.append(torch.index_select(y[0], 0, obj[0]))
    return outs[0], outs[1]


# ----------------------------------------------------------------------------------------------
# pose extraction / composition  (DenseFusion/tools/utils.py:7-86, lib/transformations.py:1254-1363)
# ----------------------------------------------------------------------------------------------
def quat_to_base(q):
    """9 row-major terms of loss.py:20-28 == tools/utils.py:46-67 for q[...,4] = (w,x,y,z) already normalised."""
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return torch.stack((1.0 - 2.0 * (y ** 2 + z ** 2), 2.0 * x * y - 2.0 * w * z, 2.0 * w * y + 2.0 * x * z,
                        2.0 * x * y + 2.0 * z * w, 1.0 - 2.0 * (x ** 2 + z ** 2), -2.0 * w * x + 2.0 * y * z,
                        -2.0 * w * y + 2.0 * x * z, 2.0 * w * x + 2.0 * y * z, 1.0 - 2.0 * (x ** 2 + y ** 2)),
                       dim=-1).view(*q.shape[:-1], 3, 3)


def get_new_points(pred_r, pred_t, pred_c, points):
    # tools/utils.py:43-86
    bs, num_p, _ = pred_c.shape
    q = pred_r / torch.norm(pred_r, dim=2).view(bs, num_p, 1)
    base = quat_to_base(q).view(bs * num_p, 3, 3)
    which = torch.max(pred_c.view(bs, num_p), 1)[1][0]
    t = pred_t.view(bs * num_p, 1, 3)[which] + points.view(bs * num_p, 1, 3)[which]
    return torch.bmm(points.view(1, bs * num_p, 3) - t.view(1, 1, 3), base[which].view(1, 3, 3)).contiguous()


def estimator_prediction(pred_r, pred_t, pred_c, num_points, bs, cloud):
    # tools/utils.py:7-18
    q = pred_r / torch.norm(pred_r, dim=2).view(1, num_points, 1)
    which = torch.max(pred_c.view(bs, num_points), 1)[1][0]
    my_r = q[0][which].view(-1).numpy()
    my_t = (cloud.view(bs * num_points, 1, 3) + pred_t.view(bs * num_points, 1, 3))[which].view(-1).numpy()
    return np.append(my_r, my_t), my_r, my_t


def quaternion_matrix(q):
    # transformations.py:1254-1278
    q = np.array(q, dtype=np.float64, copy=True)
    n = np.dot(q, q)
    if n < np.finfo(float).eps * 4.0:
        return np.identity(4)
    q *= math.sqrt(2.0 / n)
    q = np.outer(q, q)
    return np.array([[1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0], 0.0],
                     [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0], 0.0],
                     [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2], 0.0],
                     [0.0, 0.0, 0.0, 1.0]])


def quaternion_from_matrix_precise(M):
    # transformations.py:1320-1341,1361-1363 (isprecise=True branch)
    M = np.asarray(M, dtype=np.float64)[:4, :4]
    q = np.empty((4,))
    t = np.trace(M)
    if t > M[3, 3]:
        q[0] = t
        q[3] = M[1, 0] - M[0, 1]
        q[2] = M[0, 2] - M[2, 0]
        q[1] = M[2, 1] - M[1, 2]
    else:
        i, j, k = 0, 1, 2
        if M[1, 1] > M[0, 0]:
            i, j, k = 1, 2, 0
        if M[2, 2] > M[i, i]:
            i, j, k = 2, 0, 1
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        q[i] = t
        q[j] = M[i, j] + M[j, i]
        q[k] = M[k, i] + M[i, k]
        q[3] = M[k, j] - M[j, k]
        q = q[[3, 0, 1, 2]]
    q *= 0.5 / math.sqrt(t * M[3, 3])
    if q[0] < 0.0:
        np.negative(q, q)
    return q


def refined_prediction(pred_r, pred_t, my_r, my_t):
    # tools/utils.py:20-40 -- float64 4x4 compose on the host
    m1 = quaternion_matrix(my_r)
    m1[0:3, 3] = my_t
    q2 = pred_r.view(1, 1, -1)
    q2 = (q2 / torch.norm(q2, dim=2).view(1, 1, 1)).view(-1).numpy()
    m2 = quaternion_matrix(q2)
    m2[0:3, 3] = pred_t.view(-1).numpy()
    mf = np.dot(m1, m2)
    rf = mf.copy()
    rf[0:3, 3] = 0
    r = quaternion_from_matrix_precise(rf)
    t = np.array([mf[0][3], mf[1][3], mf[2][3]])
    return np.append(r, t), r, t


# ----------------------------------------------------------------------------------------------
# ADD / ADD-S  (DenseFusion/lib/loss.py:12-73, loss_refiner.py:12-64, tools/eval_linemod.py:118-130)
# ----------------------------------------------------------------------------------------------
def knn1(ref, query):
    """1-based 1-NN indices with the tie rule of knn_cpu.cpp:21-43 (lowest ref index wins).
    torch restatement for large cases; bit-exact C restatement lives in oracle/knn_oracle.c."""
    d = ((ref[0].t()[None, :, :] - query[0].t()[:, None, :]) ** 2)
    d = (d[..., 0] + d[..., 1]) + d[..., 2]
    # argmin returns the first minimum on CPU
    return (torch.argmin(d, dim=1) + 1).view(1, 1, -1)


def loss_forward(pred_r, pred_t, pred_c, target, model_points, idx, points, w, refine, num_pt_mesh, sym_list,
                 knn=knn1):
    # loss.py:12-73
    bs, num_p, _ = pred_c.shape
    q = pred_r / torch.norm(pred_r, dim=2).view(bs, num_p, 1)
    ori_base = quat_to_base(q).view(bs * num_p, 3, 3)
    base = ori_base.transpose(2, 1).contiguous()
    mp = model_points.view(bs, 1, num_pt_mesh, 3).repeat(1, num_p, 1, 1).view(bs * num_p, num_pt_mesh, 3)
    tg = target.view(bs, 1, num_pt_mesh, 3).repeat(1, num_p, 1, 1).view(bs * num_p, num_pt_mesh, 3)
    ori_target = tg
    pt = pred_t.contiguous().view(bs * num_p, 1, 3)
    pts = points.contiguous().view(bs * num_p, 1, 3)
    c = pred_c.contiguous().view(bs * num_p)
    pred = torch.bmm(mp, base) + (pts + pt)
    if not refine and idx[0].item() in sym_list:
        t3 = tg[0].transpose(1, 0).contiguous().view(3, -1)
        p3 = pred.permute(2, 0, 1).contiguous().view(3, -1)
        inds = knn(t3.unsqueeze(0), p3.unsqueeze(0))
        t3 = torch.index_select(t3, 1, inds.view(-1) - 1)
        tg = t3.view(3, bs * num_p, num_pt_mesh).permute(1, 2, 0).contiguous()
        pred = p3.view(3, bs * num_p, num_pt_mesh).permute(1, 2, 0).contiguous()
    nrm = torch.norm(pred - tg, dim=2)
    dis = torch.mean(nrm, dim=1)
    std = torch.std(nrm, dim=1)
    loss = torch.mean((dis + 2 * std) * c - w * torch.log(c), dim=0)
    which = torch.max(c.view(bs, num_p), 1)[1][0]
    t = pt[which] + pts[which]
    ob = ori_base[which].view(1, 3, 3)
    new_points = torch.bmm(pts.view(1, bs * num_p, 3) - t.view(1, 1, 3), ob).contiguous()
    new_target = torch.bmm(ori_target[0].view(1, num_pt_mesh, 3) - t.view(1, 1, 3), ob).contiguous()
    return loss, dis.view(bs, num_p)[0][which], new_points, new_target, pred


def loss_refine_forward(pred_r, pred_t, target, model_points, idx, points, num_pt_mesh, sym_list, knn=knn1):
    # loss_refiner.py:12-64
    q = pred_r.view(1, 1, -1)
    q = q / torch.norm(q, dim=2).view(1, 1, 1)
    ori_base = quat_to_base(q).view(1, 3, 3)
    base = ori_base.transpose(2, 1).contiguous()
    mp = model_points.view(1, num_pt_mesh, 3)
    tg = target.view(1, num_pt_mesh, 3)
    ori_target = tg
    pt = pred_t.view(1, 1, 3)
    pred = torch.bmm(mp, base) + pt
    if idx[0].item() in sym_list:
        t3 = tg[0].transpose(1, 0).contiguous().view(3, -1)
        p3 = pred.permute(2, 0, 1).contiguous().view(3, -1)
        inds = knn(t3.unsqueeze(0), p3.unsqueeze(0))
        t3 = torch.index_select(t3, 1, inds.view(-1) - 1)
        tg = t3.view(3, 1, num_pt_mesh).permute(1, 2, 0).contiguous()
        pred = p3.view(3, 1, num_pt_mesh).permute(1, 2, 0).contiguous()
    dis = torch.mean(torch.norm(pred - tg, dim=2), dim=1)
    n_in = points.shape[1]
    new_points = torch.bmm(points.view(1, n_in, 3) - pt, ori_base).contiguous()
    new_target = torch.bmm(ori_target - pt, ori_base).contiguous()
    return dis, new_points, new_target, pred


# ----------------------------------------------------------------------------------------------
# bbox / selection / back-projection  (myDatasetAugmented/dataset.py:338-380, pipeline/utils.py:518-561)
# ----------------------------------------------------------------------------------------------
def get_bbox(label, img_h=480, img_w=640):
    # dataset.py:342-380: each side rounded UP to the next multiple of 40 unless already one (strict < >),
    # re-centred with int() truncation, shifted inside the image.
    rows = np.any(label, axis=1)
    cols = np.any(label, axis=0)
    rr = np.where(rows)[0]
    cc = np.where(cols)[0]
    rmin, rmax, cmin, cmax = int(rr[0]), int(rr[-1]) + 1, int(cc[0]), int(cc[-1]) + 1

    def up40(v):
        return v if v % 40 == 0 else (v // 40 + 1) * 40

    r_b, c_b = up40(rmax - rmin), up40(cmax - cmin)
    cr, ccn = int((rmin + rmax) / 2), int((cmin + cmax) / 2)
    rmin, rmax = cr - int(r_b / 2), cr + int(r_b / 2)
    cmin, cmax = ccn - int(c_b / 2), ccn + int(c_b / 2)
    if rmin < 0:
        rmax += -rmin
        rmin = 0
    if cmin < 0:
        cmax += -cmin
        cmin = 0
    if rmax > img_h:
        rmin -= rmax - img_h
        rmax = img_h
    if cmax > img_w:
        cmin -= cmax - img_w
        cmax = img_w
    return rmin, rmax, cmin, cmax


def select_choose(mask_crop_flat_nonzero, num_points, shuffle_mask=None):
    """pipeline/utils.py:529-539.  `shuffle_mask` injects the 0/1 selection the reference draws with the
    unseeded np.random.shuffle (it keeps raster order); <=N candidates are wrap-padded."""
    choose = mask_crop_flat_nonzero
    if len(choose) > num_points:
        if shuffle_mask is None:
            raise ValueError("inject shuffle_mask for deterministic parity")
        return choose[shuffle_mask.nonzero()]
    return np.pad(choose, (0, num_points - len(choose)), "wrap")


def backproject(depth, choose, rmin, rmax, cmin, cmax, meta):
    """pipeline/utils.py:542-553 -- float32 numpy with Python-float intrinsics (so NEP-50 keeps float32)."""
    d = depth[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    wc = cmax - cmin
    rows = (choose // wc + rmin)[:, np.newaxis].astype(np.float32)   # xmap (row index) pipeline/utils.py:518
    cols = (choose % wc + cmin)[:, np.newaxis].astype(np.float32)    # ymap (col index) pipeline/utils.py:519
    pt2 = d * meta["depth_scale"]
    pt0 = (cols - meta["intr"]["ppx"]) * pt2 / meta["intr"]["fx"]
    pt1 = (rows - meta["intr"]["ppy"]) * pt2 / meta["intr"]["fy"]
    return np.concatenate((pt0, pt1, pt2), axis=1).astype(np.float32)


_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
_STD = np.array([0.229, 0.224, 0.225], np.float32)


def crop_image(rgb, rmin, rmax, cmin, cmax):
    """pipeline/utils.py:559-560 -- raw 0..255 floats, ImageNet mean/std, NO /255."""
    img = np.transpose(rgb[:, :, :3], (2, 0, 1))[:, rmin:rmax, cmin:cmax].astype(np.float32)
    t = torch.from_numpy(img)
    return ((t - torch.from_numpy(_MEAN)[:, None, None]) / torch.from_numpy(_STD)[:, None, None]).unsqueeze(0)


def seg_input(rgb):
    """pipeline/utils.py:421-427 -- ToTensor (/255) then Normalize."""
    t = torch.from_numpy(np.ascontiguousarray(rgb[:, :, :3])).permute(2, 0, 1).float().div(255)
    return ((t - torch.from_numpy(_MEAN)[:, None, None]) / torch.from_numpy(_STD)[:, None, None]).unsqueeze(0)


# ----------------------------------------------------------------------------------------------
# segmentation post-processing  (pipeline/utils.py:430-469; twin at label_generator/create_labels.py:127-147)
# ----------------------------------------------------------------------------------------------
def connected_components8(binary):
    """Raster-order 8-connectivity labelling (labels 1.. in order of each component's first pixel in
    raster scan).  Stands in for cv2.connectedComponents(connectivity=8) -- numbering is UNPINNED against
    OpenCV; it only matters for exact score ties at pipeline/utils.py:461."""
    from scipy import ndimage
    lab, n = ndimage.label(binary != 0, structure=np.ones((3, 3), np.int32))
    return n + 1, lab.astype(np.int32)


def seg_postprocess(pred, min_pixels=100):
    """pred: softmaxed [C,H,W] float32 tensor.  Returns {cls(int>=1): mask u8 {0,255}} following
    pipeline/utils.py:435-469 (classes with >100 px; best component = highest mean class-probability,
    strict '>' so the first wins ties; score computed on cls_pred = cls * prob as the reference does)."""
    pred_arg = torch.argmax(pred, dim=0).numpy()
    found, counts = np.unique(pred_arg, return_counts=True)
    out = {}
    for cls, cnt in zip(found, counts):
        if cls == 0 or cnt <= min_pixels:
            continue
        cls_arg = np.where(pred_arg == cls, pred_arg, 0)
        cls_pred = cls_arg * pred[cls].numpy()
        _, labels = connected_components8(cls_arg.astype(np.uint8))
        biggest, biggest_score = 1, 0
        for u in np.unique(labels):
            if u == 0:
                continue
            score = np.mean(cls_pred[labels == u])
            if score > biggest_score:
                biggest_score, biggest = score, u
        cls_pred = np.where(labels != biggest, 0, cls_pred)
        out[int(cls)] = np.where(cls_pred != 0, 255, 0).astype(np.uint8)
    return out


def segmentor_predict(seg_sd, x, classes, backend="resnet18"):
    """`get_model('PsPNet', cfg).predict(x)` as the build defines it (SURVEY.md 8c DECISION): the in-repo
    PSPNet; first `classes` channels of the `final` 1x1 conv; activation='softmax' applied by predict
    (create_labels.py:23).  full_prediction then applies softmax AGAIN (pipeline/utils.py:430)."""
    logits = pspnet_forward(seg_sd, x, "", backend, logits_only=True)[:, :classes]
    return F.softmax(logits, dim=1)


def full_prediction(rgb, depth, meta, seg_sd, est_sd, ref_sd, class_names, num_points=1000,
                    choose_masks=None, backend="resnet18", refine_mode="live_compat", choose_fn=None, inject_logits=None):
    """CPU restatement of pipeline/utils.py:410-641 without the drawing code.

    refine_mode 'live_compat' reproduces the reference's live loop literally (:569-571: two refiner
    forwards on the SAME new_points, one composition); 'iterative' follows DenseFusion/tools/eval_ycb.py:205-229.
    `choose_masks[cls_name]` injects the random sub-selection (see select_choose); `choose_fn(name, nz, N)` may return the
    chosen crop indices directly.  `inject_logits[1,C,H,W]` replaces the segmentor's raw logits."""
    n_cls = len(class_names)
    x = seg_input(rgb)
    with torch.no_grad():
        if inject_logits is None:
            pred = F.softmax(segmentor_predict(seg_sd, x, n_cls + 1, backend), dim=1)[0]
        else:
            pred = F.softmax(F.softmax(inject_logits[:, :n_cls + 1], dim=1), dim=1)[0]
    masks = seg_postprocess(pred)
    out = {}
    for cls, mask in masks.items():
        name = class_names[cls - 1]
        mask_label = mask == 255
        rmin, rmax, cmin, cmax = get_bbox(mask_label, *mask.shape)
        m = mask_label * (depth != 0)
        nz = m[rmin:rmax, cmin:cmax].flatten().nonzero()[0]
        if len(nz) == 0:
            continue
        if choose_fn is not None:
            choose = np.asarray(choose_fn(name, nz, num_points))
        else:
            choose = select_choose(nz, num_points, None if choose_masks is None else choose_masks.get(name))
        pts = torch.from_numpy(backproject(depth, choose, rmin, rmax, cmin, cmax, meta)).unsqueeze(0)
        ch = torch.from_numpy(choose.astype(np.int64)).view(1, 1, -1)
        img = crop_image(rgb, rmin, rmax, cmin, cmax)
        idx = torch.tensor([[class_names.index(name)]], dtype=torch.int64)
        with torch.no_grad():
            pr, pt, pc, emb = posenet_forward(est_sd, img, pts, ch, idx, n_cls)
            new_points = get_new_points(pr, pt, pc, pts)
            _, my_r, my_t = estimator_prediction(pr, pt, pc, num_points, 1, pts)
            if refine_mode == "iterative":   # DenseFusion/tools/eval_ycb.py:205-229
                for _ in range(2):
                    T = torch.from_numpy(my_t.astype(np.float32)).view(1, 1, 3)
                    R = torch.from_numpy(quaternion_matrix(my_r)[:3, :3].astype(np.float32)).view(1, 3, 3)
                    new_cloud = torch.bmm(pts - T, R).contiguous()
                    rr, rt = refiner_forward(ref_sd, new_cloud, emb, idx, n_cls)
                    _, my_r, my_t = refined_prediction(rr, rt, my_r, my_t)
            else:                            # pipeline/utils.py:569-571, literally
                for _ in range(2):
                    rr, rt = refiner_forward(ref_sd, new_points, emb, idx, n_cls)
                _, my_r, my_t = refined_prediction(rr, rt, my_r, my_t)
        out[name] = {"mask": mask, "position": my_t, "rotation": my_r,
                     "bbox": (rmin, rmax, cmin, cmax), "choose": choose}
    return out
