"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE (build container only).

    make -C oracle ref && python tools/gen_golden.py

Imports the reference from /root/reference through tools/ref_shim.py, loads the deterministic
synthetic weights (autoposeestimation_amd/synthetic.py -- strict=True, which also proves our key
names/shapes equal the reference's), runs the reference modules/functions on seeded inputs and
stores inputs + expected outputs.  Fixtures are data only; no reference source is stored.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
warnings.filterwarnings("ignore")

import ref_shim  # noqa: E402

ref_shim.install()

from DenseFusion.lib.network import PoseNet, PoseRefineNet  # noqa: E402
from DenseFusion.lib.loss import Loss  # noqa: E402
from DenseFusion.lib.loss_refiner import Loss_refine  # noqa: E402
from DenseFusion.lib.pspnet import PSPNet  # noqa: E402
from DenseFusion.tools.utils import my_estimator_prediction, my_refined_prediction, get_new_points  # noqa: E402
from DenseFusion.datasets.myDatasetAugmented.dataset import get_bbox  # noqa: E402
from DenseFusion.lib.transformations import quaternion_matrix, quaternion_from_matrix  # noqa: E402
import pc_reconstruction.open3d_utils as ref_pc  # noqa: E402

from autoposeestimation_amd import synthetic as S  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)
N_TAP = 2048


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


def crop_inputs(frame_id, cls, box, size, bbox, n, meta, seed):
    """Selection + back-projection + crop: the INLINE arithmetic of pipeline/utils.py:518-561, executed
    here statement by statement (it is not a callable function in the reference)."""
    rgb, depth, label = S.synthetic_frame(frame_id, cls=cls, box=box, size=size)
    rmin, rmax, cmin, cmax = bbox
    xmap = np.array([[j for i in range(640)] for j in range(480)])
    ymap = np.array([[i for i in range(640)] for j in range(480)])
    mask = (label == cls) * (depth != 0)
    choose = mask[rmin:rmax, cmin:cmax].flatten().nonzero()[0]
    rng = np.random.default_rng(seed)
    c_mask = None
    if len(choose) > n:
        c_mask = np.zeros(len(choose), dtype=int)
        c_mask[:n] = 1
        rng.shuffle(c_mask)            # reference: unseeded np.random.shuffle (:536) -- injected here
        choose = choose[c_mask.nonzero()]
    else:
        choose = np.pad(choose, (0, n - len(choose)), 'wrap')
    depth_masked = depth[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    xmap_masked = xmap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    ymap_masked = ymap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    pt2 = depth_masked * meta['depth_scale']
    pt0 = (ymap_masked - meta['intr']['ppx']) * pt2 / meta['intr']['fx']
    pt1 = (xmap_masked - meta['intr']['ppy']) * pt2 / meta['intr']['fy']
    points = np.concatenate((pt0, pt1, pt2), axis=1).astype(np.float32)
    img = np.transpose(np.array(rgb)[:, :, :3], (2, 0, 1))[:, rmin:rmax, cmin:cmax].astype(np.float32)
    img = (img - MEAN[:, None, None]) / STD[:, None, None]     # torchvision Normalize (pipeline/utils.py:560)
    return rgb, depth, label, choose, c_mask, points, img


def tap_samples(taps, seed):
    out = {}
    for k, v in taps.items():
        flat = v.detach().reshape(-1)
        rng = np.random.default_rng([seed, len(k), flat.numel()])
        idx = np.sort(rng.choice(flat.numel(), size=min(N_TAP, flat.numel()), replace=False))
        out["tap_" + k + "_idx"] = idx.astype(np.int64)
        out["tap_" + k + "_val"] = flat[idx].numpy()
        out["tap_" + k + "_shape"] = np.array(v.shape, np.int64)
    return out


def gen_posenet():
    cases = [  # name, N, num_obj, frame, cls, box(r0,c0), size(h,w), obj index, meta
        ("cfg1_n500_o21_80x120", 500, 21, 11, 5, (200, 300), (75, 110), 4, S.YCB_META),
        ("n1000_o12_40x40", 1000, 12, 12, 2, (100, 100), (33, 38), 1, S.REALSENSE_META),
        ("n1000_o12_160x160", 1000, 12, 0, 4, (150, 250), (150, 150), 3, S.REALSENSE_META),
        ("n1000_o12_120x200", 1000, 12, 13, 7, (300, 400), (101, 170), 6, S.REALSENSE_META),
    ]
    # synthetic-weight seed per case: with seed 0 the tiny 40x40 crop saturates the confidence head (c in 0.9978..0.9999 and a top-2
    # margin of 1e-6, below fp32 noise: the arg-max winner -- hence R, t -- was not comparable); seed 4 gives c in 0.08..0.15 and a
    # margin of 1e-2
    wseeds = {"n1000_o12_40x40": 4}
    only = os.environ.get("APE_GOLDEN_ONLY")
    for name, n, num_obj, frame, cls, box, size, obj, meta in cases:
        if only and only != "posenet_" + name:
            continue
        est = PoseNet(num_points=n, num_obj=num_obj).eval()
        wseed = wseeds.get(name, 0)
        est.load_state_dict(S.posenet_state_dict(num_obj, seed=wseed), strict=True)
        refiner = PoseRefineNet(num_points=n, num_obj=num_obj).eval()
        refiner.load_state_dict(S.refiner_state_dict(num_obj, seed=wseed), strict=True)
        lab = np.zeros((480, 640), bool)
        lab[box[0]:box[0] + size[0], box[1]:box[1] + size[1]] = True
        bbox = tuple(int(v) for v in get_bbox(lab))
        rgb, depth, label, choose, c_mask, points, img = crop_inputs(frame, cls, box, size, bbox, n, meta, seed=frame)
        taps = {}
        psp = est.cnn.model.module
        hooks = []
        for tn in ("feats", "psp", "up_1", "up_2", "up_3", "final"):
            hooks.append(getattr(psp, tn).register_forward_hook(
                lambda m, i, o, tn=tn: taps.__setitem__(tn, o[0] if isinstance(o, tuple) else o)))
        hooks.append(est.feat.register_forward_hook(lambda m, i, o: taps.__setitem__("posenetfeat", o)))
        t_img = torch.from_numpy(img).unsqueeze(0)
        t_pts = torch.from_numpy(points).unsqueeze(0)
        t_ch = torch.LongTensor(np.array([choose]).astype(np.int32)).unsqueeze(0)
        t_idx = torch.LongTensor([obj]).unsqueeze(0)
        with torch.no_grad():
            pred_r, pred_t, pred_c, emb = est(t_img, t_pts, t_ch, t_idx)
            new_points = get_new_points(pred_r, pred_t, pred_c, t_pts)
            _, my_r, my_t = my_estimator_prediction(pred_r, pred_t, pred_c, n, 1, t_pts)
            for ite in range(0, 2):
                ref_r, ref_t = refiner(new_points, emb, t_idx)
            _, fin_r, fin_t = my_refined_prediction(ref_r, ref_t, my_r, my_t)
            # iterative form, DenseFusion/tools/eval_ycb.py:205-229 (CPU tensors instead of .cuda())
            it_r, it_t = my_r, my_t
            for ite in range(0, 2):
                T = torch.from_numpy(it_t.astype(np.float32)).view(1, 3).repeat(n, 1).contiguous().view(1, n, 3)
                my_mat = quaternion_matrix(it_r)
                R = torch.from_numpy(my_mat[:3, :3].astype(np.float32)).view(1, 3, 3)
                new_cloud = torch.bmm((t_pts - T), R).contiguous()
                rr, rt = refiner(new_cloud, emb, t_idx)
                _, it_r, it_t = my_refined_prediction(rr, rt, it_r, it_t)
        for h in hooks:
            h.remove()
        c = pred_c.view(-1)
        top2 = torch.topk(c, 2).values
        save("posenet_" + name,
             n=n, num_obj=num_obj, obj=obj, bbox=np.array(bbox), frame=frame, cls=cls, wseed=wseed,
             box=np.array(box), size=np.array(size),
             fx=meta['intr']['fx'], fy=meta['intr']['fy'], ppx=meta['intr']['ppx'], ppy=meta['intr']['ppy'],
             depth_scale=meta['depth_scale'],
             rgb_crop=rgb[bbox[0]:bbox[1], bbox[2]:bbox[3]], depth_crop=depth[bbox[0]:bbox[1], bbox[2]:bbox[3]],
             choose=choose.astype(np.int64), c_mask=(c_mask if c_mask is not None else np.zeros(0, int)),
             points=points, img=img,
             pred_r=pred_r.numpy(), pred_t=pred_t.numpy(), pred_c=pred_c.numpy(), emb=emb.numpy(),
             new_points=new_points.numpy(), my_r=my_r, my_t=my_t, ref_r=ref_r.numpy(), ref_t=ref_t.numpy(),
             fin_r=fin_r, fin_t=fin_t, it_r=it_r, it_t=it_t, c_margin=float(top2[0] - top2[1]),
             **tap_samples(taps, seed=frame))


def gen_pspnet_seg():
    """In-repo PSPNet as the 'PsPNet' segmentor (SURVEY 8c DECISION): small 3x96x128 input, r18 and r34."""
    for backend in ("resnet18", "resnet34"):
        net = PSPNet(sizes=(1, 2, 3, 6), psp_size=512, deep_features_size=256, backend=backend).eval()
        net.load_state_dict(S.pspnet_state_dict(backend, seed=3), strict=True)
        rgb, _, _ = S.synthetic_frame(21, cls=3, box=(20, 30), size=(50, 60), h=96, w=128)
        x = torch.from_numpy(rgb).permute(2, 0, 1).float().div(255)
        x = ((x - torch.from_numpy(MEAN)[:, None, None]) / torch.from_numpy(STD)[:, None, None]).unsqueeze(0)
        with torch.no_grad():
            out = net(x)
        save("pspnet_%s_96x128" % backend, rgb=rgb, x=x.numpy(), out=out.numpy())


def gen_pose_utils():
    rng = np.random.default_rng(7)
    n = 1000
    pred_r = torch.from_numpy(rng.standard_normal((1, n, 4)).astype(np.float32))
    pred_t = torch.from_numpy((rng.standard_normal((1, n, 3)) * 0.05).astype(np.float32))
    pred_c = torch.from_numpy(rng.random((1, n, 1)).astype(np.float32))
    pts = torch.from_numpy((rng.standard_normal((1, n, 3)) * 0.2 + [0, 0, 0.6]).astype(np.float32))
    newp = get_new_points(pred_r, pred_t, pred_c, pts)
    _, my_r, my_t = my_estimator_prediction(pred_r, pred_t, pred_c, n, 1, pts)
    out = dict(pred_r=pred_r.numpy(), pred_t=pred_t.numpy(), pred_c=pred_c.numpy(), points=pts.numpy(),
               new_points=newp.numpy(), my_r=my_r, my_t=my_t)
    # refine compose: exercise both branches of quaternion_from_matrix(isprecise=True)
    rr = rng.standard_normal((8, 4)).astype(np.float32)
    rr[0] = [1, 0, 0, 0]
    rr[1] = [0.01, 1, 0.2, 0.1]     # trace <= M[3,3] branch
    rr[2] = [0.0, 0.1, 1, 0.2]
    rr[3] = [-0.02, 0.1, 0.2, 1]
    tt = (rng.standard_normal((8, 3)) * 0.02).astype(np.float32)
    fr, ft = [], []
    for i in range(8):
        base_r = my_r if i % 2 == 0 else np.array(rr[(i + 3) % 8] / np.linalg.norm(rr[(i + 3) % 8]), dtype=np.float32)
        _, r, t = my_refined_prediction(torch.from_numpy(rr[i]).view(1, 4), torch.from_numpy(tt[i]).view(1, 3),
                                        base_r, my_t)
        fr.append(r)
        ft.append(t)
    out.update(ref_r=rr, ref_t=tt, fin_r=np.array(fr), fin_t=np.array(ft))
    qm = np.array([quaternion_matrix(np.asarray(q, np.float64)) for q in rr])
    out.update(quat_mats=qm)
    save("pose_utils", **out)


def gen_knn():
    rng = np.random.default_rng(3)
    cases = {}
    # (B, D, Nr, Nq, k, quantise)
    specs = [(1, 3, 500, 500, 1, 0), (1, 3, 1000, 1500, 1, 0), (2, 3, 257, 300, 1, 0), (1, 3, 64, 200, 1, 0.25),
             (1, 3, 300, 200, 1, 0.5), (2, 3, 50, 70, 4, 0), (1, 3, 40, 60, 3, 0.5), (1, 8, 33, 65, 2, 0),
             (1, 3, 1, 17, 1, 0), (1, 3, 7, 0, 1, 0)]
    for i, (b, d, nr, nq, k, qz) in enumerate(specs):
        ref = rng.standard_normal((b, d, nr)).astype(np.float32)
        qry = rng.standard_normal((b, d, nq)).astype(np.float32)
        if qz:
            ref = (np.round(ref / qz) * qz).astype(np.float32)   # crafted exact distance ties + duplicates
            qry = (np.round(qry / qz) * qz).astype(np.float32)
        idx = ref_shim.ref_knn(torch.from_numpy(ref), torch.from_numpy(qry), k).numpy()
        cases["ref_%d" % i], cases["query_%d" % i], cases["idx_%d" % i] = ref, qry, idx
    cases["n_cases"] = len(specs)
    save("knn", **cases)


def gen_loss():
    rng = np.random.default_rng(9)
    out = {}
    ci = 0
    for (n, m) in ((60, 50), (200, 120)):
        pred_r = torch.from_numpy(rng.standard_normal((1, n, 4)).astype(np.float32))
        pred_t = torch.from_numpy((rng.standard_normal((1, n, 3)) * 0.03).astype(np.float32))
        pred_c = torch.from_numpy((rng.random((1, n, 1)) * 0.9 + 0.05).astype(np.float32))
        pts = torch.from_numpy((rng.standard_normal((1, n, 3)) * 0.1 + [0, 0, 0.6]).astype(np.float32))
        model = torch.from_numpy(((rng.random((1, m, 3)) - 0.5) * 0.1).astype(np.float32))
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        Rm = quaternion_matrix(q)[:3, :3].astype(np.float32)
        target = torch.from_numpy((model[0].numpy() @ Rm.T + np.array([0.02, -0.01, 0.6], np.float32))[None])
        idx = torch.LongTensor([2]).view(1)
        for sym in ([], [2]):
            for refine in (False, True):
                crit = Loss(m, sym)
                loss, dis, newp, newt, pred = crit(pred_r, pred_t, pred_c, target, model, idx, pts, 0.015, refine)
                p = "c%d_" % ci
                out.update({p + "n": n, p + "m": m, p + "sym": int(bool(sym)), p + "refine": int(refine),
                            p + "pred_r": pred_r.numpy(), p + "pred_t": pred_t.numpy(), p + "pred_c": pred_c.numpy(),
                            p + "points": pts.numpy(), p + "model": model.numpy(), p + "target": target.numpy(),
                            p + "loss": loss.numpy(), p + "dis": dis.numpy(), p + "new_points": newp.numpy(),
                            p + "new_target": newt.numpy()})
                # refiner loss on the re-centred cloud
                rr = torch.from_numpy(rng.standard_normal((1, 4)).astype(np.float32) * 0.1 + np.array([[1, 0, 0, 0]], np.float32))
                rt = torch.from_numpy((rng.standard_normal((1, 3)) * 0.01).astype(np.float32))
                crit_r = Loss_refine(m, sym)
                d2, np2, nt2, _ = crit_r(rr, rt, newt, model, idx, newp)
                out.update({p + "rr": rr.numpy(), p + "rt": rt.numpy(), p + "r_dis": d2.numpy(),
                            p + "r_new_points": np2.numpy(), p + "r_new_target": nt2.numpy()})
                ci += 1
    out["n_cases"] = ci
    save("loss", **out)


def gen_bbox():
    rects = [(100, 190, 200, 333), (0, 35, 0, 38), (0, 479, 0, 639), (440, 479, 600, 639), (10, 50, 10, 90),
             (200, 240, 300, 380), (5, 6, 630, 639), (470, 479, 0, 5), (120, 321, 77, 400), (0, 41, 599, 639),
             (239, 241, 319, 321), (60, 100, 20, 61)]
    boxes = []
    for (r0, r1, c0, c1) in rects:      # inclusive extents
        lab = np.zeros((480, 640), bool)
        lab[r0:r1 + 1, c0:c1 + 1] = True
        boxes.append([int(v) for v in get_bbox(lab)])
    # non-rectangular blob
    lab = np.zeros((480, 640), bool)
    yy, xx = np.mgrid[0:480, 0:640]
    lab[((yy - 300) / 70.0) ** 2 + ((xx - 500) / 130.0) ** 2 < 1] = True
    rects.append((-1, -1, -1, -1))
    boxes.append([int(v) for v in get_bbox(lab)])
    save("bbox", rects=np.array(rects), boxes=np.array(boxes), blob=np.packbits(lab))


def gen_pc_utils():
    rng = np.random.default_rng(5)
    pts = rng.standard_normal((200, 3)) * 50 + [0, 0, 600]
    intr = dict(S.REALSENSE_META["intr"])
    px = np.array(ref_pc.points2pixel(pts, intr))

    class _PC:
        def __init__(self, p):
            self.points = p

        def get_center(self):
            return np.mean(self.points, axis=0)
    ctr = ref_pc.get_my_source_center(_PC(pts))
    depth = rng.integers(0, 900, size=(480, 640)).astype(np.float64)
    depth[rng.random((480, 640)) < 0.1] = 0
    pix = np.stack([rng.integers(0, 480, 300), rng.integers(0, 640, 300)], 1)
    p3 = np.array(ref_pc.pixels2points(pix, depth, intr))
    save("pc_utils", points=pts, pixels=px, centre=ctr, depth=depth.astype(np.uint16), pix=pix, pix_points=p3)


if __name__ == "__main__":
    torch.manual_seed(0)
    if os.environ.get("APE_GOLDEN_ONLY", "").startswith("posenet_"):       # regenerate one PoseNet fixture, leave the rest untouched
        gen_posenet()
        sys.exit(0)
    gen_knn()
    gen_bbox()
    gen_pose_utils()
    gen_loss()
    gen_pc_utils()
    gen_pspnet_seg()
    gen_posenet()
