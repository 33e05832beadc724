"""End-to-end parity of the live path (segmentor -> masks -> crop -> PoseNet -> 2x refiner -> pose) on synthetic 640x480
RGB-D frames against the CPU oracle restatement of pipeline/utils.py:410-641.  Mask bit-exact, R/t within 1e-4."""
import numpy as np
import pytest
import torch

from autoposeestimation_amd import synthetic as S
from oracle import densefusion_oracle as O

pytestmark = pytest.mark.gpu
CLASSES = ["obj%02d" % i for i in range(12)]


def _models(backend="resnet18"):
    from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet
    from autoposeestimation_amd.segmentation.utils import get_model
    seg = get_model("PsPNet", {"encoder_name": backend, "encoder_weights": None, "activation": "softmax",
                               "in_channels": 3, "classes": 13})
    seg_sd = S.pspnet_state_dict(backend, seed=5, stem_gain=1.0)
    seg.load_state_dict(seg_sd)
    est = PoseNet(1000, 12)
    est_sd = S.posenet_state_dict(12, 0)
    est.load_state_dict(est_sd)
    ref = PoseRefineNet(1000, 12)
    ref_sd = S.refiner_state_dict(12, 0)
    ref.load_state_dict(ref_sd)
    return seg.cuda().eval(), est.cuda().eval(), ref.cuda().eval(), seg_sd, est_sd, ref_sd


def _fit_segmentor(seg, seg_sd, frames):
    """"train" the final 1x1 conv by least squares on the HIP features of the given frames (see synthetic.fit_final_layer)"""
    from autoposeestimation_amd import engine as E
    feats, labels = [], []
    for rgb, _, label in frames:
        x4 = E.preprocess_u8(torch.from_numpy(rgb[None]).cuda(), torch.zeros(1, 3, dtype=torch.int32).cuda(), 480, 640, True)
        f = seg.plan().features(x4)[0].reshape(-1, 64)
        lab = torch.from_numpy(label.reshape(-1).astype(np.int64))
        rng = np.random.default_rng(0)
        fg = np.nonzero(label.reshape(-1))[0]
        bg = rng.choice(np.nonzero(label.reshape(-1) == 0)[0], size=len(fg), replace=False)
        sel = torch.from_numpy(np.concatenate([fg, bg]))
        feats.append(f[sel.cuda()])
        labels.append(lab[sel])
    w, b = S.fit_final_layer(torch.cat(feats), torch.cat(labels), 13)
    seg_sd = dict(seg_sd)
    fw, fb = seg_sd["final.0.weight"].clone(), seg_sd["final.0.bias"].clone()
    fw[:13, :, 0, 0], fb[:13] = w, b
    seg_sd["final.0.weight"], seg_sd["final.0.bias"] = fw, fb
    prec = seg.precision
    seg.load_state_dict(seg_sd)
    return seg.cuda().eval().set_precision(prec), seg_sd


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("refine_mode", ["live_compat", "iterative"])
def test_full_prediction_vs_oracle(refine_mode, precision):
    """precision 'bf16x3' is the configuration bench.py measures (split-bf16 operands on the bf16 matrix cores)."""
    from autoposeestimation_amd.pipeline.utils import full_prediction
    seg, est, ref, seg_sd, est_sd, ref_sd = _models()
    for m in (seg, est, ref):
        m.set_precision(precision)
    frames = [S.synthetic_frame(100 + i, cls=c, box=bx, size=sz) for i, (c, bx, sz) in
              enumerate([(4, (150, 250), (150, 150)), (9, (40, 60), (70, 110))])]
    seg, seg_sd = _fit_segmentor(seg, seg_sd, frames)
    seg.set_precision(precision)
    meta = S.REALSENSE_META
    for rgb, depth, label in frames:
        chosen = {}

        def choose_fn(name, nz, n):
            rng = np.random.default_rng(len(nz))
            ch = np.sort(rng.choice(nz, size=n, replace=False)) if len(nz) > n else np.pad(nz, (0, n - len(nz)), "wrap")
            chosen[name] = ch
            return ch

        want = O.full_prediction(rgb, depth, meta, seg_sd, est_sd, ref_sd, CLASSES, choose_fn=choose_fn, refine_mode=refine_mode)
        got = full_prediction(rgb, depth, meta, seg, est, ref, None, None, torch.device("cuda:0"), True, {},
                              class_names=CLASSES, refine_mode=refine_mode, choose_override=chosen)
        assert set(got["predictions"]) == set(want) and len(want) >= 1
        assert set(got["elapsed_times"]) == {"segmentation", "pose_estimation", "total"}
        for name, w in want.items():
            g = got["predictions"][name]
            diff = int((g["mask"] != w["mask"]).sum())
            assert diff == 0, "mask differs in %d pixels" % diff
            assert g["mask"].dtype == np.uint8 and set(np.unique(g["mask"])) <= {0, 255}
            q = g["rotation"] if np.dot(g["rotation"], w["rotation"]) >= 0 else -g["rotation"]
            assert np.abs(q - w["rotation"]).max() <= 1e-4, (name, np.abs(q - w["rotation"]).max())
            assert np.abs(g["position"] - w["position"]).max() <= 1e-4, (name, np.abs(g["position"] - w["position"]).max())


def test_batched_pipeline_equals_single_frames():
    """FramePipeline over B frames (mixed crop sizes -> two buckets) == per-frame full_prediction."""
    from autoposeestimation_amd.pipeline.utils import FramePipeline, full_prediction
    seg, est, ref, seg_sd, _, _ = _models()
    spec = [(1, (150, 250), (150, 150)), (2, (200, 100), (150, 150)), (3, (40, 60), (70, 110)), (5, (300, 400), (150, 150))]
    frames = [S.synthetic_frame(200 + i, cls=c, box=bx, size=sz) for i, (c, bx, sz) in enumerate(spec)]
    seg, _ = _fit_segmentor(seg, seg_sd, frames)
    rgb = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    depth = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    pipe = FramePipeline(seg, est, ref, CLASSES)
    out = pipe.run(rgb, depth, S.REALSENSE_META, seed=3)
    assert len(out["objects"]) >= len(frames)
    choose = out["choose"].cpu().numpy()
    pose = out["pose"].cpu().numpy()
    for i, o in enumerate(out["objects"]):
        b, cls = o[0], o[1]
        single = full_prediction(frames[b][0], frames[b][1], S.REALSENSE_META, seg, est, ref, None, None,
                                 torch.device("cuda:0"), True, {}, class_names=CLASSES,
                                 choose_override={CLASSES[cls - 1]: choose[i]})
        p = single["predictions"][CLASSES[cls - 1]]
        np.testing.assert_allclose(pose[i, :4], p["rotation"], atol=1e-6)
        np.testing.assert_allclose(pose[i, 4:], p["position"], atol=1e-6)
        assert np.array_equal((out["objmap"][b].cpu().numpy() == cls), p["mask"] == 255)


def test_pipeline_is_bitwise_deterministic():
    """Same frames, same seed -> identical masks, choose, poses (atomics only ever combine order-independent integers)."""
    from autoposeestimation_amd.pipeline.utils import FramePipeline
    seg, est, ref, seg_sd, _, _ = _models()
    frames = [S.synthetic_frame(300 + i, cls=1 + i % 3, box=(60 + 30 * i, 80 + 50 * i), size=(126, 126)) for i in range(6)]
    seg, _ = _fit_segmentor(seg, seg_sd, frames[:3])
    for m in (seg, est, ref):
        m.set_precision("bf16x3")
    rgb = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    depth = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    pipe = FramePipeline(seg, est, ref, CLASSES)
    a = pipe.run(rgb, depth, S.REALSENSE_META, seed=11)
    b = pipe.run(rgb, depth, S.REALSENSE_META, seed=11)
    assert a["objects"] == b["objects"] and len(a["objects"]) >= 6
    assert torch.equal(a["objmap"], b["objmap"]) and torch.equal(a["choose"], b["choose"]) and torch.equal(a["pose"], b["pose"])
    assert torch.isfinite(a["pose"]).all()
    # pose stage on a second stream (bench.py's cross-step overlap): same results, consumer waits for the returned stream
    side = FramePipeline(seg, est, ref, CLASSES, pose_stream=True)
    for _ in range(2):
        c = side.run(rgb, depth, S.REALSENSE_META, seed=11)
    c["stream"].synchronize()
    assert c["objects"] == a["objects"] and torch.equal(c["objmap"], a["objmap"])
    assert torch.equal(c["choose"], a["choose"]) and torch.equal(c["pose"], a["pose"])
    q = a["pose"][:, :4]
    assert torch.allclose(q.norm(dim=1), torch.ones(len(q), dtype=torch.float64, device=q.device), atol=1e-9) and (q[:, 0] >= 0).all()


def test_get_prediction_models_from_a_dataset_tree(tmp_path):
    """pipeline/utils.py:643-718 on a synthetic dataset tree in the reference's layout (classes.txt, <cls>.xyz model clouds in mm,
    pose_model.pth / pose_refine_model.pth state dicts, the segmentor checkpoint with a 'state_dict' entry): the 9-tuple contract, the
    weights really loaded, and the tuple drives full_prediction to the same result as directly built models."""
    import os
    from autoposeestimation_amd.pipeline.utils import full_prediction, get_prediction_models, read_xyz_cloud
    root, ds = str(tmp_path), "synth"
    seg, est, ref, seg_sd, est_sd, ref_sd = _models("resnet34")
    frames = [S.synthetic_frame(700 + i, cls=c, box=bx, size=sz) for i, (c, bx, sz) in enumerate([(2, (150, 250), (150, 150)), (7, (60, 80), (100, 120))])]
    seg, seg_sd = _fit_segmentor(seg, seg_sd, frames)
    os.makedirs(os.path.join(root, "label_generator", "data_sets", "segmentation", ds))
    with open(os.path.join(root, "label_generator", "data_sets", "segmentation", ds, "classes.txt"), "w") as f:
        f.write("\n".join(CLASSES) + "\n")
    rng = np.random.default_rng(0)
    clouds = {}
    for name in CLASSES:
        d = os.path.join(root, "pc_reconstruction", "data", name)
        os.makedirs(d)
        clouds[name] = (rng.random((1200, 3)) - 0.5) * 100.0          # mm
        with open(os.path.join(d, name + ".xyz"), "w") as f:
            for p in clouds[name]:
                f.write("[{} {} {}]\n".format(*p))                    # create_pointcloud.py:373-376 writes str(ndarray row)
    pose_dir = os.path.join(root, "DenseFusion", "trained_models", ds)
    os.makedirs(pose_dir)
    torch.save(est_sd, os.path.join(pose_dir, "pose_model.pth"))
    torch.save(ref_sd, os.path.join(pose_dir, "pose_refine_model.pth"))
    seg_dir = os.path.join(root, "segmentation", "trained_models", ds)
    os.makedirs(seg_dir)
    torch.save({"state_dict": seg_sd, "epoch": 3, "iou": 0.9}, os.path.join(seg_dir, "PsPNet_resnet34.ckpt"))

    out = get_prediction_models(root, ds)
    assert len(out) == 9
    segmentor, estimator, refiner, classes, to_tensor, normalize, cld, device, cuda = out
    assert classes == CLASSES and cuda is True and device.type == "cuda" and sorted(cld) == list(range(12))
    np.testing.assert_allclose(cld[3], clouds[CLASSES[3]] / 1000.0, rtol=0, atol=1e-12)             # mm -> m (pipeline/utils.py:667-686)
    np.testing.assert_allclose(read_xyz_cloud(os.path.join(root, "pc_reconstruction", "data", CLASSES[0], CLASSES[0] + ".xyz"), False), clouds[CLASSES[0]])
    assert torch.equal(estimator.state_dict()["conv4_r.weight"].cpu(), est_sd["conv4_r.weight"])
    assert torch.equal(refiner.state_dict()["conv3_t.bias"].cpu(), ref_sd["conv3_t.bias"])
    assert torch.equal(segmentor.state_dict()["final.0.weight"].cpu(), seg_sd["final.0.weight"])
    t = normalize(to_tensor(frames[0][0]))
    assert t.shape == (3, 480, 640) and abs(float(t.mean())) < 3
    rgb, depth, _ = frames[0]
    got = full_prediction(rgb, depth, S.REALSENSE_META, segmentor, estimator, refiner, to_tensor, normalize, device, cuda, {}, class_names=classes)
    want = full_prediction(rgb, depth, S.REALSENSE_META, seg, est, ref, None, None, torch.device("cuda:0"), True, {}, class_names=CLASSES)
    assert set(got["predictions"]) == set(want["predictions"]) and len(got["predictions"]) >= 1
    for name, w in want["predictions"].items():
        assert np.array_equal(got["predictions"][name]["mask"], w["mask"])
    # the reference's smp checkpoint name is recognised and refused with an explanation
    os.rename(os.path.join(seg_dir, "PsPNet_resnet34.ckpt"), os.path.join(seg_dir, "Unet_resnet34.ckpt"))
    with pytest.raises(FileNotFoundError, match="segmentation_models_pytorch"):
        get_prediction_models(root, ds)


@pytest.mark.parametrize("pose_stream", [False, True])
def test_pose_buckets_replayed_as_hip_graphs_equal_the_eager_launches(pose_stream):
    """FramePipeline(pose_graphs=True): frames with several objects of different sizes (pipeline/utils.py:444-470,522-561 loops over every
    detected class) give one pose-stage pass per crop-size bucket; each bucket's launches are captured in a HIP graph on its second
    occurrence and replayed afterwards.  Objects come from painted label maps (the product's own component / bbox kernels), the sampling
    seed changes every step (it reaches the captured launch through device memory): poses, candidate counts and chosen pixels of the
    eager step, the capturing step and two replays are those of a pipeline that never uses graphs, bit for bit."""
    from autoposeestimation_amd import engine as E
    from autoposeestimation_amd.pipeline.utils import FramePipeline
    _, est, ref, *_ = _models()
    for m in (est, ref):
        m.set_precision("bf16x3")
    frames = [S.mixed_frame(7000 + i) for i in range(6)]
    rgb = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    depth = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    label = torch.from_numpy(np.stack([f[2] for f in frames]).astype(np.uint8)).cuda()
    objmap, det = E.seg_components(label, torch.ones(label.shape, dtype=torch.float32, device="cuda"), 13, 100)
    handle = {"objmap": objmap, "det": det, "det_h": None, "event": None}
    plain = FramePipeline(None, est, ref, CLASSES, pose_stream=False)
    graphs = FramePipeline(None, est, ref, CLASSES, pose_stream=pose_stream, pose_graphs=True)
    sizes = set()
    for step in range(4):
        want = plain.finish(dict(handle), rgb, depth, S.REALSENSE_META, seed=step)
        got = graphs.finish(dict(handle), rgb, depth, S.REALSENSE_META, seed=step)
        torch.cuda.synchronize()
        assert got["objects"] == want["objects"] and len(want["objects"]) >= 8
        sizes |= {(o[3] - o[2], o[5] - o[4]) for o in want["objects"]}
        for k in ("pose", "n_cand", "choose"):
            assert torch.equal(got[k], want[k]), (step, k)
    assert len(sizes) >= 3 and len(graphs._graphs) == len(sizes)          # several buckets, one graph each
    # a seed that changes the sampled pixels really reaches the replayed launch
    def chosen(seed):
        out = graphs.finish(dict(handle), rgb, depth, S.REALSENSE_META, seed=seed)
        if out.get("stream") is not None:            # (the results belong to the pose stream: its consumer waits for it, FramePipeline.__init__)
            torch.cuda.current_stream().wait_stream(out["stream"])
        return out["choose"].clone()
    a, b = chosen(11), chosen(12)
    torch.cuda.synchronize()
    assert not torch.equal(a, b)
