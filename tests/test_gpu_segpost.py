"""GPU parity of the integer / byte stages (seg post-processing, bbox, choose, back-projection, normalisation) against
the oracle: bit-exact, from INJECTED logits so no conv rounding can blur the comparison."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from autoposeestimation_amd import synthetic as S
from oracle import densefusion_oracle as O

pytestmark = pytest.mark.gpu
H, W = 480, 640


def _blob_logits(seed, n_cls=13):
    """label image with several blobs per class (different confidences) + thin 8-connected bridges + a <=100 px class"""
    rng = np.random.default_rng(seed)
    lab = np.zeros((H, W), np.int64)
    conf = np.full((H, W), 2.0, np.float32)
    for k in range(9):
        c = int(rng.integers(1, 6))
        r0, c0 = int(rng.integers(0, H - 60)), int(rng.integers(0, W - 80))
        hh, ww = int(rng.integers(12, 120)), int(rng.integers(12, 160))
        lab[r0:r0 + hh, c0:c0 + ww] = c
        conf[r0:r0 + hh, c0:c0 + ww] = rng.uniform(1.0, 6.0)
    for k in range(6):           # diagonal one-pixel chains: only 8-connectivity joins them
        c = int(rng.integers(1, 6))
        r0, c0 = int(rng.integers(0, H - 40)), int(rng.integers(40, W - 40))
        sgn = 1 if k % 2 else -1
        for i in range(30):
            lab[r0 + i, c0 + sgn * i] = c
    lab[5:12, 5:15] = 7          # 70 px: below the >100 threshold
    lab[0:3, W - 50:W] = 8       # touches the border, 150 px
    lab[H - 2:H, 0:80] = 9
    logits = rng.standard_normal((n_cls, H, W)).astype(np.float32) * 0.3
    oh = np.eye(n_cls, dtype=np.float32)[lab].transpose(2, 0, 1)
    logits += oh * conf[None]
    return torch.from_numpy(logits)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_masks_and_bbox_bit_exact_from_injected_logits(seed):
    from autoposeestimation_amd import engine as E
    n_cls = 13
    logits = torch.stack([_blob_logits(seed * 10 + i, n_cls) for i in range(2)])          # [2,C,H,W]
    dev = logits.permute(0, 2, 3, 1).contiguous().cuda()
    label, score = E.seg_argmax(dev, n_cls, double_softmax=True)
    objmap, det = E.seg_components(label, score, n_cls, 100)
    objmap, det = objmap.cpu().numpy(), det.cpu().numpy()
    for b in range(2):
        pred = F.softmax(F.softmax(logits[b], dim=0), dim=0)
        assert np.array_equal(label[b].cpu().numpy(), torch.argmax(pred, 0).numpy().astype(np.uint8))
        want = O.seg_postprocess(pred)
        got_classes = {c for c in range(1, n_cls) if det[b, c, 0]}
        assert got_classes == set(want.keys())
        for c, mask in want.items():
            assert np.array_equal(np.where(objmap[b] == c, 255, 0).astype(np.uint8), mask), "class %d" % c
            assert tuple(det[b, c, 1:]) == O.get_bbox(mask == 255, H, W)
        assert not np.isin(objmap[b], list(set(range(1, n_cls)) - got_classes)).any()


@pytest.mark.parametrize("hw", [(61, 77), (64, 128), (37, 200)])
def test_components_on_speckle_odd_sizes(hw):
    """Run-based union-find on dense random labels (runs of every length, diagonal-only contacts, widths that are not a
    multiple of the 64-pixel wave): same best component per class as the oracle's raster-order 8-connectivity labelling."""
    from autoposeestimation_amd import engine as E
    h, w = hw
    n_cls = 5
    rng = np.random.default_rng(h * 1000 + w)
    lab = rng.integers(0, n_cls, size=(2, h, w))
    lab[0, : h // 2] = np.where(rng.random((h // 2, w)) < 0.6, 0, lab[0, : h // 2])        # sparser half: small components
    conf = rng.uniform(1.0, 5.0, size=(2, h, w)).astype(np.float32)
    logits = rng.standard_normal((2, n_cls, h, w)).astype(np.float32) * 0.05
    logits += np.eye(n_cls, dtype=np.float32)[lab].transpose(0, 3, 1, 2) * conf[:, None]
    logits = torch.from_numpy(logits)
    label, score = E.seg_argmax(logits.permute(0, 2, 3, 1).contiguous().cuda(), n_cls, double_softmax=True)
    objmap, det = E.seg_components(label, score, n_cls, 100)
    objmap, det = objmap.cpu().numpy(), det.cpu().numpy()
    for b in range(2):
        pred = F.softmax(F.softmax(logits[b], dim=0), dim=0)
        want = O.seg_postprocess(pred)
        assert {c for c in range(1, n_cls) if det[b, c, 0]} == set(want.keys())
        for c, mask in want.items():
            assert np.array_equal(np.where(objmap[b] == c, 255, 0).astype(np.uint8), mask), "class %d" % c


def test_bbox_golden_rects():
    """get_bbox goldens from the reference (touching every border, exact multiples of 40) through the HIP bbox kernel."""
    from autoposeestimation_amd import engine as E
    from conftest import golden
    g = golden("bbox")
    blob = np.unpackbits(g["blob"]).reshape(H, W).astype(bool)
    labs = []
    for rect in g["rects"]:
        lab = blob if rect[0] < 0 else np.zeros((H, W), bool)
        if rect[0] >= 0:
            lab = lab.copy()
            lab[rect[0]:rect[1] + 1, rect[2]:rect[3] + 1] = True
        labs.append(lab)
    label = torch.from_numpy(np.stack(labs).astype(np.uint8)).cuda()
    score = torch.full(label.shape, 0.5, device="cuda")
    _, det = E.seg_components(label, score, 2, 0)
    det = det.cpu().numpy()
    for i, box in enumerate(g["boxes"]):
        assert det[i, 1, 0] == 1 and list(det[i, 1, 1:]) == list(box), i


def test_choose_backproject_normalise_bit_exact():
    from autoposeestimation_amd import engine as E
    meta = S.REALSENSE_META
    frames = [S.synthetic_frame(31, cls=3, box=(150, 250), size=(150, 150)),     # 22 500 candidates > N
              S.synthetic_frame(32, cls=5, box=(10, 20), size=(25, 30)),         # 750 candidates < N  -> wrap
              S.synthetic_frame(33, cls=2, box=(400, 500), size=(75, 120))]
    rgb = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    depth_np = np.stack([f[1] for f in frames])
    depth_np[1, 10:35, 20:50][::2] = 0
    depth = torch.from_numpy(depth_np).cuda()
    objmap = torch.from_numpy(np.stack([f[2] for f in frames])).cuda()
    objects = []
    for b, f in enumerate(frames):
        cls = int(f[2].max())
        objects.append((b, cls) + O.get_bbox(f[2] == cls, H, W))
    n = 1000
    for o in objects:       # one launch per bucket in the pipeline; here one per object
        objs = torch.tensor([o], dtype=torch.int32).cuda()
        choose, n_cand = E.choose_points(objmap, depth, objs, n, seed=7)
        b, cls, rmin, rmax, cmin, cmax = o
        m = (frames[b][2] == cls) * (depth_np[b] != 0)
        nz = m[rmin:rmax, cmin:cmax].flatten().nonzero()[0]
        assert int(n_cand[0]) == len(nz)
        ch = choose[0].cpu().numpy()
        if len(nz) <= n:
            assert np.array_equal(ch, np.pad(nz, (0, n - len(nz)), "wrap"))
        else:   # ordered subset of the candidates, no repeats
            assert np.all(np.diff(ch) > 0) and np.isin(ch, nz).all() and len(ch) == n
        pts = E.backproject(depth, objs, choose, meta["intr"], meta["depth_scale"]).cpu().numpy()
        want = O.backproject(depth_np[b], ch, rmin, rmax, cmin, cmax, meta)
        assert np.array_equal(pts[0, :, :3], want) and not pts[0, :, 3].any()
        rects = objs[:, [0, 2, 4]].contiguous()
        img4 = E.preprocess_u8(rgb, rects, rmax - rmin, cmax - cmin, div255=False).cpu()
        want_img = O.crop_image(frames[b][0], rmin, rmax, cmin, cmax)[0].permute(1, 2, 0)
        assert torch.equal(img4[0, :, :, :3], want_img)
    full = E.preprocess_u8(rgb, torch.tensor([[0, 0, 0], [1, 0, 0], [2, 0, 0]], dtype=torch.int32).cuda(), H, W, div255=True).cpu()
    for b in range(3):
        assert torch.equal(full[b, :, :, :3], O.seg_input(frames[b][0])[0].permute(1, 2, 0))


def test_empty_depth_object_is_dropped():
    from autoposeestimation_amd import engine as E
    rgb, depth, label = S.synthetic_frame(40, cls=1)
    depth[label == 1] = 0
    objs = torch.tensor([(0, 1) + O.get_bbox(label == 1, H, W)], dtype=torch.int32).cuda()
    choose, n_cand = E.choose_points(torch.from_numpy(label[None]).cuda(), torch.from_numpy(depth[None]).cuda(), objs, 1000)
    assert int(n_cand[0]) == 0 and not choose.any()


def test_relabel_trust_checks_vs_numpy():
    """create_labels.py:101-205 on the device: target-class component + the three trust checks, against numpy."""
    from autoposeestimation_amd import engine as E
    from autoposeestimation_amd.label_generator.create_labels import relabel_frames

    class _FakeSeg:             # injects logits so the comparison is integer-exact
        classes = 13

        def __init__(self, logits):
            self.l = logits

        def logits_nhwc(self, x4):
            return self.l

    B = 4
    logits = torch.stack([_blob_logits(50 + i, 13) for i in range(B)])
    rng = np.random.default_rng(3)
    depth = rng.integers(300, 1100, (B, H, W)).astype(np.uint16)
    depth[rng.random((B, H, W)) < 0.05] = 0
    depth[2] = 2000                                   # outside the gate everywhere -> "no depth overlap"
    bs = np.zeros((B, H, W), np.uint8)
    bs[0, 100:200, 100:300] = 255
    bs[1, 0:5, 0:5] = 255
    r2c = np.tile(np.eye(4), (B, 1, 1))
    r2c[:, :3, 3] = [0, 0, 700]
    ref = np.array([0.0, 0.0, 0.0])
    for cls in (0, 2):
        labels, save, stats = relabel_frames(_FakeSeg(logits.permute(0, 2, 3, 1).contiguous().cuda()),
                                             torch.zeros(B, H, W, 3, dtype=torch.uint8).cuda(), torch.from_numpy(depth).cuda(), r2c,
                                             ref, cls, torch.from_numpy(bs).cuda())
        labels = labels.cpu().numpy()
        for i in range(B):
            pred = F.softmax(F.softmax(logits[i], 0), 0)
            want = O.seg_postprocess(pred, min_pixels=-1).get(cls + 1, np.zeros((H, W), np.uint8))
            d = depth[i].astype(np.float64)
            d[d > 850] = 0
            d[d < 550] = 0
            if len(np.unique(want[bs[i] != 0])) <= 1:
                exp_save, exp = True, bs[i]
            elif len(np.unique(want[d != 0])) <= 1:
                exp_save, exp = False, want
            else:
                exp_save, exp = len(np.unique(want[30:H - 30, 50:W - 50])) > 1, want
            assert bool(save[i]) == exp_save, (cls, i)
            assert np.array_equal(labels[i], exp), (cls, i)


def test_fused_seg_head_matches_conv_plus_argmax():
    from autoposeestimation_amd import engine as E
    torch.manual_seed(0)
    feat = torch.randn(2, 96, 128, 64, device="cuda") * 2
    w = torch.randn(13, 64, device="cuda") / 4
    b = torch.randn(13, device="cuda")
    label, score = E.seg_head(feat, w, b, double_softmax=True)
    logits = (feat.double() @ w.double().t() + b.double())
    p2 = torch.softmax(torch.softmax(logits, -1), -1)
    top2 = logits.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert torch.equal(label[safe].long(), logits.argmax(-1)[safe])
    np.testing.assert_allclose(score[safe].cpu().numpy(), p2.max(-1).values[safe].float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    # and against the unfused device path
    conv = E.Conv(w, b, device="cuda")
    label2, score2 = E.seg_argmax(conv(feat), 13, double_softmax=True)
    assert (label != label2).sum().item() <= 2
    assert (score - score2).abs().max().item() < 1e-5


def test_argmax_breaks_probability_ties_like_torch():
    """pipeline/utils.py:430-435 takes the arg-max of the float32 PROBABILITIES: two logits closer than the rounding of exp() give equal
    probabilities and torch.argmax returns the lower class -- also when the higher class has the (marginally) larger logit."""
    from autoposeestimation_amd import engine as E
    n, C = 64, 5
    logits = torch.full((1, 1, n, 8), -4.0)
    hi = float(torch.nextafter(torch.tensor(0.1), torch.tensor(1.0)))               # one ulp (7.5e-9) above 0.1: exp(-7.5e-9) rounds to 1.0f
    logits[0, 0, :, 1] = 0.1
    logits[0, 0, :, 3] = hi
    logits[0, 0, n // 2:, 3] = 0.6                                                  # a real margin on the second half
    p = F.softmax(F.softmax(logits[..., :C], -1), -1)
    want = p.argmax(-1)[0, 0]
    assert int(want[0]) == 1 and int(want[-1]) == 3                                 # the premise: torch sees a tie on the first half
    label, score = E.seg_argmax(logits.cuda(), C, double_softmax=True)
    assert torch.equal(label[0, 0].cpu().long(), want)
    # the fused head: 64 -> C contraction with weights that reproduce these logits from a one-hot-ish feature
    feat = torch.zeros(1, 1, n, 64)
    feat[..., 0] = 1.0
    w = torch.zeros(C, 64)
    w[:, 0] = torch.tensor([-4.0, 0.1, -4.0, 0.0, -4.0])
    b = torch.zeros(C)
    w[3, 0] = hi
    feat[0, 0, n // 2:, 1] = 1.0
    w[3, 1] = 0.5
    l2, _ = E.seg_head(feat.cuda(), w.cuda().contiguous(), b.cuda(), True)
    assert torch.equal(l2[0, 0].cpu().long(), want)


def test_argmax_ties_of_the_second_softmax_resolve_like_torch():
    """The live path applies softmax twice (pipeline/utils.py:429-435) and torch.argmax sees p2 = softmax(softmax(l)).  Logit gaps of
    5..10 ulp (3.9e-8 .. 7.6e-8 at 0.1) are NOT ties after the first softmax (p1 differs by one ulp, a one-softmax arg-max returns the
    higher class 3) but round to equal p2 -- torch returns the lower class 1.  Both device arg-max forms (ape_seg_argmax_f32 and the
    fused 64 -> C head) must follow; the band's outer edge (12 ulp here) depends on the last bit of p1 and is not asserted."""
    from autoposeestimation_amd import engine as E
    C = 5
    x = torch.tensor(0.1)
    for k in range(1, 11):
        x = torch.nextafter(x, torch.tensor(1.0))
        logits = torch.full((1, 1, 32, 8), -4.0)
        logits[..., 1] = 0.1
        logits[..., 3] = float(x)
        p1 = F.softmax(logits[..., :C], -1)
        want = int(F.softmax(p1, -1).argmax(-1)[0, 0, 0])
        assert want == 1
        if k >= 5:
            assert int(p1.argmax(-1)[0, 0, 0]) == 3                       # the premise: only the SECOND softmax ties
        lab, _ = E.seg_argmax(logits.cuda(), C, double_softmax=True)
        assert int(lab.max()) == 1 and int(lab.min()) == 1, k
        feat = torch.zeros(1, 1, 32, 64)
        feat[..., 0] = 1.0
        w = torch.zeros(C, 64)
        w[:, 0] = torch.tensor([-4.0, 0.1, -4.0, float(x), -4.0])
        l2, _ = E.seg_head(feat.cuda(), w.cuda().contiguous(), torch.zeros(C).cuda(), True)
        assert int(l2.max()) == 1 and int(l2.min()) == 1, k
