"""Training step (DenseFusion/tools/train.py:205-238): every backward kernel against torch autograd of
the same op on the CPU (float64), then the whole estimator / refiner step against autograd through oracle/densefusion_oracle.py
(whose forward is pinned to the reference modules by tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import densefusion_oracle as DO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,h,w,act,res,bias", [
    (3, 64, 7, 2, 3, 1, 40, 48, "relu", False, False),      # stem (input gradient not needed but checked)
    (64, 64, 3, 1, 1, 1, 20, 24, "relu", True, False),      # BasicBlock conv2 with residual
    (64, 128, 3, 2, 1, 1, 20, 24, "relu", False, False),    # layer2.0.conv1 (stride 2)
    (64, 128, 3, 2, 1, 1, 19, 23, "none", False, True),     # odd sizes: rows the stride skips get their own gradient
    (64, 128, 1, 2, 0, 1, 20, 24, "none", False, False),    # downsample
    (128, 128, 3, 1, 2, 2, 10, 12, "relu", True, False),    # dilation 2
    (128, 64, 3, 1, 4, 4, 10, 12, "none", False, True),     # dilation 4
    (384, 640, 1, 1, 0, 1, 1, 333, "relu", False, True),    # conv1d over points
    (128, 4, 1, 1, 0, 1, 1, 100, "sigmoid", False, True),   # selected head rows
    (1024, 512, 1, 1, 0, 1, 1, 1, "relu", False, True),     # refiner Linear
])
def test_conv_backward(cin, cout, k, stride, pad, dil, h, w, act, res, bias):
    from autoposeestimation_amd import autograd as A, engine as E
    g = torch.Generator().manual_seed(cin * 7 + cout + k + h)
    x = torch.randn(2, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1 if bias else None
    ho, wo = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
    r = torch.randn(2, cout, ho, wo, generator=g) if res else None
    gy = torch.randn(2, cout, ho, wo, generator=g)
    # float64 CPU reference
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    br = b.double().requires_grad_() if bias else None
    rr = r.double().requires_grad_() if res else None
    y = F.conv2d(xr, wr, br, stride, pad, dil)
    if res:
        y = y + rr
    y = {"relu": F.relu, "sigmoid": torch.sigmoid, "none": lambda t: t}[act](y)
    y.backward(gy.double())
    # device
    c4 = (cin + 3) // 4 * 4
    xd = torch.zeros(2, h, w, c4, device=DEV)
    xd[..., :cin] = _nhwc(x).to(DEV)
    xd.requires_grad_()
    wd = wt.to(DEV).requires_grad_()
    bd = b.to(DEV).requires_grad_() if bias else None
    rd = _nhwc(r).to(DEV).requires_grad_() if res else None
    yd = A.conv(xd, wd, bd, rd, stride, pad, dil, {"relu": E.ACT_RELU, "sigmoid": E.ACT_SIGMOID, "none": E.ACT_NONE}[act])
    assert _rel(_nchw(yd), y) < 2e-5
    yd.backward(_nhwc(gy).to(DEV))
    assert _rel(wd.grad, wr.grad) < 2e-5
    assert _rel(_nchw(xd.grad[..., :cin]), xr.grad) < 2e-5
    assert not xd.grad[..., cin:].any()
    if bias:
        assert _rel(bd.grad, br.grad) < 2e-5
    if res:
        assert _rel(_nchw(rd.grad), rr.grad) < 2e-5


def test_conv1d_and_linear_weight_layouts():
    from autoposeestimation_amd import autograd as A, engine as E
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 50, 1, 32, generator=g)
    w3 = torch.randn(64, 32, 1, generator=g)                 # nn.Conv1d weight
    w2 = torch.randn(16, 64, generator=g)                    # nn.Linear weight
    xd, w3d, w2d = x.to(DEV), w3.to(DEV).requires_grad_(), w2.to(DEV).requires_grad_()
    y = A.conv(A.conv(xd, w3d, act=E.ACT_RELU), w2d)
    y.sum().backward()
    w3r, w2r = w3.double().requires_grad_(), w2.double().requires_grad_()
    yr = F.linear(F.relu(F.linear(x.double().view(50, 32), w3r[:, :, 0])), w2r)
    yr.sum().backward()
    assert w3d.grad.shape == w3.shape and w2d.grad.shape == w2.shape
    assert _rel(w3d.grad, w3r.grad) < 2e-5 and _rel(w2d.grad, w2r.grad) < 2e-5


def test_maxpool_backward_with_ties():
    from autoposeestimation_amd import autograd as A
    g = torch.Generator().manual_seed(2)
    for h, w in ((20, 24), (19, 23)):
        x = F.relu(torch.randn(2, 8, h, w, generator=g))      # zeros tie inside many windows, like the post-ReLU stem
        x[0, :, 4:9, 4:9] = 1.5                               # a plateau: first-maximum rule decides
        xr = x.double().requires_grad_()
        yr = F.max_pool2d(xr, 3, 2, 1)
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy.double())
        xd = _nhwc(x).to(DEV).requires_grad_()
        yd = A.MaxPoolFn.apply(xd)
        yd.backward(_nhwc(gy).to(DEV))
        assert torch.equal(_nchw(yd).cpu(), yr.float())
        assert _rel(_nchw(xd.grad), xr.grad) < 1e-6


@pytest.mark.parametrize("h,w", [(20, 20), (15, 20), (5, 7)])
def test_adaptive_avgpool_backward(h, w):
    from autoposeestimation_amd import autograd as A
    g = torch.Generator().manual_seed(3)
    for s in (1, 2, 3, 6):
        x = torch.randn(2, 8, h, w, generator=g)
        xr = x.double().requires_grad_()
        yr = F.adaptive_avg_pool2d(xr, (s, s))
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy.double())
        xd = _nhwc(x).to(DEV).requires_grad_()
        yd = A.AdaptiveAvgPoolFn.apply(xd, s)
        yd.backward(_nhwc(gy).to(DEV))
        assert _rel(_nchw(yd), yr) < 1e-5 and _rel(_nchw(xd.grad), xr.grad) < 1e-5


@pytest.mark.parametrize("h,w,ho,wo,ac", [(10, 12, 20, 24, True), (3, 3, 10, 12, False), (6, 6, 15, 20, False), (1, 1, 10, 10, False),
                                          (2, 2, 20, 20, False)])
def test_bilinear_backward(h, w, ho, wo, ac):
    from autoposeestimation_amd import autograd as A
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 8, h, w, generator=g)
    xr = x.double().requires_grad_()
    yr = F.interpolate(xr, size=(ho, wo), mode="bilinear", align_corners=ac)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xd = _nhwc(x).to(DEV).requires_grad_()
    yd = A.BilinearFn.apply(xd, ho, wo, ac)
    yd.backward(_nhwc(gy).to(DEV))
    assert _rel(_nchw(yd), yr) < 1e-5 and _rel(_nchw(xd.grad), xr.grad) < 1e-5


def test_prelu_logsoftmax_gather_mean_backward():
    from autoposeestimation_amd import autograd as A
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 9, 11, 16, generator=g)
    a = torch.tensor([0.25])
    gy = torch.randn(x.shape, generator=g)
    xr, ar = x.double().requires_grad_(), a.double().requires_grad_()
    F.prelu(xr, ar).backward(gy.double())
    xd, ad = x.to(DEV).requires_grad_(), a.to(DEV).requires_grad_()
    A.PReLUFn.apply(xd, ad).backward(gy.to(DEV))
    assert _rel(xd.grad, xr.grad) < 1e-6 and _rel(ad.grad, ar.grad) < 1e-5
    # log-softmax over the last axis
    xr = x.double().requires_grad_()
    F.log_softmax(xr, dim=3).backward(gy.double())
    xd = x.to(DEV).requires_grad_()
    A.LogSoftmaxRowsFn.apply(xd).backward(gy.to(DEV))
    assert _rel(xd.grad, xr.grad) < 1e-5
    # gather rows with repeated indices (wrap-padded `choose`)
    e = torch.randn(1, 120, 32, generator=g)
    idx = torch.cat([torch.arange(0, 120, 3), torch.arange(0, 120, 3)[:25]])[None]
    ge = torch.randn(1, idx.shape[1], 32, generator=g)
    er = e.double().requires_grad_()
    torch.gather(er, 1, idx[:, :, None].expand(-1, -1, 32)).backward(ge.double())
    ed = e.to(DEV).requires_grad_()
    A.GatherRowsFn.apply(ed, idx.to(DEV)).backward(ge.to(DEV))
    assert _rel(ed.grad, er.grad) < 1e-6
    # mean over points
    m = torch.randn(1, 77, 64, generator=g)
    gm = torch.randn(1, 64, generator=g)
    mr = m.double().requires_grad_()
    mr.mean(1).backward(gm.double())
    md = m.to(DEV).requires_grad_()
    A.MeanRowsFn.apply(md).backward(gm.to(DEV))
    assert _rel(md.grad, mr.grad) < 1e-6


def _loss_case(n, m, seed):
    g = torch.Generator().manual_seed(seed)
    r = torch.randn(1, n, 4, generator=g)
    t = torch.randn(1, n, 3, generator=g) * 0.05
    c = torch.rand(1, n, 1, generator=g) * 0.9 + 0.05
    points = torch.randn(1, n, 3, generator=g) * 0.1
    model = torch.randn(1, m, 3, generator=g) * 0.05
    target = model @ torch.linalg.qr(torch.randn(3, 3, generator=g))[0] + torch.tensor([0.02, -0.01, 0.4])
    return r, t, c, points, model, target


@pytest.mark.parametrize("sym", [False, True])
def test_loss_backward_matches_oracle_autograd(sym):
    from autoposeestimation_amd.DenseFusion.lib.loss import Loss
    n, m = 150, 120
    r, t, c, points, model, target = _loss_case(n, m, 6)
    idx = torch.tensor([[2]])
    sym_list = [2] if sym else []
    rr, tr, cr = (v.double().requires_grad_() for v in (r, t, c))
    loss_ref, dis_ref, _, _, _ = DO.loss_forward(rr, tr, cr, target.double(), model.double(), idx, points.double(), 0.015, False, m, sym_list)
    loss_ref.backward()
    rd, td, cd = (v.to(DEV).requires_grad_() for v in (r, t, c))
    loss, dis, new_points, new_target, pred = Loss(m, sym_list)(rd, td, cd, target.to(DEV), model.to(DEV), idx.to(DEV), points.to(DEV), 0.015, False)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-5 * max(1.0, abs(float(loss_ref.detach())))
    assert not dis.requires_grad and not new_points.requires_grad and not new_target.requires_grad
    (loss * 3.0).backward()                                  # a non-unit upstream gradient
    assert _rel(rd.grad, 3 * rr.grad) < 2e-4 and _rel(td.grad, 3 * tr.grad) < 2e-4 and _rel(cd.grad, 3 * cr.grad) < 2e-5


@pytest.mark.parametrize("sym", [False, True])
def test_refine_loss_backward_matches_oracle_autograd(sym):
    from autoposeestimation_amd.DenseFusion.lib.loss_refiner import Loss_refine
    m = 200
    r, t, _, points, model, target = _loss_case(1, m, 7)
    pts = torch.randn(1, 300, 3)
    idx = torch.tensor([[0]])
    sym_list = [0] if sym else []
    rr, tr = r[0].double().requires_grad_(), t[0].double().requires_grad_()
    dis_ref, _, _, _ = DO.loss_refine_forward(rr, tr, target.double(), model.double(), idx, pts.double(), m, sym_list)
    dis_ref.backward()
    rd, td = r[0].to(DEV).requires_grad_(), t[0].to(DEV).requires_grad_()
    dis, new_points, new_target, pred = Loss_refine(m, sym_list)(rd, td, target.to(DEV), model.to(DEV), idx.to(DEV), pts.to(DEV))
    assert abs(float(dis.detach()) - float(dis_ref.detach())) < 1e-6
    dis.backward()
    assert _rel(rd.grad, rr.grad) < 2e-4 and _rel(td.grad, tr.grad) < 2e-4
    assert not new_points.requires_grad


def _sample(n, hc, wc, num_obj, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(1, 3, hc, wc, generator=g)
    x = torch.randn(1, n, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 0.6])
    base = torch.randperm(hc * wc, generator=g)[: n - 20].sort()[0]
    choose = torch.cat([base, base[:20]]).view(1, 1, n)                   # wrap-padded like pipeline/utils.py:536-539
    obj = torch.tensor([[3]])
    return img, x, choose, obj


def _dropout_masks(seed, dev):
    g = torch.Generator().manual_seed(seed)
    mk = lambda c, p: (torch.bernoulli(torch.full((1, c), 1 - p), generator=g) / (1 - p)).to(dev)  # noqa: E731
    return {"drop_1": mk(1024, 0.3), "drop_2a": mk(256, 0.15), "drop_2b": mk(64, 0.15)}


def test_estimator_training_step_gradients_match_oracle():
    """train.py:216-225 with refine_start False: estimator(img, points, choose, idx) -> criterion -> loss.backward();
    all 77 parameter gradients vs torch autograd through the CPU oracle (same Dropout2d multipliers)."""
    from autoposeestimation_amd import synthetic as S
    from autoposeestimation_amd.DenseFusion.lib.loss import Loss
    from autoposeestimation_amd.DenseFusion.lib.network import PoseNet
    n, m, num_obj, hc, wc = 200, 150, 5, 40, 80
    sd = S.posenet_state_dict(num_obj, seed=5)
    img, x, choose, obj = _sample(n, hc, wc, num_obj, 8)
    _, _, _, _, model, target = _loss_case(n, m, 9)
    target = target + torch.tensor([0.0, 0.0, 0.2])
    masks = _dropout_masks(10, "cpu")
    # oracle
    sdr = {k: v.clone().float().requires_grad_() for k, v in sd.items()}
    pr, pt, pc, emb_r = DO.posenet_forward(sdr, img, x, choose, obj, num_obj, drop=masks)
    loss_ref, dis_ref, npts_ref, ntgt_ref, _ = DO.loss_forward(pr, pt, pc, target, model, obj, x, 0.015, False, m, [])
    loss_ref.backward()
    # device
    est = PoseNet(n, num_obj)
    est.load_state_dict(sd)
    est.to(DEV).train()
    est.set_dropout_masks({k: v.to(DEV) for k, v in masks.items()})
    r, t, c, emb = est(img.to(DEV), x.to(DEV), choose.to(DEV), obj.to(DEV))
    assert r.shape == (1, n, 4) and t.shape == (1, n, 3) and c.shape == (1, n, 1) and emb.shape == (1, 32, n)
    assert _rel(r, pr) < 1e-4 and _rel(t, pt) < 1e-4 and _rel(c, pc) < 1e-4 and _rel(emb, emb_r) < 1e-4
    loss, dis, new_points, new_target, _ = Loss(m, [])(r, t, c, target.to(DEV), model.to(DEV), obj.to(DEV), x.to(DEV), 0.015, False)
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))
    loss.backward()
    worst = {}
    for k, p in est.named_parameters():
        gref = sdr[k].grad
        if k.startswith("cnn.model.module.classifier"):
            assert p.grad is None and gref is None           # unused branch (pspnet.py:57-61 never reaches the loss)
            continue
        assert p.grad is not None, k
        assert p.grad.shape == p.shape
        if k.startswith("conv4_"):                           # only the selected object's rows receive gradient
            kk = {"r": 4, "t": 3, "c": 1}[k[6]]
            rows = torch.zeros(p.shape[0], dtype=torch.bool)
            rows[3 * kk:4 * kk] = True
            assert not p.grad.cpu()[~rows].any()
        worst[k] = _rel(p.grad, gref)
    bad = {k: v for k, v in worst.items() if v > 2e-3}
    assert not bad, bad
    assert len(worst) == 77 - 4


def test_training_forward_without_dropout_equals_eval_forward():
    from autoposeestimation_amd import synthetic as S
    from autoposeestimation_amd.DenseFusion.lib.network import PoseNet
    n, num_obj = 100, 5
    sd = S.posenet_state_dict(num_obj, seed=5)
    img, x, choose, obj = _sample(n, 40, 40, num_obj, 11)
    est = PoseNet(n, num_obj)
    est.load_state_dict(sd)
    est.to(DEV).eval()
    with torch.no_grad():
        ev = est(img.to(DEV), x.to(DEV), choose.to(DEV), obj.to(DEV))
    est.train()
    ones = {"drop_1": torch.ones(1, 1024), "drop_2a": torch.ones(1, 256), "drop_2b": torch.ones(1, 64)}
    tr = est.set_dropout_masks(ones)(img.to(DEV), x.to(DEV), choose.to(DEV), obj.to(DEV))
    for a, b in zip(tr, ev):
        assert _rel(a, b) < 1e-4
    # sampled dropout really drops channels and rescales
    est.set_dropout_masks(None)
    tr2 = est(img.to(DEV), x.to(DEV), choose.to(DEV), obj.to(DEV))
    assert _rel(tr2[0], ev[0]) > 1e-3


def test_refiner_training_step_and_adam_match_torch():
    """train.py:219-222 + :232-234 with refine_start True: refiner forward -> Loss_refine -> dis.backward() twice (iteration=2),
    then optimizer.step(): gradients and the updated parameters vs torch autograd + torch.optim.Adam on the CPU oracle."""
    from autoposeestimation_amd import synthetic as S
    from autoposeestimation_amd.autograd import Adam
    from autoposeestimation_amd.DenseFusion.lib.loss_refiner import Loss_refine
    from autoposeestimation_amd.DenseFusion.lib.network import PoseRefineNet
    n, m, num_obj = 300, 200, 5
    sd = S.refiner_state_dict(num_obj, seed=6)
    g = torch.Generator().manual_seed(12)
    emb = torch.randn(1, 32, n, generator=g)
    pts = torch.randn(1, n, 3, generator=g) * 0.05
    _, _, _, _, model, target = _loss_case(1, m, 13)
    target = target - torch.tensor([0.02, -0.01, 0.4]) + 0.01
    obj = torch.tensor([[1]])
    # oracle: two accumulated iterations, then Adam
    params = {k: v.clone().float().requires_grad_() for k, v in sd.items()}
    opt_ref = torch.optim.Adam(list(params.values()), lr=1e-3)
    np_ref, nt_ref = pts, target
    for _ in range(2):
        r, t = DO.refiner_forward(params, np_ref, emb, obj, num_obj)
        dis_ref, np_ref, nt_ref, _ = DO.loss_refine_forward(r, t, nt_ref, model, obj, np_ref, m, [])
        dis_ref.backward()
        np_ref, nt_ref = np_ref.detach(), nt_ref.detach()
    grads_ref = {k: v.grad.clone() for k, v in params.items()}
    opt_ref.step()
    # device
    ref = PoseRefineNet(n, num_obj)
    ref.load_state_dict(sd)
    ref.to(DEV).train()
    opt = Adam(ref.parameters(), lr=1e-3)
    opt.zero_grad()
    crit = Loss_refine(m, [])
    np_d, nt_d = pts.to(DEV), target.to(DEV)
    for _ in range(2):
        r, t = ref(np_d, emb.to(DEV), obj.to(DEV))
        assert r.shape == (1, 4) and t.shape == (1, 3)
        dis, np_d, nt_d, _ = crit(r, t, nt_d, model.to(DEV), obj.to(DEV), np_d)
        dis.backward()
    assert abs(float(dis.detach()) - float(dis_ref.detach())) < 1e-5
    for k, p in ref.named_parameters():
        assert _rel(p.grad, grads_ref[k]) < 2e-3, k
    opt.step()
    for k, p in ref.named_parameters():
        # Adam's first step is lr * g / (|g| + eps): entries whose gradient is ~eps amplify last-bit gradient differences
        d = (p.detach().cpu() - params[k].detach()).abs()
        big = grads_ref[k].abs() > 1e-5
        assert d.max() <= 2.001e-3 and (not big.any() or d[big].max() < 5e-6), k      # |step| <= lr everywhere
    # eval forward after the update uses the NEW weights (the cached plan is rebuilt)
    ref.eval()
    with torch.no_grad():
        r2, t2 = ref(pts.to(DEV), emb.to(DEV), obj.to(DEV))
    r_ref, t_ref = DO.refiner_forward({k: v.detach() for k, v in params.items()}, pts, emb, obj, num_obj)
    assert _rel(r2, r_ref) < 1e-4 and _rel(t2, t_ref) < 1e-4


def test_adam_kernel_matches_torch_over_steps():
    from autoposeestimation_amd.autograd import Adam
    g = torch.Generator().manual_seed(14)
    p0 = torch.randn(1000, generator=g)
    pr = p0.clone().requires_grad_()
    pd = p0.clone().to(DEV).requires_grad_()
    o_ref, o = torch.optim.Adam([pr], lr=3e-3), Adam([pd], lr=3e-3)
    for step in range(5):
        gr = torch.randn(1000, generator=g)
        pr.grad, pd.grad = gr.clone(), gr.clone().to(DEV)
        o_ref.step()
        o.step()
        assert (pd.detach().cpu() - pr.detach()).abs().max() < 1e-6, step
    o.zero_grad()
    assert pd.grad is None


def test_estimator_loss_decreases_over_adam_steps():
    """a few whole steps of train.py's loop on one sample: forward, loss, backward, Adam -- the loss goes down"""
    from autoposeestimation_amd import synthetic as S
    from autoposeestimation_amd.autograd import Adam
    from autoposeestimation_amd.DenseFusion.lib.loss import Loss
    from autoposeestimation_amd.DenseFusion.lib.network import PoseNet
    n, m, num_obj = 100, 100, 3
    est = PoseNet(n, num_obj)
    est.load_state_dict(S.posenet_state_dict(num_obj, seed=7))
    est.to(DEV).train()
    ones = {"drop_1": torch.ones(1, 1024), "drop_2a": torch.ones(1, 256), "drop_2b": torch.ones(1, 64)}
    est.set_dropout_masks(ones)
    img, x, choose, obj = (v.to(DEV) for v in _sample(n, 40, 40, num_obj, 15))
    obj = torch.tensor([[1]], device=DEV)
    _, _, _, _, model, target = _loss_case(n, m, 16)
    crit, opt = Loss(m, [1]), Adam(est.parameters(), lr=1e-4)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        r, t, c, _ = est(img, x, choose, obj)
        loss = crit(r, t, c, target.to(DEV), model.to(DEV), obj, x, 0.015, False)[0]
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_train_epoch_driver_cadence_and_progress():
    """DenseFusion/tools/train.py's loop on three synthetic samples: optimizer steps after `batch_size` samples and for the
    remainder; estimator phase then refiner phase; evaluation distance improves on the training samples."""
    from types import SimpleNamespace
    from autoposeestimation_amd import synthetic as S
    from autoposeestimation_amd.autograd import Adam
    from autoposeestimation_amd.DenseFusion.lib.loss import Loss
    from autoposeestimation_amd.DenseFusion.lib.loss_refiner import Loss_refine
    from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet
    from autoposeestimation_amd.DenseFusion.tools.train import evaluate, train_epoch
    n, m, num_obj = 100, 80, 3
    est, ref = PoseNet(n, num_obj), PoseRefineNet(n, num_obj)
    est.load_state_dict(S.posenet_state_dict(num_obj, seed=7))
    ref.load_state_dict(S.refiner_state_dict(num_obj, seed=8))
    est.to(DEV)
    ref.to(DEV)
    ones = {"drop_1": torch.ones(1, 1024), "drop_2a": torch.ones(1, 256), "drop_2b": torch.ones(1, 64)}
    est.set_dropout_masks(ones)
    data = []
    for i in range(3):
        img, x, choose, _ = _sample(n, 40, 40, num_obj, 20 + i)
        _, _, _, _, model, target = _loss_case(n, m, 30 + i)
        data.append((x, choose, img, target, model, torch.tensor([[i % num_obj]])))
    crit, crit_r = Loss(m, [1]), Loss_refine(m, [1])
    opt = SimpleNamespace(w=0.015, refine_start=False, iteration=2, batch_size=2, repeat_epoch=1)
    before = evaluate(est, ref, crit, crit_r, data, opt)
    optim = Adam(est.parameters(), lr=1e-4)
    stats = [train_epoch(est, ref, optim, crit, crit_r, data, opt) for _ in range(3)]
    assert stats[0]["samples"] == 3 and stats[0]["optimizer_steps"] == 2 and stats[0]["refiner_loss"] == 0.0
    assert stats[-1]["loss"] < stats[0]["loss"]
    assert evaluate(est, ref, crit, crit_r, data, opt) < before
    # refiner phase: the estimator is frozen in eval mode, only the refiner's parameters move
    opt.refine_start = True
    w_est = est.get_parameter("conv1_r.weight").detach().clone()
    w_ref = ref.get_parameter("conv1_r.weight").detach().clone()
    optim_r = Adam(ref.parameters(), lr=1e-4)
    rstats = [train_epoch(est, ref, optim_r, crit, crit_r, data, opt) for _ in range(3)]
    assert not est.training and ref.training
    assert torch.equal(est.get_parameter("conv1_r.weight").detach(), w_est)
    assert not torch.equal(ref.get_parameter("conv1_r.weight").detach(), w_ref)
    assert rstats[-1]["refiner_loss"] < rstats[0]["refiner_loss"]


def test_train_epoch_fed_by_the_train_mode_dataset(tmp_path):
    """train.py:96-101,190-238 as the reference wires it: PoseDataset('train', ..., add_noise=True) -> torch DataLoader(batch_size=1)
    -> the per-sample step with an optimizer step every `batch_size` samples, then the evaluation pass on PoseDataset('test')."""
    from types import SimpleNamespace
    from autoposeestimation_amd import synthetic as S
    from autoposeestimation_amd.autograd import Adam
    from autoposeestimation_amd.DenseFusion.datasets.myDatasetAugmented.dataset import PoseDataset
    from autoposeestimation_amd.DenseFusion.lib.loss import Loss
    from autoposeestimation_amd.DenseFusion.lib.loss_refiner import Loss_refine
    from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet
    from autoposeestimation_amd.DenseFusion.tools.train import evaluate, train_epoch
    root = str(tmp_path)
    S.pose_dataset_tree(root)
    n = 500
    train = PoseDataset("train", n, True, 0.03, False, "synth", root, p_extra_data=0.25, seed=1)
    test = PoseDataset("test", n, False, 0.0, False, "synth", root)
    assert len(train) == 10 and len(test) == 2
    loader = torch.utils.data.DataLoader(train, batch_size=1, shuffle=False, num_workers=0)
    test_loader = torch.utils.data.DataLoader([test[i][:6] for i in range(len(test))], batch_size=1, shuffle=False)
    est, ref = PoseNet(n, train.num_classes), PoseRefineNet(n, train.num_classes)
    est.load_state_dict(S.posenet_state_dict(train.num_classes, seed=7))
    ref.load_state_dict(S.refiner_state_dict(train.num_classes, seed=8))
    est.to(DEV)
    ref.to(DEV)
    crit = Loss(train.get_num_points_mesh(), train.get_sym_list())
    crit_r = Loss_refine(train.get_num_points_mesh(), train.get_sym_list())
    opt = SimpleNamespace(w=0.015, refine_start=False, iteration=2, batch_size=4, repeat_epoch=1)
    optim = Adam(est.parameters(), lr=1e-4)
    w0 = est.get_parameter("conv1_r.weight").detach().clone()
    st = train_epoch(est, ref, optim, crit, crit_r, loader, opt)
    assert st["samples"] == 10 and st["optimizer_steps"] == 3 and np.isfinite(st["loss"]) and np.isfinite(st["train_dis"])
    assert not torch.equal(est.get_parameter("conv1_r.weight").detach(), w0)
    assert np.isfinite(evaluate(est, ref, crit, crit_r, test_loader, opt))


@pytest.mark.parametrize("shape", [(64, 3, 7, 7), (128, 64, 3, 3), (640, 384, 1), (512, 1024), (5, 6, 3, 3)])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_weight_bank_operands_equal_the_per_call_packing_bitwise(shape, precision):
    """autograd.WeightBank (one launch packs the forward operand AND the flipped / transposed input-gradient operand straight from the
    parameter, kept until the optimizer moves it) against the round-2 path: engine.Conv packing the tensor / its flipped transpose per
    call.  Then the staleness rules: an in-place torch op on the parameter rebuilds the bank, Adam.step() refreshes it in place."""
    from autoposeestimation_amd import autograd as AG
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(5)
    w = torch.nn.Parameter(torch.randn(*shape, generator=g).to(DEV))
    bank = AG.weight_bank(w, precision)
    w4 = AG._w4(w.detach())

    def check(bank):
        for tr, ref_w in ((False, w4), (True, w4.flip(2, 3).permute(1, 0, 2, 3))):
            want = E.Conv(ref_w, device=DEV, precision=precision)
            got = bank.conv(tr, 1, 0, 1, E.ACT_NONE)
            assert torch.equal(got.w, want.w) and got.cin_real == want.cin_real and (got.cout, got.kh, got.kw, got.cin) == (want.cout, want.kh, want.kw, want.cin)
            if precision != "f32":
                assert torch.equal(got.wp.view(torch.int16), want.wp.view(torch.int16))
    check(bank)
    assert AG.weight_bank(w, precision) is bank                      # unchanged parameter: the same bank
    with torch.no_grad():
        w.mul_(1.5)                                                  # torch moved the parameter: stale
    bank2 = AG.weight_bank(w, precision)
    assert bank2 is not bank
    check(bank2)
    opt = AG.Adam([w], lr=1e-2)
    w.grad = torch.randn(*shape, generator=g).to(DEV)
    before = w.detach().clone()
    opt.step()
    assert not torch.equal(before, w.detach())
    assert AG.weight_bank(w, precision) is bank2                     # refreshed in place by the optimizer
    check(bank2)


def test_weight_bank_of_a_column_block_packs_in_place_and_follows_the_optimizer():
    """network.py:104-121 feeds conv1_r/t/c two COLUMN blocks of one [640, 1408, 1] weight (`W[:, :384]` on the point features,
    `W[:, 384:]` on the global feature): their banks are packed straight from the parameter's storage (row stride 1408, `src_ld`), kept on
    the parameter, and refreshed by Adam.step() together with the parameter's own bank"""
    from autoposeestimation_amd import autograd as AG
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(21)
    w = torch.nn.Parameter(torch.randn(640, 1408, 1, generator=g).to(DEV))
    w2 = w[:, :, 0]
    blocks = [w2[:, :384], w2[:, 384:]]

    def check():
        for blk in (w2[:, :384], w2[:, 384:]):
            bank = AG.weight_bank(blk, "bf16x3")
            assert bank.version is not None and bank.src.data_ptr() == blk.data_ptr()            # no dense copy was made
            dense = blk.detach().contiguous()
            for tr, ref_w in ((False, dense[:, :, None, None]), (True, dense.t().contiguous()[:, :, None, None])):
                want = E.Conv(ref_w, device=DEV, precision="bf16x3")
                got = bank.conv(tr, 1, 0, 1, E.ACT_NONE)
                assert torch.equal(got.w, want.w) and torch.equal(got.wp.view(torch.int16), want.wp.view(torch.int16))
    check()
    banks = [AG.weight_bank(b, "bf16x3") for b in blocks]
    assert [AG.weight_bank(b, "bf16x3") for b in (w2[:, :384], w2[:, 384:])] == banks           # fresh views find the kept banks
    opt = AG.Adam([w], lr=1e-2)
    w.grad = torch.randn(640, 1408, 1, generator=g).to(DEV)
    before = w.detach().clone()
    opt.step()
    assert not torch.equal(before, w.detach())
    assert [AG.weight_bank(b, "bf16x3") for b in (w2[:, :384], w2[:, 384:])] == banks           # refreshed in place
    check()


def test_training_forward_follows_writes_the_version_counter_does_not_see():
    """the kept conv operands (autograd.WeightBank) are keyed on the parameter's version counter; `p.data.mul_()`, a copy through `.data`
    or a replaced parameter do not move it: the training forward re-packs the module's banks first (sync_banks), so the outputs follow
    the parameters -- checked against a freshly built module holding the same values"""
    from autoposeestimation_amd import synthetic as S
    from autoposeestimation_amd.DenseFusion.lib.network import PoseRefineNet
    n, num_obj = 200, 4
    sd = S.refiner_state_dict(num_obj, seed=9)
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(1, 32, n, generator=g).to(DEV)
    pts = (torch.randn(1, n, 3, generator=g) * 0.05).to(DEV)
    obj = torch.tensor([[2]]).to(DEV)

    def fresh(state):
        m = PoseRefineNet(n, num_obj)
        m.load_state_dict(state)
        return m.to(DEV).train()

    net = fresh(sd)
    out0 = [t.detach().clone() for t in net(pts, emb, obj)]
    # (1) in-place through .data on every parameter
    with torch.no_grad():
        for p in net.parameters():
            v = p._version
            p.data.mul_(1.25)
            assert p._version == v                                    # the write is invisible to the counter
    out1 = [t.detach().clone() for t in net(pts, emb, obj)]
    want1 = [t.detach() for t in fresh({k: v * 1.25 for k, v in sd.items()})(pts, emb, obj)]
    assert all(torch.equal(a, b) for a, b in zip(out1, want1))
    assert not torch.equal(out1[0], out0[0])
    # (2) a copy through .data (what dist.broadcast(p.data) does)
    for k, p in net.named_parameters():
        p.data.copy_(sd[k].to(DEV))
    out2 = [t.detach() for t in net(pts, emb, obj)]
    assert all(torch.equal(a, b) for a, b in zip(out2, out0))
    # (3) a parameter replaced by assignment: the name -> parameter cache is dropped
    name, old = next((k, p) for k, p in net.named_parameters() if k.endswith("conv1_r.weight"))
    mod = net
    for part in name.split(".")[:-1]:
        mod = getattr(mod, part)
    setattr(mod, name.split(".")[-1], torch.nn.Parameter(old.detach() * 0.5))
    out3 = [t.detach() for t in net(pts, emb, obj)]
    sd3 = dict(sd)
    sd3[name] = sd[name] * 0.5
    want3 = [t.detach() for t in fresh(sd3)(pts, emb, obj)]
    assert all(torch.equal(a, b) for a, b in zip(out3, want3))


def test_multi_tensor_adam_equals_the_one_buffer_kernel_bitwise():
    """ape_adam_step_multi_f32 (all parameters in one launch, 64 per launch: 150 buffers = 3 launches) against ape_adam_step_f32 buffer by
    buffer, three steps, buffers of 1 .. 3 M elements, one without a gradient"""
    from autoposeestimation_amd import _lib
    from autoposeestimation_amd import autograd as AG
    g = torch.Generator().manual_seed(9)
    sizes = [1, 3, 64, 1000, 3_000_000] + [17 * (i + 1) for i in range(145)]
    ps = [torch.nn.Parameter(torch.randn(n, generator=g).to(DEV)) for n in sizes]
    ref = [p.detach().clone() for p in ps]
    m = [torch.zeros_like(r) for r in ref]
    v = [torch.zeros_like(r) for r in ref]
    opt = AG.Adam(ps, lr=3e-3, weight_decay=0.01)
    for step in range(1, 4):
        for i, p in enumerate(ps):
            p.grad = None if i == 7 else torch.randn(sizes[i], generator=g).to(DEV)
        opt.step()
        for i, p in enumerate(ps):
            if p.grad is None:
                continue
            rc = _lib.lib().ape_adam_step_f32(_lib.dptr(ref[i]), _lib.dptr(p.grad), _lib.dptr(m[i]), _lib.dptr(v[i]), sizes[i], 3e-3, 0.9, 0.999, 1e-8,
                                              step, 0.01, _lib.stream_ptr())
            assert rc == 0
    for p, r in zip(ps, ref):
        assert torch.equal(p.detach(), r)


@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,h,w,act,res,bias", [
    (512, 512, 3, 1, 4, 4, 20, 20, "relu", True, True),      # layer4 block conv at the training crop's 20 x 20 map: 16 tiles, 144 k-tiles
    (64, 128, 3, 2, 1, 1, 40, 40, "none", False, False),     # layer2.0.conv1 (stride 2): 18 k-tiles
    (256, 200, 3, 1, 2, 2, 20, 20, "prelu", False, True),    # ragged Cout
    (1024, 512, 1, 1, 0, 1, 1, 300, "relu", False, True),    # a pure 1x1 over points
])
def test_split_k_conv_matches_the_one_pass_kernel(cin, cout, k, stride, pad, dil, h, w, act, res, bias):
    """ape_conv_gemm_bf16_splitk (k-tiles dealt over workgroups, fixed-order second pass with bias / residual / activation) against
    ape_conv_gemm_bf16 on the same operands: same products, other summation order -> 2e-6 of the output scale; and it must actually
    split these shapes (workspace > 0) while leaving a chip-filling one alone"""
    import ctypes
    from autoposeestimation_amd import _lib
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, h, w, cin, generator=g).to(DEV)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    conv = E.Conv(wt, None, stride, pad, dil, {"none": E.ACT_NONE, "relu": E.ACT_RELU, "prelu": E.ACT_PRELU}[act], alpha=0.25, device=DEV, precision="bf16x3")
    ho, wo = conv.out_hw(h, w)
    r = torch.randn(1, ho, wo, cout, generator=g).to(DEV) if res else None
    b = torch.randn(cout, generator=g).to(DEV) if bias else None
    one = conv(x, residual=r, bias=b)
    two = conv(x, residual=r, bias=b, splitk=True)
    p = _lib.ConvParams(B=1, H=h, W=w, Cin=cin, ldx=cin, xoff=0, Ho=ho, Wo=wo, Cout=cout, ldy=cout, yoff=0, KH=k, KW=k, stride=stride, pad=pad, dil=dil,
                        act=0, alpha=0.0, bias_bstride=0, ldr=0, roff=0, ups=0)
    assert _lib.lib().ape_conv_gemm_splitk_workspace_bytes(ctypes.byref(p)) > 0
    assert not torch.equal(one, two) or cin * k * k <= 256
    assert float((one - two).abs().max()) <= 2e-6 * float(one.abs().max())
    p.B = 64
    assert _lib.lib().ape_conv_gemm_splitk_workspace_bytes(ctypes.byref(p)) == 0
