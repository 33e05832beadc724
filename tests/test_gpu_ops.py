"""HBM-bound glue kernels against plain PyTorch fp32 references of the same op (and against each other where two device
paths must agree bit for bit)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 60, 80, 64), (3, 20, 20, 32), (1, 7, 13, 8)])
def test_psp_prior_sum_equals_four_accumulating_resizes(shape):
    """ape_psp_prior_sum_f32 == sum over s in (1,2,3,6) of F.upsample(prior_s, (h,w), 'bilinear') (pspnet.py:22), bit for bit
    against four accumulating ape_bilinear_nhwc_f32 passes and within fp32 rounding of torch."""
    from autoposeestimation_amd import engine as E
    b, h, w, c = shape
    g = torch.Generator().manual_seed(b * 100 + h)
    zs = [torch.randn(b, s, s, c, generator=g) for s in (1, 2, 3, 6)]
    zd = [z.cuda() for z in zs]
    got = E.psp_prior_sum(zd, h, w)
    acc = torch.zeros(b, h, w, c, device="cuda")
    for z in zd:
        E.bilinear(z, h, w, False, out=acc, accumulate=True)
    assert torch.equal(got, acc)
    want = sum(F.interpolate(z.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=False) for z in zs)
    assert (got.cpu().permute(0, 3, 1, 2) - want).abs().max().item() <= 1e-5


@pytest.mark.parametrize("shape", [(3, 1000, 1024), (2, 500, 128), (2, 77, 12), (1, 64, 256), (2, 1000, 6)])
def test_mean_rows(shape):
    """nn.AvgPool1d(num_points) (network.py:51,65,149,166): both the 16-wave float4 kernel and the dword fallback."""
    from autoposeestimation_amd import engine as E
    b, n, c = shape
    x = torch.randn(b, n, c, generator=torch.Generator().manual_seed(n))
    got = E.mean_rows(x.cuda()).cpu()
    assert (got.view(b, c) - x.mean(1)).abs().max().item() <= 2e-6


@pytest.mark.parametrize("shape", [(2, 20, 20, 64, 128), (1, 9, 13, 32, 160), (2, 6, 50, 32, 256), (1, 5, 9, 16, 8), (2, 20, 20, 64, 32)])
def test_upconv_matches_torch(shape):
    """PSPUpsample (pspnet.py:27-37: Upsample x2 align_corners=True -> Conv2d 3x3 pad 1 -> PReLU) as low-resolution channel
    mixing + tap gather, against torch."""
    from autoposeestimation_amd import engine as E
    b, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(h * 10 + w)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    want = F.prelu(F.conv2d(up, wt, bias, padding=1), torch.tensor([0.25]))
    op = E.UpConv(wt, bias, 0.25, device="cuda", precision="f32")
    got = op(x.permute(0, 2, 3, 1).contiguous().cuda())
    err = (got.cpu().permute(0, 3, 1, 2) - want).abs().max().item() / max(1.0, want.abs().max().item())
    assert err <= 5e-6, err


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 50, 70), (3, 160, 160), (1, 17, 9), (1, 480, 640)])
def test_fused_stem_conv_pool(shape, precision):
    """Conv2d(3->64, 7x7, s2, p3) + ReLU + MaxPool2d(3, 2, 1) in one kernel (extractors.py:82-85,111-117) against torch and
    against the two-launch path it replaces."""
    from autoposeestimation_amd import engine as E
    b, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(b, 3, h, w, generator=g) * 2
    wt = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    want = F.max_pool2d(F.relu(F.conv2d(x, wt, None, 2, 3)), 3, 2, 1)
    conv = E.Conv(wt, None, 2, 3, 1, E.ACT_RELU, device="cuda", precision=precision)
    x4 = torch.zeros(b, h, w, 4, device="cuda")
    x4[..., :3] = x.permute(0, 2, 3, 1).cuda()
    old = E.USE_FUSED_STEM
    try:
        E.USE_FUSED_STEM = True
        got = E.stem_pool(conv, x4)
        E.USE_FUSED_STEM = False
        two = E.stem_pool(conv, x4)
    finally:
        E.USE_FUSED_STEM = old
    assert got.shape == two.shape == (b,) + tuple(want.shape[2:]) + (64,)
    tol = {"bf16x3": 5e-5, "bf16": 2e-2}[precision]
    scale = max(1.0, want.abs().max().item())
    assert (got.cpu().permute(0, 3, 1, 2) - want).abs().max().item() / scale <= tol
    assert (got - two).abs().max().item() / scale <= (2e-6 if precision == "bf16x3" else 2e-2)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_up3_at_chosen_pixels_equals_full_conv_gather(precision):
    """Upsample(x2, align_corners=True) + Conv2d 3x3 + PReLU evaluated only at chosen pixels (patch gather + one contraction)
    == the full-resolution convolution gathered at those pixels (pspnet.py:30-33 + network.py:100-102)."""
    from autoposeestimation_amd import engine as E
    b, h, w, c, n = 3, 20, 24, 64, 333
    g = torch.Generator().manual_seed(5)
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(64, c, 3, 3, generator=g) / (9 * c) ** 0.5
    bias = torch.randn(64, generator=g)
    choose = torch.stack([torch.randperm(4 * h * w, generator=g)[:n] for _ in range(b)])
    choose[0, :4] = torch.tensor([0, 2 * w - 1, (2 * h - 1) * 2 * w, 4 * h * w - 1])          # the four image corners
    full = F.prelu(F.conv2d(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True), wt, bias, padding=1), torch.tensor([0.25]))
    want = torch.gather(full.reshape(b, 64, -1), 2, choose[:, None, :].expand(b, 64, n)).permute(0, 2, 1)
    conv = E.Conv(wt, bias, 1, 1, 1, E.ACT_PRELU, alpha=0.25, device="cuda", precision=precision)
    patches = E.ups_patch_gather(x.permute(0, 2, 3, 1).contiguous().cuda(), choose.cuda())
    assert patches.shape == (b * n, 1, 1, 9 * c)
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    cols = F.unfold(up, 3, padding=1).view(b, c, 9, -1)                    # [b, c, tap, pixel]
    wantp = torch.gather(cols, 3, choose[:, None, None, :].expand(b, c, 9, n)).permute(0, 3, 2, 1).reshape(b * n, 9 * c)
    assert (patches.view(b * n, -1).cpu() - wantp).abs().max().item() <= 2e-6 * wantp.abs().max().item()
    got = E.conv3x3_as_matrix(conv)(patches).view(b, n, 64).cpu()
    tol = {"f32": 2e-6, "bf16x3": 5e-5}[precision]
    assert (got - want).abs().max().item() / want.abs().max().item() <= tol


@pytest.mark.parametrize("shape", [(2, 60, 80, 64), (3, 20, 20, 32), (1, 7, 13, 8), (2, 30, 40, 512), (1, 97, 131, 16)])
def test_adaptive_avgpool_multi(shape):
    """nn.AdaptiveAvgPool2d for the PSP sizes 2, 3, 6 (pspnet.py:15) in one pass (atom sums) against torch, including geometries
    whose bin edges overlap and ones that fall back to the per-size kernel."""
    from autoposeestimation_amd import engine as E
    b, h, w, c = shape
    x = torch.randn(b, c, h, w, generator=torch.Generator().manual_seed(h * w))
    got = E.adaptive_avgpool_multi(x.permute(0, 2, 3, 1).contiguous().cuda(), (2, 3, 6))
    for s in (2, 3, 6):
        want = F.adaptive_avg_pool2d(x, s).permute(0, 2, 3, 1)
        assert got[s].shape == want.shape
        assert (got[s].cpu() - want).abs().max().item() <= 2e-6


@pytest.mark.parametrize("s32", [False, True])
@pytest.mark.parametrize("shape", [(2, 60, 80, 576, 512), (3, 20, 20, 96, 64), (1, 9, 14, 64, 32)])
def test_adaptive_avgpool_multi_of_the_leading_channels(shape, s32):
    """ape_adaptive_avgpool_multi_nhwc_ld: the first C channels of a map with ldx channels per pixel (the PSP feature map with the folded
    form's 64 spare channels behind it), fp32 and pre-split input: bit for bit what pooling a contiguous copy of those channels gives,
    whatever the trailing channels hold (NaN here)"""
    from autoposeestimation_amd import engine as E
    b, h, w, ld, c = shape
    x = torch.randn(b, h, w, ld, generator=torch.Generator().manual_seed(ld * h)).cuda()
    lead = x[..., :c].contiguous()
    if s32:
        xs, ls = E.S32.from_f32(x), E.S32.from_f32(lead)
        xs.t[..., c:] = float("nan")               # (the trailing chunks of the pre-split image)
        got, want = E.adaptive_avgpool_multi(xs, (2, 3, 6), channels=c), E.adaptive_avgpool_multi(ls, (2, 3, 6))
    else:
        x[..., c:] = float("nan")
        got, want = E.adaptive_avgpool_multi(x, (2, 3, 6), channels=c), E.adaptive_avgpool_multi(lead, (2, 3, 6))
    for s in (2, 3, 6):
        assert got[s].shape == (b, s, s, c) and torch.equal(got[s], want[s])
        ref = F.adaptive_avg_pool2d(lead.permute(0, 3, 1, 2).cpu(), s).permute(0, 2, 3, 1)
        assert (got[s].cpu() - ref).abs().max().item() <= (2e-4 if s32 else 2e-6)


@pytest.mark.parametrize("div255", [True, False])
@pytest.mark.parametrize("geom", [(3, 96, 128, None), (2, 480, 640, None), (5, 160, 160, "crops"), (4, 80, 120, "crops"), (3, 41, 57, "crops")])
def test_stem_on_uint8_frames_equals_preprocess_then_stem_bitwise(geom, div255):
    """ape_stem_conv_pool_u8 (ToTensor / Normalize fused into the stem's patch load; pipeline/utils.py:421-427, 556-560 + extractors.py:82-85,
    111-117) against ape_preprocess_u8_nhwc4 -> ape_stem_conv_pool_bf16: whole frames and crops (windows at the frame's corners and in its
    middle, odd sizes), bit for bit"""
    from autoposeestimation_amd import engine as E
    n, hc, wc, mode = geom
    g = torch.Generator().manual_seed(hc * 1000 + wc)
    F_, H_, W_ = (n, hc, wc) if mode is None else (3, 480, 640)
    rgb = torch.randint(0, 256, (F_, H_, W_, 3), generator=g, dtype=torch.uint8).cuda()
    if mode is None:
        rects = torch.zeros(n, 3, dtype=torch.int32)
        rects[:, 0] = torch.arange(n)
    else:
        corners = [(0, 0), (H_ - hc, W_ - wc), (0, W_ - wc), (H_ - hc, 0), (H_ // 3, W_ // 2 - wc // 2)]
        rects = torch.tensor([[i % F_, *corners[i % len(corners)]] for i in range(n)], dtype=torch.int32)
    rects = rects.cuda()
    conv = E.Conv(torch.randn(64, 3, 7, 7, generator=g) / 12, None, 2, 3, 1, E.ACT_RELU, device="cuda", precision="bf16x3")
    want = E.stem_pool(conv, E.preprocess_u8(rgb, rects, hc, wc, div255))
    got = E.stem_pool(conv, E.U8Frames(rgb, rects, hc, wc, div255))
    assert got.shape == want.shape and torch.equal(got, want)
    assert E.U8Frames(rgb, rects, hc, wc, div255)._x4 is None          # (nothing was materialised on the way)
