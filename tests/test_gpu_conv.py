"""Numerics of the implicit-GEMM conv kernels (exact-fp32 MFMA, split-bf16 x3, plain bf16) against a plain PyTorch fp32
CPU reference of the same op (F.conv2d + bias + residual + activation), across the geometries the PSPNet / PointNet
graphs use plus ragged edge cases.  Tolerances are relative to the output's max magnitude."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = {"f32": 2e-6, "bf16x3": 5e-5, "bf16": 2e-2}

CASES = [  # B, H, W, Cin, Cout, k, stride, pad, dil
    (2, 40, 40, 3, 64, 7, 2, 3, 1),      # stem (Cin padded to 4, K = 196: k-tail)
    (2, 20, 20, 64, 64, 3, 1, 1, 1),
    (1, 21, 19, 64, 128, 3, 2, 1, 1),    # odd sizes, stride 2
    (1, 20, 20, 128, 256, 3, 1, 2, 2),   # dilation 2
    (1, 12, 12, 256, 512, 3, 1, 4, 4),   # dilation 4
    (1, 9, 11, 64, 128, 1, 2, 0, 1),     # 1x1 stride-2 downsample
    (3, 5, 5, 512, 1024, 1, 1, 0, 1),
    (1, 1000, 1, 32, 64, 1, 1, 0, 1),    # PointNet 1x1
    (2, 333, 1, 384, 1920, 1, 1, 0, 1),  # ragged M, Cout not a tile multiple
    (1, 17, 23, 64, 33, 3, 1, 1, 1),     # Cout = 33
    (1, 8, 8, 8, 13, 1, 1, 0, 1),        # tiny Cout (segmentor final conv)
    (5, 1, 1, 1024, 512, 1, 1, 0, 1),    # Linear on B rows
    (1, 260, 256, 64, 288, 1, 1, 0, 1),  # M >= 65536 and Cout >= 256: the 256x256-tile variant, ragged in M and N
    (2, 32, 48, 64, 256, 3, 1, 1, 1),    # LDS-halo kernel (full 16x16 tiles), two N tiles
    (1, 30, 45, 32, 200, 3, 1, 2, 2),    # halo kernel, dilation 2, ragged tiles (94 % full) and ragged Cout
    (1, 48, 32, 96, 130, 3, 1, 4, 4),    # halo kernel, dilation 4
    (3, 16, 16, 64, 64, 3, 1, 1, 1),     # halo kernel, BN = 64 variant
]


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_conv_matches_torch(case, precision):
    from autoposeestimation_amd import engine as E
    b, h, w, cin, cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g) * 3
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    want = F.conv2d(x, wt, bias, stride, pad, dil)
    res = torch.randn(want.shape, generator=g)
    want = F.relu(want + res)
    conv = E.Conv(wt, bias, stride, pad, dil, E.ACT_RELU, device="cuda", precision=precision)
    cin4 = (cin + 3) // 4 * 4
    x4 = torch.zeros(b, h, w, cin4, device="cuda")
    x4[..., :cin] = x.permute(0, 2, 3, 1).cuda()
    got = conv(x4, residual=res.permute(0, 2, 3, 1).contiguous().cuda()).permute(0, 3, 1, 2).cpu()
    err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
    assert err <= TOL[precision], (case, precision, err)


GEMM_CASES = [  # B, H, W, Cin, Cout, k, stride, pad, dil
    (1, 300, 1, 64, 200, 1, 1, 0, 1),    # pure GEMM, ragged M and N
    (2, 64, 130, 32, 288, 1, 1, 0, 1),   # single k-tile (nk = 1), M = 16640
    (1, 23, 21, 96, 72, 3, 1, 1, 1),     # 3x3 with zero-filled taps, odd image, three 32-channel chunks
    (1, 12, 12, 256, 512, 3, 1, 4, 4),   # dilation 4 (crop layer4 geometry)
    (1, 9, 11, 64, 128, 1, 2, 0, 1),     # 1x1 stride 2 (not a pure GEMM)
]


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_kernel_every_block_shape(case, variant, precision):
    """conv_gemm.hip with each block shape forced (256x256 / 128x128 / 256x64 / 256x192), against F.conv2d and against the older
    conv_bf16.hip kernel (same operand rounding, different summation order)."""
    from autoposeestimation_amd import engine as E
    b, h, w, cin, cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + variant)
    x = torch.randn(b, cin, h, w, generator=g) * 3
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    want = F.conv2d(x, wt, bias, stride, pad, dil)
    res = torch.randn(want.shape, generator=g)
    want = F.relu(want + res)
    conv = E.Conv(wt, bias, stride, pad, dil, E.ACT_RELU, device="cuda", precision=precision)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    rd = res.permute(0, 2, 3, 1).contiguous().cuda()
    old = (E.USE_HALO_KERNEL, E.USE_GEMM_KERNEL, E.GEMM_VARIANT)
    try:
        E.USE_HALO_KERNEL, E.USE_GEMM_KERNEL, E.GEMM_VARIANT = False, True, variant
        got = conv(xd, residual=rd).permute(0, 3, 1, 2).cpu()
        E.USE_GEMM_KERNEL = False
        ref = conv(xd, residual=rd).permute(0, 3, 1, 2).cpu()
    finally:
        E.USE_HALO_KERNEL, E.USE_GEMM_KERNEL, E.GEMM_VARIANT = old
    scale = max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() / scale <= TOL[precision], (case, variant, precision)
    assert (got - ref).abs().max().item() / scale <= 2e-6, (case, variant, precision)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_conv_channel_slices_per_image_bias_prelu(precision):
    """(ld, offset) slices on input/output/residual, per-image bias, PReLU and sigmoid epilogues."""
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(3)
    b, n = 3, 130
    buf = torch.randn(b, n, 1, 96, generator=g)
    wt = torch.randn(40, 32, generator=g) / 6
    bias_img = torch.randn(b, 48, generator=g)            # bias_bstride = 48 > Cout
    res = torch.randn(b, n, 1, 64, generator=g)
    want = torch.einsum("bnc,oc->bno", buf[:, :, 0, 32:64], wt) + bias_img[:, None, :40] + res[:, :, 0, 8:48]
    want = torch.where(want > 0, want, 0.1 * want)
    out = torch.full((b, n, 1, 80), 7.0, device="cuda")
    conv = E.Conv(wt, None, act=E.ACT_PRELU, alpha=0.1, device="cuda", precision=precision)
    conv(buf.cuda(), out=out, xoff=32, yoff=16, residual=res.cuda(), roff=8, bias=bias_img.cuda(), bias_bstride=48)
    got = out.cpu()
    assert torch.all(got[..., :16] == 7.0) and torch.all(got[..., 56:] == 7.0)          # untouched outside the slice
    err = (got[:, :, 0, 16:56] - want).abs().max().item() / want.abs().max().item()
    assert err <= TOL[precision]
    sig = E.Conv(wt, None, act=E.ACT_SIGMOID, device="cuda", precision=precision)(buf.cuda(), xoff=32).cpu()
    want_s = torch.sigmoid(torch.einsum("bnc,oc->bno", buf[:, :, 0, 32:64], wt))
    assert (sig[:, :, 0] - want_s).abs().max().item() <= max(TOL[precision], 2e-6) * 4


def test_conv_rejects_bad_geometry():
    from autoposeestimation_amd import engine as E
    from autoposeestimation_amd._lib import ApeError
    conv = E.Conv(torch.randn(8, 6, 3, 3), None, 1, 1, 1, device="cuda")
    with pytest.raises(ApeError):
        conv(torch.zeros(1, 8, 8, 6, device="cuda"))         # ld not a multiple of 4
    with pytest.raises(ValueError):
        conv(torch.zeros(1, 8, 8, 8, device="cuda"), out=torch.zeros(1, 7, 8, 8, device="cuda"))


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16"])
# Grids larger than the chip (256 CUs x 2 workgroups): round 1 saw a table-read fault of the fused up-sampling only there.  Per
# instantiation (BNH = 64 for Cout <= 64, else 128; NSPLIT from the precision) at least three such grids, ragged tile counts included:
# (2,120,160,64,64) 600 workgroups, (1,200,168,64,64) 525, (3,136,104,64,64) 663 | (1,200,168,32,160) 1050, (2,120,160,64,128) 600,
# (3,136,104,32,128) 663
@pytest.mark.parametrize("shape", [(2, 24, 40, 64, 64), (1, 17, 9, 32, 160), (1, 120, 160, 64, 64), (2, 120, 160, 64, 64),
                                   (1, 200, 168, 64, 64), (3, 136, 104, 64, 64), (1, 200, 168, 32, 160), (2, 120, 160, 64, 128),
                                   (3, 136, 104, 32, 128)])
def test_fused_upsample_conv_equals_materialised(shape, precision):
    """nn.Upsample(x2, align_corners=True) + 3x3 conv (pspnet.py:30-32): the up-sampling fused into the LDS-halo kernel's
    load must equal bilinear kernel -> same conv, bit for bit (same interpolation op order, same products)."""
    from autoposeestimation_amd import engine as E
    b, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(b, h, w, cin, generator=g) * 2).cuda()
    conv = E.Conv(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5, torch.randn(cout, generator=g), 1, 1, 1, E.ACT_PRELU,
                  alpha=0.25, device="cuda", precision=precision)
    fused = conv(x, upsample2x=True)
    ref = conv(E.bilinear(x, 2 * h, 2 * w, True))
    assert fused.shape == (b, 2 * h, 2 * w, cout)
    if precision == "f32" or (2 * h * 2 * w) >= 0.8 * (-(-2 * h // 16) * -(-2 * w // 16) * 256):
        assert torch.equal(fused, ref)          # same kernel on both sides: the interpolation op order makes them bit-identical
    else:                                       # small maps: the unfused side runs the generic kernel (different MFMA shape)
        assert (fused - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    want = F.prelu(F.conv2d(F.interpolate(x.cpu().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True),
                            conv.w.cpu().permute(0, 3, 1, 2), conv.bias.cpu(), 1, 1), torch.tensor([0.25])).permute(0, 2, 3, 1)
    err = (fused.cpu() - want).abs().max().item() / want.abs().max().item()
    assert err <= TOL[precision]


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
# b = 3 below; the larger cases launch 900 / 663 / 1575 workgroups (more than 256 CUs x 2) for every UPS x HEAD instantiation
@pytest.mark.parametrize("ups,h,w", [(True, 24, 40), (False, 48, 80), (True, 21, 33), (True, 120, 160), (True, 136, 104), (True, 200, 168),
                                     (False, 240, 320), (False, 272, 208), (False, 400, 336)])
def test_halo_conv_with_fused_seg_head_is_bit_identical_to_the_unfused_pair(precision, ups, h, w):
    """ape_conv3x3_halo_seghead_bf16 == ape_conv3x3_halo_bf16 followed by ape_seg_head_f32 (labels and scores bit for bit),
    incl. ragged tiles and the fused x2 up-sampling"""
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(h * 100 + w)
    b, c = 3, 5
    x = torch.randn(b, h, w, 64, generator=g).to("cuda")
    wt = (torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5).to("cuda")
    bias = (torch.randn(64, generator=g) * 0.1).to("cuda")
    hw = (torch.randn(c, 64, generator=g) * 0.3).to("cuda").contiguous()
    hb = (torch.randn(c, generator=g) * 0.1).to("cuda")
    conv = E.Conv(wt, bias, 1, 1, 1, E.ACT_PRELU, alpha=0.25, device="cuda", precision=precision)
    feat = conv(x, upsample2x=ups)
    want_label, want_score = E.seg_head(feat, hw, hb, True)
    label, score = E.conv_seg_head(conv, x, hw, hb, True, upsample2x=ups)
    assert label.shape == want_label.shape
    assert torch.equal(label, want_label) and torch.equal(score, want_score)
    assert len(torch.unique(label)) > 1




@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("batch,sizes,cin,cout", [(64, (1, 2, 3, 6), 512, 512), (3, (1, 2, 3, 6), 512, 512), (1, (6,), 64, 200), (5, (2, 7, 1), 96, 33)])
def test_multi_problem_1x1_launch_is_bit_identical_to_the_separate_launches(batch, sizes, cin, cout, precision):
    """ape_conv_gemm_bf16_multi (the PSP module's stage convolutions, pspnet.py:15-18, 22, in one launch) against one ape_conv_gemm_bf16 call per
    problem: different weights, maps and tile counts per problem, ragged M and Cout; bitwise."""
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(5)
    convs = [E.Conv(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, torch.randn(cout, generator=g) if i % 2 else None, act=E.ACT_RELU if i == 1 else E.ACT_NONE,
                    device="cuda", precision=precision) for i in range(len(sizes))]
    xs = [torch.randn(batch, s, s, cin, generator=g).cuda() for s in sizes]
    want = [c(x) for c, x in zip(convs, xs)]
    got = E.conv1x1_multi(convs, xs)
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.equal(a, b)
    # it is the one-launch path that ran, not the fall-back: the library takes these problems
    assert all(c.nsplit for c in convs)


def test_multi_problem_launch_rejects_what_it_does_not_take():
    import ctypes
    from autoposeestimation_amd import _lib, engine as E
    conv3 = E.Conv(torch.randn(64, 64, 3, 3) / 24, None, 1, 1, 1, device="cuda", precision="bf16x3")
    x = torch.randn(1, 8, 8, 64).cuda()
    y = torch.empty(1, 8, 8, 64).cuda()
    p = (E.ConvParams * 1)(E.ConvParams(B=1, H=8, W=8, Cin=64, ldx=64, xoff=0, Ho=8, Wo=8, Cout=64, ldy=64, yoff=0, KH=3, KW=3, stride=1, pad=1, dil=1, act=0, alpha=0.0,
                                        bias_bstride=0, ldr=0, roff=0, ups=0))
    arr = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())  # noqa: E731
    assert _lib.lib().ape_conv_gemm_bf16_multi(1, arr(x), arr(conv3.wp), None, arr(y), p, 3, None) == -1      # a 3x3 problem
    assert _lib.lib().ape_conv_gemm_bf16_multi(5, arr(x), arr(conv3.wp), None, arr(y), p, 3, None) == -1      # more than four
    assert _lib.lib().ape_conv_gemm_bf16_multi(0, arr(x), arr(conv3.wp), None, arr(y), p, 3, None) == -1
    # and the engine helper falls back to the separate launches for them
    assert torch.equal(E.conv1x1_multi([conv3], [x])[0], conv3(x))
