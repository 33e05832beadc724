"""The one-kernel PSPUpsample (csrc/upconv_fused.hip; DenseFusion/lib/pspnet.py:27-37,51,53-55): low-resolution channel mixing on the matrix
cores + row interpolation on the accumulators + column interpolation through LDS (+ the segmentation head), against
  * its own two-call form (ape_conv_gemm_s32 -> ape_upconv3x3_gather_ex -> ape_seg_head_f32): bit for bit, both interpolation arithmetics,
    both output formats, ragged sizes, more tiles than compute units (the persistent tile walk), the first / last image rows and columns
    (the floor pattern's exceptions);
  * the reference's formulation, nn.Upsample(x2, bilinear, align_corners=True) -> Conv2d(3x3, pad 1) -> PReLU in float64."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, h, w): one partial tile; ragged in both axes; exactly one tile; several tile rows and columns; more tiles than CUs (2 x 30 x 27 = 1620)
SHAPES = [(1, 5, 7), (2, 17, 9), (1, 8, 12), (3, 24, 40), (1, 33, 61), (2, 240, 320)]


def _layer(fma, seed=0):
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g)
    return E.UpConv(w, b, 0.25, device="cuda", precision="bf16x3", fma=fma), w, b


def _input(b, h, w, seed=1):
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(seed + 131 * h + w)
    x = (torch.randn(b, h, w, 64, generator=g) * 2).cuda()
    return E.S32.from_f32(x)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("fma", [False, True])
@pytest.mark.parametrize("out_s32", [False, True])
def test_fused_equals_the_two_call_form_bitwise(shape, fma, out_s32):
    from autoposeestimation_amd import engine as E
    b, h, w = shape
    up, _, _ = _layer(fma)
    xs = _input(b, h, w)
    assert up.fusable(xs)
    fmt = E.FMT_S32 if out_s32 else E.FMT_F32
    want = up(xs, out_fmt=fmt, fused=False)
    got = up(xs, out_fmt=fmt, fused=True)
    wt, gt = (want.t, got.t) if out_s32 else (want, got)
    assert gt.shape == (b, 2 * h, 2 * w, 64)
    if out_s32:      # raw bit patterns (the float view of an S32 buffer may hold NaN patterns)
        assert torch.equal(gt.view(torch.int32), wt.view(torch.int32))
    else:
        assert torch.equal(gt, wt)


@pytest.mark.parametrize("shape", SHAPES[:5])
@pytest.mark.parametrize("fma", [False, True])
def test_fused_matches_the_reference_formulation(shape, fma):
    """pspnet.py:30-36 literally (float64): up-sample, convolve, PReLU"""
    b, h, w = shape
    up, wt, bias = _layer(fma)
    xs = _input(b, h, w)
    x = xs.to_f32().double().cpu().permute(0, 3, 1, 2)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    ref = F.prelu(F.conv2d(ref, wt.double(), bias.double(), padding=1), torch.tensor([0.25], dtype=torch.float64))
    got = up(xs, fused=True).double().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 5e-5 * ref.abs().max().item()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("fma", [False, True])
@pytest.mark.parametrize("classes,double_softmax", [(13, True), (16, False), (3, True)])
def test_fused_head_equals_the_three_call_form_bitwise(shape, fma, classes, double_softmax):
    b, h, w = shape
    up, _, _ = _layer(fma, seed=5)
    xs = _input(b, h, w, seed=7)
    g = torch.Generator().manual_seed(classes)
    hw = (torch.randn(classes, 64, generator=g) / 8).cuda()
    hb = torch.randn(classes, generator=g).cuda()
    want_l, want_s = up.seg_head(xs, hw, hb, double_softmax, fused=False)
    got_l, got_s = up.seg_head(xs, hw, hb, double_softmax, fused=True)
    assert got_l.shape == (b, 2 * h, 2 * w)
    assert torch.equal(got_l, want_l)
    assert torch.equal(got_s, want_s)
    assert int(got_l.max()) < classes


def test_fused_without_bias_and_bounds_are_respected():
    """guard bands around the outputs stay untouched (ragged last tiles store nothing past the image)"""
    from autoposeestimation_amd import _lib, engine as E
    up, _, _ = _layer(True, seed=3)
    b, h, w = 2, 17, 9
    xs = _input(b, h, w, seed=11)
    n = b * 4 * h * w
    lab = torch.full((n + 256,), 77, dtype=torch.uint8, device="cuda")
    sco = torch.full((n + 256,), -5.0, dtype=torch.float32, device="cuda")
    hw = torch.randn(7, 64, device="cuda") / 8
    rc = _lib.lib().ape_upconv3x3_fused_seghead_s32(_lib.dptr(xs.t, torch.float32), _lib.dptr(up.mix.s32k()), None, b, h, w, 64, E.ACT_PRELU, 0.25, 1,
                                                    _lib.dptr(hw), None, 7, _lib.dptr(lab[128:]), _lib.dptr(sco[128:]), 1, _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert bool((lab[:128] == 77).all()) and bool((lab[128 + n:] == 77).all())
    assert bool((sco[:128] == -5.0).all()) and bool((sco[128 + n:] == -5.0).all())
    assert int(lab[128:128 + n].max()) < 7
    # the same call through the layer without its bias
    up.bias = None
    want_l, want_s = up.seg_head(xs, hw, None, True, fused=False)
    assert torch.equal(lab[128:128 + n].view(b, 2 * h, 2 * w), want_l)
    assert torch.equal(sco[128:128 + n].view(b, 2 * h, 2 * w), want_s)


def test_unsupported_geometries_are_refused():
    from autoposeestimation_amd import _lib, engine as E
    lib = _lib.lib()
    assert lib.ape_upconv3x3_fused_supported(240, 320, 64, 64) == 1
    assert lib.ape_upconv3x3_fused_supported(120, 160, 256, 64) == 0       # served by the two-call form
    assert lib.ape_upconv3x3_fused_supported(240, 320, 64, 32) == 0
    up, _, _ = _layer(False)
    x = torch.randn(1, 8, 8, 64, device="cuda")
    assert not up.fusable(x)                                               # fp32 input: two calls
    with pytest.raises(ValueError):
        up(x, fused=True)
    assert up(x).shape == (1, 16, 16, 64)
    assert E.UpConv(torch.randn(64, 64, 3, 3), torch.zeros(64), 0.1, precision="f32").fusable(E.S32.from_f32(x)) is False


@pytest.mark.parametrize("b,h,w,c", [(1, 1, 1, 64), (2, 5, 7, 64), (1, 8, 8, 128), (2, 17, 9, 64), (1, 33, 61, 256), (3, 24, 40, 64), (1, 60, 80, 256)])
@pytest.mark.parametrize("fma", [0, 1])
@pytest.mark.parametrize("out_s32", [False, True])
def test_strip_gather_equals_the_row_gather_bitwise(b, h, w, c, fma, out_s32):
    """ape_upconv3x3_gather_ex's two kernels (csrc/ops.hip): the strip walk (C % 64 == 0: z rows held in registers while a workgroup moves
    down the image) against the one-row kernel, bit for bit -- both arithmetics, both output formats, strips of 1, 2, 7 and 30 rows and one
    strip taller than the image (the row-pair hand-over at every parity, the first / last rows' clamps, ragged last strips and columns)"""
    from autoposeestimation_amd import _lib, engine as E
    lib = _lib.lib()
    g = torch.Generator().manual_seed(7 * h + w + c)
    z = torch.randn(b, h, w, 9 * c, generator=g).cuda()
    bias = torch.randn(c, generator=g).cuda()
    fmt = E.FMT_S32 if out_s32 else E.FMT_F32

    def run(rows):
        old = lib.ape_upconv3x3_gather_strip_rows(rows)
        try:
            out = torch.full((b, 2 * h, 2 * w, c), float("nan"), dtype=torch.float32, device="cuda")
            rc = lib.ape_upconv3x3_gather_ex(_lib.dptr(z), _lib.dptr(bias), _lib.dptr(out), fmt, b, h, w, c, E.ACT_PRELU, 0.25, fma, _lib.stream_ptr())
            assert rc == 0
            torch.cuda.synchronize()
            return out.view(torch.int32)
        finally:
            lib.ape_upconv3x3_gather_strip_rows(old)

    want = run(0)
    for rows in (1, 2, 7, 30, 1000):
        assert torch.equal(run(rows), want), rows
    assert lib.ape_upconv3x3_gather_strip_rows(-1) == 30            # the default is back


@pytest.mark.parametrize("shape,reps", [((2, 240, 320), 300), ((64, 240, 320), 50)])
def test_fused_head_stress_beside_a_busy_stream(shape, reps):
    """Back-to-back launches at bench-like sizes while a second stream keeps changing which compute units are free (the default bench runs
    the pose stage beside the segmentor), every launch compared bit for bit with the three-call form.  This is the test that the historic
    operand order of the hand-written interpolations (csrc/upconv_fused.hip, APE_ASM_SRC1_BCAST) fails in EVERY launch; the idle-GPU,
    one-launch-per-size tests above never saw that fault."""
    from autoposeestimation_amd import engine as E
    b, h, w = shape
    up, _, _ = _layer(True, seed=5)
    xs = _input(b, h, w, seed=7)
    g = torch.Generator().manual_seed(13)
    hw = (torch.randn(13, 64, generator=g) / 8).cuda()
    hb = torch.randn(13, generator=g).cuda()
    want_l, want_s = up.seg_head(xs, hw, hb, True, fused=False)
    want_a = up(xs, fused=False) if b <= 8 else None
    conv = E.Conv(torch.randn(256, 256, 3, 3, generator=g) / 48, None, 1, 1, 1, E.ACT_RELU, precision="bf16x3")
    xb = torch.randn(4, 60, 80, 256, generator=g).cuda()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(20 * reps):
            conv(xb)
    bad = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(reps):
        got_l, got_s = up.seg_head(xs, hw, hb, True, fused=True)
        bad[0] += ((got_s != want_s) | (got_l != want_l)).sum()
        if want_a is not None:
            bad[1] += (up(xs, fused=True) != want_a).sum()
    torch.cuda.synchronize()
    assert bad.tolist() == [0, 0], "wrong head pixels / wrong activations over %d launches: %s" % (reps, bad.tolist())
