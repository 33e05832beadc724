"""Pre-split ("S32") activations and the LDS-DMA GEMM that consumes them (csrc/conv_gemm_s32.hip): the MFMA operands are bit-identical
to what conv_gemm.hip derives from fp32 activations, so with an fp32 output and no residual the two kernels must agree bit for bit;
the S32 output / residual forms within the split's own 2^-16."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_s32_roundtrip_is_hi_plus_lo():
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(3, 5, 7, 96, generator=g) * 50).cuda()
    s = E.S32.from_f32(x)
    back = s.to_f32()
    hi = x.bfloat16().float()
    lo = (x - hi).bfloat16().float()
    assert torch.equal(back, hi + lo)
    assert (back - x).abs().max().item() <= 2.0 ** -16 * x.abs().max().item()
    raw = s.t.view(torch.bfloat16).view(3, 5, 7, 3, 2, 32)          # [.., group, hi|lo, 32]
    assert torch.equal(raw[..., 0, :].float().reshape(3, 5, 7, 96), hi)


# (B, H, W, Cin, Cout): ragged M (not a multiple of 256), every block shape (Cout <= 128 -> 128, 576 -> 192, else 256), odd k-tile counts
@pytest.mark.parametrize("shape", [(2, 24, 40, 256, 576), (1, 17, 9, 96, 160), (3, 20, 20, 512, 1024), (1, 60, 80, 1024, 2304),
                                   (2, 30, 40, 64, 128), (1, 13, 11, 32, 2304), (5, 16, 16, 160, 384)])
def test_gemm_s32_equals_conv_gemm_bitwise(shape):
    from autoposeestimation_amd import engine as E
    b, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(cin + cout)
    x = (torch.randn(b, h, w, cin, generator=g) * 3).cuda()
    conv = E.Conv(torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.randn(cout, generator=g), act=E.ACT_RELU, device="cuda", precision="bf16x3")
    want = conv(x)
    got = conv(E.S32.from_f32(x))
    assert got.shape == want.shape and torch.equal(got, want)
    ref = F.relu(torch.einsum("bhwc,oc->bhwo", x.double(), conv.w.view(cout, -1)[:, :cin].double()) + conv.bias.double())
    assert (got.double() - ref).abs().max().item() <= 5e-5 * ref.abs().max().item()


# more tiles than CUs: the residual-free kernel then WALKS its tiles (one workgroup per CU, the k-tile stream running through the tile
# boundary); ragged M and Cout, 4 / 6 / 8 k-tiles, an odd k-tile count (no walk), both output formats
@pytest.mark.parametrize("shape", [(16, 60, 80, 128, 256), (5, 61, 83, 256, 576), (10, 50, 70, 192, 300), (16, 60, 80, 160, 256)])
@pytest.mark.parametrize("out_s32", [False, True])
def test_gemm_s32_tile_walk_equals_one_tile_per_workgroup_bitwise(shape, out_s32):
    from autoposeestimation_amd import _lib
    from autoposeestimation_amd import engine as E
    b, h, w, cin, cout = shape
    if out_s32 and cout % 32:
        pytest.skip("S32 output needs Cout % 32 == 0")
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = (torch.randn(b, h, w, cin, generator=g) * 3).cuda()
    conv = E.Conv(torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.randn(cout, generator=g), act=E.ACT_RELU, device="cuda", precision="bf16x3")
    xs = E.S32.from_f32(x)
    fmt = E.FMT_S32 if out_s32 else E.FMT_F32
    try:
        walk = conv(xs, out_fmt=fmt)
        assert _lib.lib().ape_conv_gemm_s32_debug(64) == 0          # bit 64: one tile per workgroup
        one = conv(xs, out_fmt=fmt)
    finally:
        _lib.lib().ape_conv_gemm_s32_debug(0)
    wt, ot = (walk.t, one.t) if out_s32 else (walk, one)
    assert torch.equal(wt.view(torch.int32), ot.view(torch.int32))
    if not out_s32:
        assert torch.equal(walk, conv(x))                            # and the fp32-activation kernel, as above


def test_gemm_s32_output_residual_and_slices():
    from autoposeestimation_amd import engine as E
    g = torch.Generator().manual_seed(3)
    b, h, w = 2, 19, 23
    buf = (torch.randn(b, h, w, 192, generator=g)).cuda()              # the layer reads channels 64..191 of a 192-channel buffer
    conv = E.Conv(torch.randn(256, 128, generator=g) / 11, torch.randn(256, generator=g), act=E.ACT_PRELU, alpha=0.1, device="cuda", precision="bf16x3")
    res = torch.randn(b, h, w, 256, generator=g).cuda()
    want = conv(buf, xoff=64, residual=res)
    xs = E.S32.from_f32(buf)
    # fp32 residual, S32 output
    ys = conv(xs, xoff=64, residual=res, out_fmt=E.FMT_S32)
    assert isinstance(ys, E.S32)
    hi = want.bfloat16().float()
    assert torch.equal(ys.to_f32(), hi + (want - hi).bfloat16().float())       # = split(want): same accumulators, split in the epilogue
    # S32 residual: hi + lo of the residual is added instead of its fp32 value
    got = conv(xs, xoff=64, residual=E.S32.from_f32(res))
    assert (got - want).abs().max().item() <= 2.0 ** -15 * res.abs().max().item()
    # output written into a channel slice of a wider S32 buffer
    wide = E.S32(torch.zeros(b, h, w, 320, device="cuda"))
    conv(xs, out=wide, xoff=64, yoff=32, residual=res, out_fmt=E.FMT_S32)
    w32 = wide.to_f32()
    assert torch.equal(w32[..., 32:288], ys.to_f32()) and torch.all(w32[..., :32] == 0) and torch.all(w32[..., 288:] == 0)


# (B, H, W, Cin, Cout, dil): ragged tiles, every dilation, odd and even chunk counts, Cout not a multiple of 128, grids larger than the chip
@pytest.mark.parametrize("shape", [(1, 16, 16, 32, 128, 1), (2, 24, 40, 64, 128, 1), (1, 17, 9, 96, 160, 1), (3, 60, 80, 128, 256, 2), (1, 21, 33, 64, 192, 4),
                                   (2, 60, 80, 256, 256, 4), (4, 30, 40, 160, 128, 2), (8, 60, 80, 64, 512, 1)])
def test_halo_s32_equals_halo_bitwise(shape):
    """ape_conv3x3_halo_s32 == ape_conv3x3_halo_bf16(nsplit 3) on the fp32 form of the same input: same products, same K order"""
    from autoposeestimation_amd import engine as E
    b, h, w, cin, cout, dil = shape
    g = torch.Generator().manual_seed(cin + cout + dil)
    x = (torch.randn(b, h, w, cin, generator=g) * 2).cuda()
    conv = E.Conv(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5, torch.randn(cout, generator=g), 1, dil, dil, E.ACT_RELU,
                  device="cuda", precision="bf16x3")
    res = torch.randn(b, h, w, cout, generator=g).cuda()
    E.USE_HALO_KERNEL = True
    want = conv(x, residual=res)
    xs = E.S32.from_f32(x)
    got = conv(xs, residual=res)
    if (h * w) >= 0.8 * (-(-h // 16) * -(-w // 16) * 256):       # else `want` came from the generic GEMM kernel (another k order)
        assert torch.equal(got, want)
    else:
        assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item()
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), conv.w.permute(0, 3, 1, 2)[:, :cin].double(), conv.bias.double(), 1, dil, dil)
                 + res.permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1)
    assert (got.double() - ref).abs().max().item() <= 5e-5 * ref.abs().max().item()
    # S32 output and S32 residual
    ys = conv(xs, residual=res, out_fmt=E.FMT_S32)
    hi = got.bfloat16().float()
    assert torch.equal(ys.to_f32(), hi + (got - hi).bfloat16().float())
    got2 = conv(xs, residual=E.S32.from_f32(res))
    assert (got2 - got).abs().max().item() <= 2.0 ** -15 * res.abs().max().item()


@pytest.mark.parametrize("shape", [(1, 16, 16, 32, 128, 1), (1, 17, 9, 96, 160, 1), (3, 60, 80, 128, 256, 2), (1, 21, 33, 64, 192, 4), (4, 30, 40, 160, 128, 2),
                                   (64, 60, 80, 256, 256, 1), (64, 60, 80, 512, 512, 4), (40, 28, 50, 128, 384, 4), (17, 60, 80, 64, 128, 2)])
def test_halo_s32_ping_pong_equals_the_lockstep_schedule_bitwise(shape):
    """halo_s32's round-6 schedule (the two waves of a SIMD run a tap's matrix segment and its load segment in opposite order; waves 0-3
    close their barrier interval behind the loads, waves 4-7 behind the MFMAs) against the one it replaced (ape_conv3x3_halo_s32_debug bit
    4096 launches the lockstep kernel; both are in the product library): same products, same order per accumulator, same DMA duty per tap ->
    the same bits, on one-tile and many-tile workgroups, odd and even chunk counts, ragged tiles, every dilation, with and without an S32
    residual, fp32 and S32 outputs -- and the same bits again on every one of ten repetitions of the big shapes (an LDS-DMA piece read
    before the barrier that publishes it would come and go with timing)."""
    from autoposeestimation_amd import _lib
    from autoposeestimation_amd import engine as E
    b, h, w, cin, cout, dil = shape
    g = torch.Generator().manual_seed(cin * 7 + cout + dil + b)
    xs = E.S32.from_f32((torch.randn(b, h, w, cin, generator=g) * 2).cuda())
    conv = E.Conv(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5, torch.randn(cout, generator=g), 1, dil, dil, E.ACT_PRELU, alpha=0.25,
                  device="cuda", precision="bf16x3")
    res = E.S32.from_f32(torch.randn(b, h, w, cout, generator=g).cuda())
    lib = _lib.lib()
    try:
        lib.ape_conv3x3_halo_s32_debug(4096)
        want = [conv(xs).clone(), conv(xs, residual=res, out_fmt=E.FMT_S32).t.clone()]
        lib.ape_conv3x3_halo_s32_debug(0)
        reps = 10 if b * h * w >= 64 * 60 * 80 else 2
        for rep in range(reps):
            got = [conv(xs), conv(xs, residual=res, out_fmt=E.FMT_S32).t]
            for k, (gt, wt) in enumerate(zip(got, want)):
                assert torch.equal(gt.view(torch.int32), wt.view(torch.int32)), (shape, rep, k, int((gt.view(torch.int32) != wt.view(torch.int32)).sum()))
    finally:
        lib.ape_conv3x3_halo_s32_debug(0)


@pytest.mark.parametrize("shape", [(1, 16, 16, 64, 128), (3, 19, 23, 96, 160), (8, 60, 80, 128, 256), (64, 60, 80, 256, 512), (4, 120, 160, 256, 576), (2, 1000, 1, 384, 1920)])
def test_gemm_s32_ping_pong_form_equals_the_lockstep_form_bitwise(shape):
    """gemm_s32's ping-pong schedule (round 6, ape_conv_gemm_s32_debug bit 8192; the lockstep form stays the product: tools/mb_gemm_pp.py) --
    opposite segment order on the two waves of a SIMD, every DMA piece from waves 4-7, two pixel slots + three weight slots -- gives the
    lockstep form's bits: one-tile and walking workgroups, odd k-tile counts, ragged row and channel tiles, all three channel-tile widths,
    with and without a residual, repeated"""
    from autoposeestimation_amd import _lib
    from autoposeestimation_amd import engine as E
    b, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(cin + cout + b)
    xs = E.S32.from_f32((torch.randn(b, h, w, cin, generator=g) * 2).cuda())
    conv = E.Conv(torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.randn(cout, generator=g), act=E.ACT_PRELU, alpha=0.25, device="cuda", precision="bf16x3")
    res = torch.randn(b, h, w, cout, generator=g).cuda()
    lib = _lib.lib()
    try:
        want = [conv(xs).clone(), conv(xs, residual=res, out_fmt=E.FMT_S32).t.clone()]
        lib.ape_conv_gemm_s32_debug(8192)
        for rep in range(3):
            got = [conv(xs), conv(xs, residual=res, out_fmt=E.FMT_S32).t]
            for k, (gt, wt) in enumerate(zip(got, want)):
                assert torch.equal(gt.view(torch.int32), wt.view(torch.int32)), (shape, rep, k)
    finally:
        lib.ape_conv_gemm_s32_debug(0)


def test_s32_path_rejects_what_it_cannot_run():
    from autoposeestimation_amd import engine as E
    x = E.S32(torch.zeros(1, 8, 8, 64, device="cuda"))
    with pytest.raises(ValueError):
        E.Conv(torch.randn(128, 64), None, device="cuda", precision="f32")(x)
    with pytest.raises(ValueError):
        E.Conv(torch.randn(64, 64, 3, 3), None, 1, 1, 1, device="cuda", precision="bf16x3")(x)       # the S32 3x3 kernel needs Cout >= 128
    with pytest.raises(ValueError):
        E.Conv(torch.randn(128, 64, 3, 3), None, 2, 1, 1, device="cuda", precision="bf16x3")(x)      # ... and stride 1
    with pytest.raises(ValueError):
        E.S32(torch.zeros(1, 8, 8, 48, device="cuda"))


# (B, h, w, Cin, Cout): rows per image below one tile, ragged last tiles, the segmentor's map (4800 rows = 18.75 tiles), more tiles than CUs
@pytest.mark.parametrize("shape", [(3, 12, 16, 128, 256), (2, 33, 61, 512, 1024), (5, 60, 80, 512, 1024), (20, 60, 80, 128, 192)])
@pytest.mark.parametrize("out_s32", [False, True])
def test_gemm_s32_per_image_equals_one_call_per_image_bitwise(shape, out_s32):
    """ape_conv_gemm_s32_per_image: image i multiplies with the weights at w + i * stride and no tile holds rows of two images -- i.e. it is
    the one-image call repeated, bit for bit (same tiles, same k order)."""
    import ctypes
    from autoposeestimation_amd import _lib, engine as E
    b, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(cin + 3 * cout + h)
    xs = E.S32.from_f32((torch.randn(b, h, w, cin, generator=g) * 3).cuda())
    bias = torch.randn(cout, generator=g).cuda()
    convs = [E.Conv(torch.randn(cout, cin, generator=g) / cin ** 0.5, bias, act=E.ACT_RELU, device="cuda", precision="bf16x3") for _ in range(b)]
    fmt = E.FMT_S32 if out_s32 else E.FMT_F32
    want = torch.stack([(convs[i](xs[i:i + 1], out_fmt=fmt).t if out_s32 else convs[i](xs[i:i + 1]))[0] for i in range(b)])
    wall = torch.stack([c.s32k() for c in convs]).contiguous()              # [B][Cout * K * 2] bf16
    got = torch.empty(b, h, w, cout, dtype=torch.float32, device="cuda")
    p = E.ConvParams(B=b, H=h, W=w, Cin=cin, ldx=cin, xoff=0, Ho=h, Wo=w, Cout=cout, ldy=cout, yoff=0, KH=1, KW=1, stride=1, pad=0, dil=1,
                     act=E.ACT_RELU, alpha=0.0, bias_bstride=0, ldr=0, roff=0, ups=0)
    rc = _lib.lib().ape_conv_gemm_s32_per_image(_lib.dptr(xs.t, torch.float32), _lib.dptr(wall), cout * cin * 4, _lib.dptr(bias), _lib.dptr(got), fmt,
                                                ctypes.byref(p), _lib.stream_ptr())
    assert rc == 0
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert _lib.lib().ape_conv_gemm_s32_per_image(_lib.dptr(xs.t, torch.float32), _lib.dptr(wall), 0, _lib.dptr(bias), _lib.dptr(got), fmt,
                                                  ctypes.byref(p), _lib.stream_ptr()) != 0          # a stride is required


@pytest.mark.parametrize("shape", [(2, 12, 16), (3, 33, 61), (4, 60, 80), (1, 7, 5)])
def test_psp_bottleneck_with_the_prior_sum_folded_into_k(shape):
    """pspnet.py:12-24: relu(W_f . f + sum_s upsample(z_s) + b).  Folded form (engine.psp_bottleneck_folded: 50 coefficient columns appended
    to K, per-frame weight rows) against (a) the two-kernel form -- ape_psp_prior_sum_f32 as the residual of ape_conv_gemm_s32 -- within the
    split-bf16 operand error of the extra columns, (b) float64: F.interpolate(bilinear, align_corners=False) of the priors + the 1x1 conv."""
    from autoposeestimation_amd import engine as E
    b, h, w = shape
    cin, cout = 512, 1024
    g = torch.Generator().manual_seed(h * 100 + w)
    f = (torch.randn(b, h, w, cin, generator=g) * 2).cuda()
    zs = [torch.randn(b, s, s, cout, generator=g).cuda() for s in (1, 2, 3, 6)]
    conv = E.Conv(torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.randn(cout, generator=g), act=E.ACT_RELU, device="cuda", precision="bf16x3")
    fs = E.S32.from_f32(f)
    want = conv(fs, residual=E.psp_prior_sum(zs, h, w), out_fmt=E.FMT_S32).to_f32()
    f576 = torch.full((b, h, w, cin + E.PSP_FOLD_K), float("nan"), dtype=torch.float32, device="cuda")     # (the spare channels start as garbage)
    f576[..., :cin] = fs.t
    got = E.psp_bottleneck_folded(conv, E.S32(f576), zs).to_f32()
    assert torch.equal(f576[..., :cin], fs.t)                                    # the map's own channels are untouched
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 3e-5 * scale
    ref = torch.einsum("bhwc,oc->bhwo", fs.to_f32().double(), conv.w.view(cout, -1)[:, :cin].double()) + conv.bias.double()
    for z in zs:
        ref = ref + F.interpolate(z.double().permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    ref = F.relu(ref)
    assert (got.double() - ref).abs().max().item() <= 5e-5 * ref.abs().max().item()
