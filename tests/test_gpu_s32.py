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
