"""conv3x3_halo_mx.hip: the 3x3 convolution with half the matrix passes (fp16 main product + block-scaled e2m3 cross terms through
v_mfma_scale_f32_16x16x128_f8f6f4) against a float64 evaluation of the SAME rounded operands, and its input converter against the host
packer (autoposeestimation_amd/mx6.py), bit for bit.  Reference layer: DenseFusion/lib/extractors.py:29-43 (BasicBlock convs of layer 4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lines(t):
    """float32-typed [.., C] tensor holding 128-B lines -> uint8 [.., C / 32, 128]"""
    a = t.detach().cpu().contiguous().numpy().view(np.uint8)
    return a.reshape(t.shape[:-1] + (t.shape[-1] // 32, 128))


@pytest.mark.parametrize("shape", [(2, 9, 7, 64), (1, 60, 80, 512), (3, 5, 5, 32)])
def test_converter_equals_the_host_packer_bitwise(shape):
    from autoposeestimation_amd import _lib, engine as E, mx6
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g) * torch.exp(1.5 * torch.randn(shape[0], shape[1], shape[2], 1, generator=g))
    x[0, 0, 0, :32] = 0.0                              # an all-zero block
    x[0, 0, 1, :32] = torch.tensor([2.0 ** (i - 20) for i in range(32)])       # a block spanning 31 binades
    xs = E.S32.from_f32(x.cuda())
    y = torch.empty_like(xs.t)
    _lib.check(_lib.lib().ape_s32_to_f16m6(_lib.dptr(xs.t, torch.float32), _lib.dptr(y, torch.float32), x.numel() // shape[3], shape[3], None), "ape_s32_to_f16m6")
    torch.cuda.synchronize()
    vals = xs.to_f32().cpu().numpy().reshape(shape[:3] + (shape[3] // 32, 32))           # what the S32 image holds (hi + lo)
    want = mx6.pack_lines(vals)
    got = _lines(y)
    assert got.shape == want.shape
    bad = np.argwhere(got != want)
    assert bad.size == 0, (bad[:5], got[tuple(bad[0][:-1])], want[tuple(bad[0][:-1])])


@pytest.mark.parametrize("geom", [(2, 20, 20, 64, 128, 1, False, "relu"), (1, 33, 17, 128, 256, 2, True, "none"), (2, 16, 48, 64, 128, 4, True, "relu"),
                                  (1, 60, 80, 512, 512, 4, True, "relu"), (3, 12, 12, 192, 160, 1, False, "prelu")])
def test_half_pass_conv_equals_float64_on_the_rounded_operands(geom):
    from autoposeestimation_amd import _lib, engine as E, mx6
    b, h, w, cin, cout, dil, with_res, act = geom
    g = torch.Generator().manual_seed(cin * h + cout)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    bias = torch.randn(cout, generator=g)
    x = torch.relu(torch.randn(b, h, w, cin, generator=g)) * torch.exp(torch.randn(b, h, w, 1, generator=g))
    res = torch.randn(b, h, w, cout, generator=g) if with_res else None
    conv = E.Conv(wt, bias, pad=dil, dil=dil, act={"relu": E.ACT_RELU, "none": E.ACT_NONE, "prelu": E.ACT_PRELU}[act], alpha=0.25, device="cuda", precision="bf16x3")
    xs = E.S32.from_f32(x.cuda())
    old = E.USE_MX6, E.MX6_CIN
    E.USE_MX6, E.MX6_CIN = True, (cin,)
    try:
        for out_fmt in (E.FMT_F32, E.FMT_S32):
            got = conv(xs, residual=None if res is None else res.cuda(), out_fmt=out_fmt)
            got = (got.to_f32() if out_fmt == E.FMT_S32 else got).cpu().double()
            # the operands the kernel multiplied: decoded from the lines it was given
            x1, xq1, xq2 = (torch.from_numpy(v.reshape(b, h, w, cin)).double().permute(0, 3, 1, 2)
                            for v in mx6.unpack_lines(mx6.pack_lines(xs.to_f32().cpu().numpy().reshape(b, h, w, cin // 32, 32))))
            wl = conv.mx6k().cpu().numpy()                                                 # [Cout][9 * Cin / 32][128]
            w1, wq1, wq2 = (torch.from_numpy(v.reshape(cout, 3, 3, cin)).double().permute(0, 3, 1, 2) for v in mx6.unpack_lines(wl))
            want = (F.conv2d(x1, w1, None, 1, dil, dil) + F.conv2d(xq1, wq2, None, 1, dil, dil) + F.conv2d(xq2, wq1, None, 1, dil, dil)).permute(0, 2, 3, 1)
            want = want + bias.double()
            if res is not None:
                want = want + res.double()
            want = {"relu": torch.relu(want), "none": want, "prelu": torch.where(want > 0, want, 0.25 * want)}[act]
            tol = 3e-6 if out_fmt == E.FMT_F32 else 2e-5                                   # fp32 accumulation / the S32 store's 2^-17
            err = (got - want).abs().max().item() / want.abs().max().item()
            assert err <= tol, (out_fmt, err)
            # and it is the bf16x3 result to within the operand rounding DESIGN.md 6e prices (a sanity bound, not the parity criterion)
            E.USE_MX6 = False
            ref = conv(xs, residual=None if res is None else res.cuda(), out_fmt=E.FMT_F32).cpu().double()
            E.USE_MX6 = True
            assert (got - ref).abs().max().item() / ref.abs().max().item() <= 2e-4
    finally:
        E.USE_MX6, E.MX6_CIN = old
