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
