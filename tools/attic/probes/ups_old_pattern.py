"""Development aid: where do the fused-upsampling outputs of a given library differ from the materialised path?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from autoposeestimation_amd import engine as E
torch.manual_seed(0)
for cin, cout, shape in [(32, 64, (1, 120, 160)), (64, 64, (1, 120, 160)), (64, 64, (1, 24, 40)), (96, 64, (1, 120, 160)), (64, 128, (1, 24, 40))]:
    conv = E.Conv(torch.randn(cout, cin, 3, 3) / 24, torch.randn(cout), 1, 1, 1, E.ACT_PRELU, 0.25, device="cuda", precision="bf16x3")
    B, h, w = shape
    x = torch.randn(B, h, w, cin, device="cuda")
    ref = conv(E.bilinear(x, 2 * h, 2 * w, True))
    y = conv(x, upsample2x=True)
    d = (y != ref)
    px = d.any(dim=3)[0].cpu().numpy()
    ys, xs = np.nonzero(px)
    print("cin %d cout %d %s: %d of %d pixels differ; max abs diff %.3g (ref max %.3g)" % (cin, cout, shape, px.sum(), px.size, (y - ref).abs().max().item(), ref.abs().max().item()))
    if len(ys):
        print("   rows mod 16:", np.bincount(ys % 16, minlength=16).tolist())
        print("   cols mod 16:", np.bincount(xs % 16, minlength=16).tolist())
        ty, tx = ys // 16, xs // 16
        tiles = set(zip(ty.tolist(), tx.tolist()))
        print("   tiles touched: %d of %d; first: %s" % (len(tiles), ((2*h+15)//16) * ((2*w+15)//16), sorted(tiles)[:12]))
        ch = d[0][torch.from_numpy(px).cuda()].sum(0).cpu().numpy()
        print("   channels differing counts (first 16):", ch[:16].tolist())
