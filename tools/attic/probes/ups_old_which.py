"""Development aid: with a given library, which side of fused-vs-materialised up-sampling conv is wrong (vs an fp32 torch reference), and is it repeatable?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from autoposeestimation_amd import engine as E
torch.manual_seed(0)
cin, cout, (B, h, w) = 64, 64, (1, 120, 160)
wt, bias = torch.randn(cout, cin, 3, 3) / 24, torch.randn(cout)
conv = E.Conv(wt, bias, 1, 1, 1, E.ACT_PRELU, 0.25, device="cuda", precision="bf16x3")
x = torch.randn(B, h, w, cin, device="cuda")
up = F.interpolate(x.permute(0, 3, 1, 2).double(), scale_factor=2, mode="bilinear", align_corners=True)
want = F.prelu(F.conv2d(up, wt.double().cuda(), bias.double().cuda(), 1, 1), torch.tensor([0.25], dtype=torch.float64, device="cuda")).permute(0, 2, 3, 1).float()
ups = E.bilinear(x, 2 * h, 2 * w, True)
print("bilinear kernel vs torch: max abs", (ups - up.permute(0, 2, 3, 1).float()).abs().max().item())
refs = [conv(ups).clone() for _ in range(3)]
fus = [conv(x, upsample2x=True).clone() for _ in range(3)]
torch.cuda.synchronize()
for name, ys in (("materialised", refs), ("fused", fus)):
    for i, y in enumerate(ys):
        d = (y - want).abs()
        bad = (d > 1e-3).any(dim=3)
        print("%-12s run %d: max abs err vs fp64 torch %.3g, pixels off by > 1e-3: %d, identical to run 0: %s" %
              (name, i, d.max().item(), int(bad.sum()), torch.equal(y, ys[0])))
        if bad.any():
            ys_, xs_ = np.nonzero(bad[0].cpu().numpy())
            print("      first bad pixels:", list(zip(ys_.tolist(), xs_.tolist()))[:10])
