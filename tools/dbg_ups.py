import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
torch.manual_seed(0)
conv = E.Conv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), 1, 1, 1, E.ACT_PRELU, 0.25, device="cuda", precision="bf16x3")
for B, h, w in [(1, 8, 8), (1, 16, 16), (2, 24, 40), (1, 40, 40), (1, 64, 64), (1, 120, 160), (4, 240, 320)]:
    x = torch.randn(B, h, w, 64, device="cuda")
    y1 = conv(x, upsample2x=True)
    y2 = conv(x, upsample2x=True)
    y3 = conv(E.bilinear(x, 2 * h, 2 * w, True))
    d = (y1 - y3).abs()
    bad = (d > 0).nonzero()
    print(B, h, w, "twice equal", torch.equal(y1, y2), "== materialised", torch.equal(y1, y3), "max diff %.3g" % d.max().item(),
          "n_bad", len(bad), bad[:3].tolist())
