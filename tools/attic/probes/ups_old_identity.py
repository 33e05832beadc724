"""Development aid: identity-weight 3x3 conv (centre tap = I) through the fused up-sampling kernel -> every wrong output element is one
wrong halo item; decode (chunk, j, tid) of the staging thread that produced it."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from autoposeestimation_amd import engine as E
torch.manual_seed(0)
cin = cout = 64
B, h, w = 1, 120, 160
wt = torch.zeros(cout, cin, 3, 3)
wt[torch.arange(64), torch.arange(64), 1, 1] = 1.0
conv = E.Conv(wt, None, 1, 1, 1, E.ACT_PRELU, 1.0, device="cuda", precision="bf16x3")
x = torch.rand(B, h, w, cin, device="cuda") + 0.5
ups = E.bilinear(x, 2 * h, 2 * w, True)
hist = collections.Counter()
NTH, HW = 256, 18
for it in range(8):
    y = conv(x, upsample2x=True)
    torch.cuda.synchronize()
    bad = ((y - ups).abs() > 1e-3)[0].cpu().numpy()
    Y, X, C = np.nonzero(bad)
    print("run %d: %d wrong elements" % (it, len(Y)))
    seen = set()
    for yy, xx, cc in zip(Y.tolist(), X.tolist(), C.tolist()):
        ty, tx = yy // 16, xx // 16
        hy, hx = yy % 16 + 1, xx % 16 + 1
        px = hy * HW + hx
        e = px * 8 + (cc % 32) // 4
        j, tid = e // NTH, e % NTH
        key = (cc // 32, j, tid // 64, (tid % 64) // 16)
        hist[key] += 1
        if (ty, tx, cc // 32, j, tid) not in seen and len(seen) < 6:
            seen.add((ty, tx, cc // 32, j, tid))
            print("    tile (%d,%d) chunk %d j %d tid %d (wave %d lane %d) channel %d: got %.5f want %.5f" %
                  (ty, tx, cc // 32, j, tid, tid // 64, tid % 64, cc, y[0, yy, xx, cc].item(), ups[0, yy, xx, cc].item()))
print("histogram (chunk, j, wave, lane group of 16) -> wrong elements:")
for k in sorted(hist):
    print("   ", k, hist[k])
