"""diagnostic (APE_UPFUSE_DUMP build): which values are wrong when the fused head differs from the three-call form"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E
fma = bool(int(os.environ.get("FMA", "0")))
g = torch.Generator().manual_seed(5)
w = torch.randn(64, 64, 3, 3, generator=g) / 24
b = torch.randn(64, generator=g)
up = E.UpConv(w, b, 0.25, device="cuda", precision="bf16x3", fma=fma)
B, h, wd = 2, 240, 320
g = torch.Generator().manual_seed(7 + 131 * h + wd)
xs = E.S32.from_f32((torch.randn(B, h, wd, 64, generator=g) * 2).cuda())
g = torch.Generator().manual_seed(13)
hw = (torch.randn(13, 64, generator=g) / 8).cuda()
hb = torch.randn(13, generator=g).cuda()
wl, ws = up.seg_head(xs, hw, hb, True, fused=False)
act_w = up(xs, fused=False)
dump = torch.zeros(B, 2 * h, 2 * wd, 64, dtype=torch.float32, device="cuda")
_lib.lib().ape_upconv3x3_fused_stamps(dump.data_ptr())
for rep in range(6):
    dump.zero_()
    gl, gs = up.seg_head(xs, hw, hb, True, fused=True)
    torch.cuda.synchronize()
    bad = (gs != ws) | (gl != wl)
    abad = (dump != act_w).any(dim=3)
    print("rep", rep, "wrong head pixels", int(bad.sum()), " pixels with wrong dumped activations", int(abad.sum()), " both", int((bad & abad).sum()))
    if int(abad.sum()):
        idx = abad.nonzero()[:3]
        for bb, yy, xx in idx.tolist():
            d = dump[bb, yy, xx] - act_w[bb, yy, xx]
            ch = (d != 0).nonzero().flatten().tolist()
            print("   pixel", (bb, yy, xx), "tile row", yy % 16, "col", xx % 24, " wrong channels", ch[:20], len(ch), " max |d|", float(d.abs().max()))
            # is the wrong value another pixel's right value?
            for c in ch[:2]:
                v = float(dump[bb, yy, xx, c])
                hits = (act_w[bb, :, :, c] == v).nonzero()
                print("      ch", c, "got", v, "want", float(act_w[bb, yy, xx, c]), " equals the reference at", hits[:4].tolist())
