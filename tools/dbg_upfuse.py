import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E
_lib.lib().ape_upconv3x3_fused_debug(int(os.environ.get('DBG', '0')))
for fma in (False, True):
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g)
    up = E.UpConv(w, b, 0.25, device="cuda", precision="bf16x3", fma=fma)
    for (B, h, wd) in ((3, 24, 40), (2, 240, 320)):
        g = torch.Generator().manual_seed(7 + 131 * h + wd)
        x = (torch.randn(B, h, wd, 64, generator=g) * 2).cuda()
        xs = E.S32.from_f32(x)
        classes = 13
        g = torch.Generator().manual_seed(classes)
        hw = (torch.randn(classes, 64, generator=g) / 8).cuda()
        hb = torch.randn(classes, generator=g).cuda()
        wl, ws = up.seg_head(xs, hw, hb, True, fused=False)
        act_w = up(xs, fused=False)
        for rep in range(2):
            gl, gs = up.seg_head(xs, hw, hb, True, fused=True)
            act_g = up(xs, fused=True)
            d = (gs != ws) | (gl != wl)
            n = int(d.sum())
            print("fma", fma, (B, h, wd), "rep", rep, "mismatching pixels", n, "activation mismatches", int((act_g != act_w).sum()))
            if n:
                idx = d.nonzero()
                bs, ys, xs_ = idx[:, 0], idx[:, 1], idx[:, 2]
                tiles = torch.stack([bs, ys // 16, xs_ // 24], 1).unique(dim=0)
                print("   tiles (b, ty, tx):", tiles.tolist()[:12], len(tiles), " rows in tile:", (ys % 16).unique().tolist(), " cols in tile:", (xs_ % 24).unique().tolist())
                print("   max |score diff|", float((gs - ws).abs().max()), " label diffs", int((gl != wl).sum()))
                lin = ((ys % 8) * 24 + (xs_ % 24))
                print("   groups (wave):", (lin // 16).unique().tolist(), " lanes px:", (lin % 16).unique().tolist())
