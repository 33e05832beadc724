"""halo_mx (fp16 main + block-scaled e2m3 cross terms: 6 matrix passes per tap and block) against halo_s32 (three bf16 products: 12),
layer 4's shapes at bench size, with and without the S32 -> F16M6 pass"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E
def t(f, n=5, rounds=7):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[len(ts) // 2]
b, h, w = 64, 60, 80
for cin, cout, dil in ((512, 512, 1), (512, 512, 4), (256, 512, 1), (256, 256, 2), (128, 128, 1)):
    conv = E.Conv(torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5), torch.randn(cout), pad=dil, dil=dil, act=E.ACT_RELU, device="cuda", precision="bf16x3")
    xs = E.S32.from_f32(torch.relu(torch.randn(b, h, w, cin, device="cuda")))
    res = E.S32.from_f32(torch.randn(b, h, w, cout, device="cuda"))
    out = E.S32(torch.empty(b, h, w, cout, device="cuda"))
    E.USE_MX6 = False
    t_s32 = t(lambda: conv(xs, residual=res, out=out, out_fmt=E.FMT_S32))
    E.USE_MX6, E.MX6_CIN = True, (cin,)
    t_mx = t(lambda: conv(xs, residual=res, out=out, out_fmt=E.FMT_S32))
    xq = torch.empty_like(xs.t)
    t_cv = t(lambda: _lib.lib().ape_s32_to_f16m6(_lib.dptr(xs.t, torch.float32), _lib.dptr(xq, torch.float32), b * h * w, cin, None))
    E.USE_MX6 = False
    gf = 2.0 * b * h * w * cout * 9 * cin / 1e9
    print("%d -> %d d%d:  halo_s32 %.3f ms (%.0f TF/s)   halo_mx + convert %.3f ms   convert alone %.3f ms   kernel alone %.3f ms (%.0f TF/s)"
          % (cin, cout, dil, t_s32, gf / t_s32, t_mx, t_cv, t_mx - t_cv, gf / (t_mx - t_cv)))
