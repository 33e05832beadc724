"""what the PSP bottleneck (pspnet.py:22-24) costs in its forms: 512 -> 1024 + fp32 residual (today), the same without the residual,
and 576 -> 1024 without the residual (the prior sum folded into K), at bench size"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
torch.manual_seed(0)
b, h, w = 64, 60, 80
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
x512 = E.S32.from_f32(torch.randn(b, h, w, 512, device="cuda"))
x576 = E.S32.from_f32(torch.randn(b, h, w, 576, device="cuda"))
res = torch.randn(b, h, w, 1024, device="cuda")
c512 = E.Conv(torch.randn(1024, 512) / 22, torch.randn(1024), act=E.ACT_RELU, device="cuda", precision="bf16x3")
c576 = E.Conv(torch.randn(1024, 576) / 24, torch.randn(1024), act=E.ACT_RELU, device="cuda", precision="bf16x3")
out = E.S32(torch.empty(b, h, w, 1024, device="cuda"))
print("512 -> 1024 + fp32 residual -> S32   %.3f ms" % t(lambda: c512(x512, residual=res, out=out, out_fmt=E.FMT_S32)))
print("512 -> 1024 (no residual)   -> S32   %.3f ms" % t(lambda: c512(x512, out=out, out_fmt=E.FMT_S32)))
print("576 -> 1024 (no residual)   -> S32   %.3f ms" % t(lambda: c576(x576, out=out, out_fmt=E.FMT_S32)))
x576v = E.S32(x576.t[..., :512]) if False else None
zs = [torch.randn(b, s, s, 1024, device="cuda") for s in (1, 2, 3, 6)]
print("psp_prior_sum                        %.3f ms" % t(lambda: E.psp_prior_sum(zs, h, w)))
