"""layer1's 3x3 convolutions (64 -> 64 at 120 x 160, 64 frames, fp32 activations: conv3x3_halo.hip) with and without the residual operand: time per
launch and an output checksum, for A/B runs of library builds on one box (APE_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
g = torch.Generator().manual_seed(0)
b, h, w, c = 64, 120, 160, 64
x = torch.relu(torch.randn(b, h, w, c, generator=g)).cuda()
res = torch.randn(b, h, w, c, generator=g).cuda()
conv = E.Conv(torch.randn(c, c, 3, 3, generator=g) / 24, torch.randn(c, generator=g), 1, 1, 1, E.ACT_RELU, device="cuda", precision="bf16x3")
out = torch.empty(b, h, w, c, device="cuda")
for name, kw in (("no residual", {}), ("residual", {"residual": res})):
    f = lambda: conv(x, out=out, **kw)
    for _ in range(3):
        f()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print("%-12s %.1f us   checksum %s" % (name, sorted(ts)[3] * 1e3, hex(int(out.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffff)), flush=True)
