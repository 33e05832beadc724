"""up_3 (+ fused head) and layer1 halo launches in isolation (development aid; APE_HALO_DBG ablation bits apply)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
def timeit(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 64
x = torch.randn(B, 240, 320, 64, device="cuda")
conv = E.Conv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), 1, 1, 1, E.ACT_PRELU, 0.25, device="cuda", precision="bf16x3")
hw, hb = torch.randn(13, 64, device="cuda") / 8, torch.randn(13, device="cuda")
flop = 2.0 * B * 480 * 640 * 64 * 64 * 9
ms = timeit(lambda: E.conv_seg_head(conv, x, hw, hb, True, upsample2x=True))
print("up_3 ups+head   %.3f ms %.0f TF/s" % (ms, flop / ms / 1e9))
out = torch.empty(B, 480, 640, 64, device="cuda")
ms = timeit(lambda: conv(x, out=out, upsample2x=True))
print("up_3 ups        %.3f ms %.0f TF/s" % (ms, flop / ms / 1e9))
xf = torch.randn(B, 480, 640, 64, device="cuda")
ms = timeit(lambda: conv(xf, out=out))
print("3x3 64->64 full %.3f ms %.0f TF/s" % (ms, flop / ms / 1e9))
ms = timeit(lambda: E.conv_seg_head(conv, xf, hw, hb, True))
print("3x3 + head      %.3f ms %.0f TF/s" % (ms, flop / ms / 1e9))
x1 = torch.randn(B, 120, 160, 64, device="cuda")
o1 = torch.empty(B, 120, 160, 64, device="cuda")
ms = timeit(lambda: conv(x1, out=o1, residual=x1))
print("layer1 conv     %.3f ms %.0f TF/s" % (ms, flop / 16 / ms / 1e9))
