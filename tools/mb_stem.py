"""Fused stem (conv7x7s2 + ReLU + maxpool) vs the two-launch path (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
for b, h, w in [(64, 480, 640), (64, 160, 160)]:
    x4 = torch.randn(b, h, w, 4, device="cuda")
    conv = E.Conv(torch.randn(64, 3, 7, 7) / 12, None, 2, 3, 1, E.ACT_RELU, device="cuda", precision="bf16x3")
    for fused in (True, False, True, False):
        E.USE_FUSED_STEM = fused
        for _ in range(2): E.stem_pool(conv, x4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): E.stem_pool(conv, x4)
        e1.record(); torch.cuda.synchronize()
        print(b, h, w, "fused" if fused else "two launches", "%.3f ms" % (e0.elapsed_time(e1) / 5))
