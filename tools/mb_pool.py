"""PSP pooling (pspnet.py:15) at bench size: the 60 x 80 x 512 map (+ 64 spare channels of the folded bottleneck) of 64 frames, pre-split and fp32"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
def t(f, n=10, rounds=7):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[len(ts) // 2]
for b, h, w, ld, c in ((64, 60, 80, 576, 512), (64, 60, 80, 512, 512), (64, 20, 20, 512, 512)):
    x = torch.randn(b, h, w, ld, device="cuda")
    xs = E.S32.from_f32(x)
    for name, src in (("S32", xs), ("fp32", x)):
        ms = t(lambda: E.adaptive_avgpool_multi(src, (2, 3, 6), channels=c))
        print("%dx%dx%d  %d of %d channels  %-4s  %.3f ms  %.2f TB/s" % (b, h, w, c, ld, name, ms, b * h * w * c * 4 / ms / 1e9))
