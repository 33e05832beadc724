"""stem on uint8 frames (normalisation fused into the patch load) vs ape_preprocess_u8_nhwc4 + stem, 64 frames of 480x640 and 64 crops of 160x160"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
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
g = torch.Generator().manual_seed(0)
rgb = torch.randint(0, 256, (64, 480, 640, 3), generator=g, dtype=torch.uint8).cuda()
conv = E.Conv(torch.randn(64, 3, 7, 7, generator=g) / 12, None, 2, 3, 1, E.ACT_RELU, device="cuda", precision="bf16x3")
for name, hc, wc, rects, d in (("64 frames 480x640", 480, 640, torch.tensor([[i, 0, 0] for i in range(64)], dtype=torch.int32).cuda(), True),
                               ("64 crops 160x160", 160, 160, torch.tensor([[i, 100 + i, 200 + 2 * i] for i in range(64)], dtype=torch.int32).cuda(), False)):
    a = t(lambda: E.stem_pool(conv, E.preprocess_u8(rgb, rects, hc, wc, d)))
    b = t(lambda: E.stem_pool(conv, E.U8Frames(rgb, rects, hc, wc, d)))
    y = E.stem_pool(conv, E.U8Frames(rgb, rects, hc, wc, d))
    print("%-20s preprocess + stem %.3f ms   stem on u8 %.3f ms   output checksum %s" % (name, a, b, hex(int(y.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffff)))
