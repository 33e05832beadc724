"""Back-to-back launches of tiny problems: per-launch time of the persistent LDS-DMA kernels (gemm_s32, halo_s32: 512 threads, up to 160 KB of
dynamic LDS) against conv_gemm / an element-wise torch kernel -- does a launch of the big-LDS kernels cost more than a plain one?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E

def timeit(f, n=200):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

torch.manual_seed(0)
for cin, cout in ((128, 128), (256, 192), (512, 256)):
    x = torch.randn(1, 16, 16, cin, device="cuda")
    xs = E.S32.from_f32(x)
    conv = E.Conv(torch.randn(cout, cin) / cin ** 0.5, torch.randn(cout), act=E.ACT_RELU, device="cuda", precision="bf16x3")
    out = torch.empty(1, 16, 16, cout, device="cuda")
    print("1x16x16 %d->%d  conv_gemm %.1f us/launch   gemm_s32 %.1f us/launch" % (cin, cout, timeit(lambda: conv(x, out=out)), timeit(lambda: conv(xs, out=out))))
    c3 = E.Conv(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout), 1, 1, 1, E.ACT_RELU, device="cuda", precision="bf16x3")
    o3 = torch.empty(1, 16, 16, cout, device="cuda")
    print("   3x3: halo (fp32 in) %.1f us/launch   halo_s32 %.1f us/launch" % (timeit(lambda: c3(x, out=o3)), timeit(lambda: c3(xs, out=o3))))
y = torch.zeros(1024, device="cuda")
print("torch add_ %.1f us/launch" % timeit(lambda: y.add_(1.0)))
