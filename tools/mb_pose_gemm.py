"""Pose-side 1x1 GEMM shapes (M = 64 x 1000 points / 64 x 20 x 20 crop maps): conv_gemm.hip (fp32 in) against conv_gemm_s32.hip (S32 in) and
the cost of producing S32 in the epilogue.  Result (round 2): gemm_s32 is 8-10 % faster, writing S32 costs the producer 4-10 % -- a wash, so
the pose side keeps fp32 activations.  python tools/mb_pose_gemm.py"""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from autoposeestimation_amd import engine as E
shapes = [("feat 512->1024 (x3/step)", 64, 1000, 1, 512, 1024), ("384->1920", 64, 1000, 1, 384, 1920), ("crop up_1 mix", 64, 20, 20, 1024, 2304),
          ("640->256 (x3)", 64, 1000, 1, 640, 256), ("384->512 (x2)", 64, 1000, 1, 384, 512), ("crop bottleneck 512->1024", 64, 20, 20, 512, 1024)]
torch.manual_seed(0)
for name, b, h, w, cin, cout in shapes:
    x = torch.randn(b, h, w, cin, device="cuda")
    xs = E.S32.from_f32(x)
    conv = E.Conv(torch.randn(cout, cin) / cin ** 0.5, torch.randn(cout), act=E.ACT_RELU, device="cuda", precision="bf16x3")
    out = torch.empty(b, h, w, cout, device="cuda")
    arms = {"conv_gemm (fp32 in)": lambda: conv(x, out=out), "conv_gemm fp32 in -> s32": lambda: conv(x, out=out, out_fmt=E.FMT_S32),
            "gemm_s32 -> f32": lambda: conv(xs, out=out), "gemm_s32 -> s32": lambda: conv(xs, out=out, out_fmt=E.FMT_S32)}
    for f in arms.values(): f()
    torch.cuda.synchronize()
    times = {k: [] for k in arms}
    for rnd in range(7):
        for k, f in arms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 5)
    flop = 2.0 * b * h * w * cin * cout
    print(name, "M=%d" % (b * h * w))
    for k, t in times.items():
        t = sorted(t)
        print("   %-26s median %.1f us  %.0f TF/s (%.2f)" % (k, t[3] * 1e3, flop / t[3] / 1e9, flop / t[3] / 1e9 / 833.3))
