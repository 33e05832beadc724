"""up_3 + head at bench size (64 frames, 240x320 -> 480x640): the direct form (conv3x3_halo_kernel<3,1,64,true,true>: bilinear x2 fused into the halo
load, head in the epilogue) against the low-resolution one-kernel form (upconv_fused.hip), interleaved rounds in one process."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
w = torch.randn(64, 64, 3, 3) / 24
bias = torch.randn(64)
x = torch.randn(B, 240, 320, 64, device="cuda")
xs = E.S32.from_f32(x)
conv = E.Conv(w, bias, 1, 1, 1, E.ACT_PRELU, 0.25, device="cuda", precision="bf16x3")
ups = {fma: E.UpConv(w, bias, 0.25, device="cuda", precision="bf16x3", fma=fma) for fma in (False, True)}
hw, hb = torch.randn(13, 64, device="cuda") / 8, torch.randn(13, device="cuda")
flop = 2.0 * B * 480 * 640 * 64 * 64 * 9
arms = {"direct (halo kernel, fused x2 + head)": lambda: E.conv_seg_head(conv, x, hw, hb, True, upsample2x=True),
        "low-res fused + head, fma": lambda: ups[True].seg_head(xs, hw, hb, True, fused=True),
        "low-res fused + head, separately rounded": lambda: ups[False].seg_head(xs, hw, hb, True, fused=True),
        "low-res fused -> S32 activation, fma": lambda: ups[True](xs, out_fmt=E.FMT_S32, fused=True),
        "low-res fused -> f32 activation, fma": lambda: ups[True](xs, fused=True)}
outs = {k: f() for k, f in arms.items()}
torch.cuda.synchronize()
l0, s0 = outs["direct (halo kernel, fused x2 + head)"]
l1, s1 = outs["low-res fused + head, fma"]
print("labels differing from the direct form: %d of %d; max |score diff| %.3g" % (int((l0 != l1).sum()), l0.numel(), float((s0 - s1).abs().max())), flush=True)
times = {k: [] for k in arms}
for rnd in range(7):
    for k, f in arms.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            f()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 3)
for k, t in times.items():
    t = sorted(t)
    print("%-42s median %.3f ms  min %.3f ms   %.0f TFLOP/s algorithmic (%.2f of 833)" % (k, t[len(t) // 2], t[0], flop / t[len(t) // 2] / 1e9, flop / t[len(t) // 2] / 1e9 / 833.3), flush=True)
