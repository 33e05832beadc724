"""A/B of the 1x1 layers of the segmentor at bench size: conv_gemm.hip (fp32 activations, register staging + split) vs conv_gemm_s32.hip
(pre-split activations, LDS-DMA); interleaved rounds in one process (cdna_hip_programming.md 5.4 rule 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E

shapes = [("PSP bottleneck", 64, 60, 80, 512, 1024), ("up_1 mix", 64, 60, 80, 1024, 2304), ("up_2 mix", 64, 120, 160, 256, 576)]
torch.manual_seed(0)
for name, b, h, w, cin, cout in shapes:
    x = torch.randn(b, h, w, cin, device="cuda")
    xs = E.S32.from_f32(x)
    conv = E.Conv(torch.randn(cout, cin) / cin ** 0.5, torch.randn(cout), act=E.ACT_RELU, device="cuda", precision="bf16x3")
    out = torch.empty(b, h, w, cout, device="cuda")
    from autoposeestimation_amd import _lib

    def dbg(bits):
        def f():
            _lib.lib().ape_conv_gemm_s32_debug(bits)
            conv(xs, out=out)
            _lib.lib().ape_conv_gemm_s32_debug(0)
        return f
    arms = {"conv_gemm (fp32 in)": lambda: conv(x, out=out), "gemm_s32 -> f32": lambda: conv(xs, out=out),
            "gemm_s32 -> s32": lambda: conv(xs, out=out, out_fmt=E.FMT_S32),
            "  k-rotation": dbg(16), "  no static priority": dbg(32), "  ablate: no DMA": dbg(1), "  ablate: no barrier": dbg(2), "  ablate: no DMA+bar": dbg(3), "  ablate: no MFMA": dbg(4),
            "  ablate: no reads": dbg(8), "  ablate: no stores": dbg(256), "  ablate: MFMA only": dbg(1 | 2 | 8), "  ablate: no MFMA/reads": dbg(4 | 8)}
    for f in arms.values():
        f()
    torch.cuda.synchronize()
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
    flop = 2.0 * b * h * w * cin * cout
    print("%s  M=%d K=%d N=%d" % (name, b * h * w, cin, cout))
    for k, t in times.items():
        t = sorted(t)
        print("   %-22s median %.3f ms  min %.3f ms   %.0f TFLOP/s algorithmic (%.2f of 833)" % (k, t[len(t) // 2], t[0], flop / t[len(t) // 2] / 1e9, flop / t[len(t) // 2] / 1e9 / 833.3))
