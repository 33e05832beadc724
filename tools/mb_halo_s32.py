"""A/B of the 3x3 layers of the segmentor at bench size: conv3x3_halo.hip (fp32 activations, register staging + split) vs
conv3x3_halo_s32.hip (pre-split activations, LDS-DMA rings); interleaved rounds in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E

shapes = [("layer2 128ch d1", 64, 60, 80, 128, 128, 1), ("layer3 256ch d1", 64, 60, 80, 256, 256, 1), ("layer3 256ch d2", 64, 60, 80, 256, 256, 2),
          ("layer4 256->512 d1", 64, 60, 80, 256, 512, 1), ("layer4 512ch d1", 64, 60, 80, 512, 512, 1), ("layer4 512ch d4", 64, 60, 80, 512, 512, 4)]
torch.manual_seed(0)
for name, b, h, w, cin, cout, dil in shapes:
    x = torch.randn(b, h, w, cin, device="cuda")
    xs = E.S32.from_f32(x)
    conv = E.Conv(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout), 1, dil, dil, E.ACT_RELU, device="cuda", precision="bf16x3")
    out = torch.empty(b, h, w, cout, device="cuda")
    from autoposeestimation_amd import _lib

    def noprio():
        _lib.lib().ape_conv3x3_halo_s32_debug(1)
        conv(xs, out=out, out_fmt=E.FMT_S32)
        _lib.lib().ape_conv3x3_halo_s32_debug(0)
    def oneper():
        _lib.lib().ape_conv3x3_halo_s32_debug(2)
        conv(xs, out=out, out_fmt=E.FMT_S32)
        _lib.lib().ape_conv3x3_halo_s32_debug(0)
    res32 = E.S32.from_f32(torch.randn(b, h, w, cout, device="cuda"))

    def dbg(bits, **kw):
        def f():
            _lib.lib().ape_conv3x3_halo_s32_debug(bits)
            conv(xs, out=out, out_fmt=E.FMT_S32, **kw)
            _lib.lib().ape_conv3x3_halo_s32_debug(0)
        return f
    arms = {"halo (fp32 in)": lambda: conv(x, out=out), "halo_s32 -> f32": lambda: conv(xs, out=out), "halo_s32 -> s32": lambda: conv(xs, out=out, out_fmt=E.FMT_S32),
            "  -> s32, static priority": noprio, "  -> s32, one wg per tile": oneper,
            "  -> s32 + s32 residual": lambda: conv(xs, out=out, residual=res32, out_fmt=E.FMT_S32),
            "  ABLATION no stores": dbg(4), "  ABLATION res, no stores": dbg(4, residual=res32), "  ABLATION res not loaded": dbg(8, residual=res32)}
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
    flop = 2.0 * b * h * w * cin * cout * 9
    print("%s" % name)
    for k, t in times.items():
        t = sorted(t)
        print("   %-18s median %.3f ms  min %.3f ms   %.0f TFLOP/s algorithmic (%.2f of 833)" % (k, t[len(t) // 2], t[0], flop / t[len(t) // 2] / 1e9, flop / t[len(t) // 2] / 1e9 / 833.3))
