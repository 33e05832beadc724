"""Same-process A/B of gemm_s32's two schedules on the 1x1 layer shapes of the step: PP (round 6: the two waves of a SIMD run a k-tile's matrix
segment and its load segment in opposite order; all DMA from waves 4-7; two pixel slots + three weight slots) vs the lockstep form of rounds
3-5 (ape_conv_gemm_s32_debug bit 8192 selects the PP kernels on the HOST side; the lockstep form stays the product: PP is 8.6 % slower).  Interleaved rounds; also checks that both give the same bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
from autoposeestimation_amd import _lib

# (name, rows as B x H x W, Cin, Cout)
shapes = [("layer3 downsample 128->256", 64, 60, 80, 128, 256), ("layer4 downsample 256->512", 64, 60, 80, 256, 512), ("PSP bottleneck 576->1024", 64, 60, 80, 576, 1024),
          ("up_1 mix 1024->2304", 64, 60, 80, 1024, 2304), ("up_2 mix 256->576", 64, 120, 160, 256, 576), ("pose conv5 384->512", 64, 1000, 1, 384, 512),
          ("pose conv6 512->1024", 64, 1000, 1, 512, 1024), ("pose head 640->256", 64, 1000, 1, 640, 256), ("pose heads 384->1920", 64, 1000, 1, 384, 1920),
          ("pose 256->128", 64, 1000, 1, 256, 128)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]
torch.manual_seed(0)
tot = {"pp": 0.0, "old": 0.0}
for name, b, h, w, cin, cout in shapes:
    x = torch.randn(b, h, w, cin, device="cuda")
    xs = E.S32.from_f32(x)
    conv = E.Conv(torch.randn(cout, cin) / cin ** 0.5, torch.randn(cout), act=E.ACT_RELU, device="cuda", precision="bf16x3")
    outs = {k: torch.empty(b, h, w, cout, device="cuda") for k in ("pp", "old")}

    def arm(bits, key):
        def f():
            _lib.lib().ape_conv_gemm_s32_debug(bits)
            r = conv(xs, out=outs[key], out_fmt=E.FMT_S32)
            _lib.lib().ape_conv_gemm_s32_debug(0)
            return r
        return f
    arms = {"pp": arm(8192, "pp"), "old": arm(0, "old")}
    got = {k: f().to_f32().clone() for k, f in arms.items()}
    torch.cuda.synchronize()
    same = torch.equal(got["pp"], got["old"])
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
    print("%s   bitwise equal: %s" % (name, same), flush=True)
    for k, t in times.items():
        t = sorted(t)
        print("   %-6s median %.3f ms  min %.3f ms   %.0f TFLOP/s algorithmic (%.3f of 833)" % (k, t[len(t) // 2], t[0], flop / t[len(t) // 2] / 1e9, flop / t[len(t) // 2] / 1e9 / 833.3), flush=True)
    tot["pp"] += sorted(times["pp"])[3]
    tot["old"] += sorted(times["old"])[3]
print("sum of medians: pp %.3f ms, old %.3f ms" % (tot["pp"], tot["old"]))
