"""Same-process A/B of halo_s32's two schedules on the segmentor's 3x3 layer shapes: PP (round 6: the two waves of a SIMD alternate between a
matrix slot and a load slot, two barriers per tap, waves 4-7 one slot behind) vs the one-barrier-per-tap form of rounds 2-5
(ape_conv3x3_halo_s32_debug bit 4096 selects the old kernel on the HOST side: no switch inside either kernel).  Interleaved rounds; also checks
that both give the same bits (S32 output, with and without an S32 residual)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
from autoposeestimation_amd import _lib

shapes = [("layer2 128ch d1", 64, 60, 80, 128, 128, 1), ("layer3 128->256 d1", 64, 60, 80, 128, 256, 1), ("layer3 256ch d1", 64, 60, 80, 256, 256, 1),
          ("layer3 256ch d2", 64, 60, 80, 256, 256, 2), ("layer4 256->512 d1", 64, 60, 80, 256, 512, 1), ("layer4 512ch d1", 64, 60, 80, 512, 512, 1),
          ("layer4 512ch d4", 64, 60, 80, 512, 512, 4)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]
torch.manual_seed(0)
tot = {"pp": 0.0, "old": 0.0}
for name, b, h, w, cin, cout, dil in shapes:
    x = torch.randn(b, h, w, cin, device="cuda")
    xs = E.S32.from_f32(x)
    conv = E.Conv(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout), 1, dil, dil, E.ACT_RELU, device="cuda", precision="bf16x3")
    res32 = E.S32.from_f32(torch.randn(b, h, w, cout, device="cuda"))
    outs = {k: torch.empty(b, h, w, cout, device="cuda") for k in ("pp", "old", "ppr", "oldr")}

    def arm(bits, key, **kw):
        def f():
            _lib.lib().ape_conv3x3_halo_s32_debug(bits)
            r = conv(xs, out=outs[key], out_fmt=E.FMT_S32, **kw)
            _lib.lib().ape_conv3x3_halo_s32_debug(0)
            return r
        return f
    arms = {"pp": arm(0, "pp"), "old": arm(4096, "old"), "pp, one channel tile per XCD": arm(8, "old"), "pp + s32 residual": arm(0, "ppr", residual=res32), "old + s32 residual": arm(4096, "oldr", residual=res32)}
    got = {k: f().to_f32().clone() for k, f in arms.items()}
    torch.cuda.synchronize()
    same = torch.equal(got["pp"], got["old"]) and torch.equal(got["pp + s32 residual"], got["old + s32 residual"])
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
    print("%s   bitwise equal: %s" % (name, same), flush=True)
    for k, t in times.items():
        t = sorted(t)
        print("   %-20s median %.3f ms  min %.3f ms   %.0f TFLOP/s algorithmic (%.3f of 833)" % (k, t[len(t) // 2], t[0], flop / t[len(t) // 2] / 1e9, flop / t[len(t) // 2] / 1e9 / 833.3), flush=True)
    tot["pp"] += sorted(times["pp"])[3]
    tot["old"] += sorted(times["old"])[3]
print("sum of medians: pp %.3f ms, old %.3f ms" % (tot["pp"], tot["old"]))
