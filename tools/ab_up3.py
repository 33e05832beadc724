"""up_3 + fused x2 up-sampling + head at bench size (64 frames, 240x320 -> 480x640): the one-role kernel of round 2
(conv3x3_halo_kernel<3,1,64,true,true>, the default) against the wave-specialised persistent kernel (up3_head_ws.hip, opt-in through
ape_up3_seghead_debug bit 1), interleaved rounds in one process; labels and scores compared bit for bit first."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
x = torch.randn(B, 240, 320, 64, device="cuda")
conv = E.Conv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), 1, 1, 1, E.ACT_PRELU, 0.25, device="cuda", precision="bf16x3")
hw, hb = torch.randn(13, 64, device="cuda") / 8, torch.randn(13, device="cuda")
flop = 2.0 * B * 480 * 640 * 64 * 64 * 9


def arm(bits):
    def f():
        _lib.lib().ape_up3_seghead_debug(bits)
        out = E.conv_seg_head(conv, x, hw, hb, True, upsample2x=True)
        _lib.lib().ape_up3_seghead_debug(0)
        return out
    return f


arms = {"one-role kernel (round 2)": arm(0), "wave-specialised": arm(2), "wave-specialised, matrix waves prio 1": arm(2 | 1),
        "wave-specialised, producer waves prio 3": arm(2 | 32)}
if len(sys.argv) > 2:
    arms.update({"ABLATION no halo building": arm(2 | 4), "ABLATION no MFMAs": arm(2 | 8), "ABLATION no head": arm(2 | 16), "ABLATION no MFMAs, no head": arm(2 | 24),
                 "ABLATION barriers + frag reads only": arm(2 | 28)})
outs = {k: f() for k, f in arms.items()}
torch.cuda.synchronize()
ref = outs["one-role kernel (round 2)"]
for k, o in outs.items():
    if "ABLATION" in k:
        continue
    print("%-40s labels equal %s  scores equal %s" % (k, torch.equal(o[0], ref[0]), torch.equal(o[1], ref[1])), flush=True)
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
    print("%-40s median %.3f ms  min %.3f ms   %.0f TFLOP/s algorithmic (%.2f of 833)" % (k, t[len(t) // 2], t[0], flop / t[len(t) // 2] / 1e9, flop / t[len(t) // 2] / 1e9 / 833.3))

# in-kernel stamps of the matrix waves (diagnostic build): where a tap's cycles go
if len(sys.argv) > 2:
    for bits, what in ((2, "full"), (2 | 4, "no halo building")):
        buf = torch.zeros(256 * 4 * 4, dtype=torch.int64, device="cuda")
        _lib.lib().ape_up3_seghead_stamps(buf.data_ptr())
        arm(bits)()
        torch.cuda.synchronize()
        _lib.lib().ape_up3_seghead_stamps(None)
        st = buf.view(256, 4, 4).double()
        tiles = st[..., 3].clamp(min=1)
        print("stamps (%s): per tile and matrix wave: burst %.0f cycles (%.0f per tap), wait + barrier %.0f (%.0f per tap), head %.0f" %
              (what, (st[..., 0] / tiles).mean(), (st[..., 0] / tiles).mean() / 18, (st[..., 1] / tiles).mean(), (st[..., 1] / tiles).mean() / 18, (st[..., 2] / tiles).mean()))
