"""same-process A/B of a halo_s32 debug bit (OFF_BITS, default 16 = four rows per wave-row group in every tile) on the segmentor's 3x3 layer
shapes; also checks that both forms give the same bits"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E
shapes = [("layer2 128ch d1", 64, 60, 80, 128, 128, 1), ("layer3 128->256 d1", 64, 60, 80, 128, 256, 1), ("layer3 256ch d2", 64, 60, 80, 256, 256, 2),
          ("layer4 256->512 d1", 64, 60, 80, 256, 512, 1), ("layer4 512ch d1", 64, 60, 80, 512, 512, 1), ("layer4 512ch d4", 64, 60, 80, 512, 512, 4)]
torch.manual_seed(0)
OFF = int(os.environ.get("OFF_BITS", "16"))
tot = {0: 0.0, OFF: 0.0}
for name, b, h, w, cin, cout, dil in shapes:
    xs = E.S32.from_f32(torch.randn(b, h, w, cin, device="cuda"))
    conv = E.Conv(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout), 1, dil, dil, E.ACT_RELU, device="cuda", precision="bf16x3")
    res32 = E.S32.from_f32(torch.randn(b, h, w, cout, device="cuda"))
    def run(bits):
        _lib.lib().ape_conv3x3_halo_s32_debug(bits)
        o = conv(xs, residual=res32, out_fmt=E.FMT_S32)
        _lib.lib().ape_conv3x3_halo_s32_debug(0)
        return o
    a, bb = run(0), run(OFF)
    torch.cuda.synchronize()
    same = torch.equal(a.t.view(torch.int32), bb.t.view(torch.int32))
    times = {0: [], OFF: []}
    for rnd in range(9):
        for bits in (0, OFF):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(bits)
            e1.record(); torch.cuda.synchronize()
            times[bits].append(e0.elapsed_time(e1) / 3)
    m = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    for k in m: tot[k] += m[k]
    flop = 2.0 * b * h * w * cin * cout * 9
    print("%-20s new %.3f ms (%.2f)   bit %d: %.3f ms (%.2f)   %+.1f %%   bitwise equal %s" % (name, m[0], flop / m[0] / 1e9 / 833.3, OFF, m[OFF], flop / m[OFF] / 1e9 / 833.3, (m[0] / m[OFF] - 1) * 100, same), flush=True)
print("sum  new %.3f ms   bit %d: %.3f ms" % (tot[0], OFF, tot[OFF]))
