"""Per-layer conv timing (development aid): generic bf16 kernel vs LDS-halo kernel at the bench's segmentation shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
shapes = [("layer1", 120, 160, 64, 64, 1), ("layer2", 60, 80, 128, 128, 1), ("layer3.0", 60, 80, 256, 256, 1), ("layer3.1", 60, 80, 256, 256, 2),
          ("layer4.0b", 60, 80, 512, 512, 1), ("layer4.1", 60, 80, 512, 512, 4), ("up_3", 480, 640, 64, 64, 1)]
for name, h, w, cin, cout, d in shapes:
    x = torch.randn(B, h, w, cin, device="cuda")
    wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    conv = E.Conv(wt, None, 1, d, d, E.ACT_RELU, device="cuda", precision=PREC)
    out = torch.empty(B, h, w, cout, device="cuda")
    res = {}
    for halo in (False, True):
        E.USE_HALO_KERNEL = halo
        for _ in range(2): conv(x, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): conv(x, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res[halo] = (ms, out.clone())
    flop = 2.0 * B * h * w * cout * cin * 9
    diff = (res[True][1] - res[False][1]).abs().max().item() / res[False][1].abs().max().item()
    print("%-10s %s generic %7.3f ms %6.1f TF/s | halo %7.3f ms %6.1f TF/s | rel diff %.2e" % (name, PREC, res[False][0], flop / res[False][0] / 1e9, res[True][0], flop / res[True][0] / 1e9, diff))
