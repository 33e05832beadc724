import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
B = 64; PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
for name, h, w, cin, cout in [("up_1 mix", 60, 80, 1024, 2304), ("up_2 mix", 120, 160, 256, 576), ("bott_feats", 60, 80, 512, 1024), ("final", 480, 640, 64, 13), ("l1_point", 1000, 1, 384, 1920)]:
    x = torch.randn(B, h, w, cin, device="cuda")
    conv = E.Conv(torch.randn(cout, cin) / cin ** 0.5, torch.randn(cout), act=E.ACT_RELU, device="cuda", precision=PREC)
    out = torch.empty(B, h, w, cout, device="cuda")
    for _ in range(2): conv(x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): conv(x, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flop = 2.0 * B * h * w * cout * cin
    gb = (x.numel() + out.numel()) * 4 / 1e9
    print("%-10s %s %7.3f ms %6.1f TF/s  in+out %.2f GB -> %.2f TB/s" % (name, PREC, ms, flop / ms / 1e9, gb, gb / ms))
