"""conv_gemm.hip against conv_bf16.hip at every GEMM-shaped layer the bench sends through them (development aid):
    python tools/microbench_gemm.py [bf16x3|bf16]
One process, interleaved rounds (cdna_hip_programming.md 5.4 rule 24); prints the median of 5 rounds per arm."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
E.USE_HALO_KERNEL = False
# name, B, H, W, Cin, Cout, k, stride, pad, dil
shapes = [
    ("seg l2 down 1x1s2", 64, 120, 160, 64, 128, 1, 2, 0, 1),
    ("seg l2.0 3x3s2", 64, 120, 160, 64, 128, 3, 2, 1, 1),
    ("seg l3 down 1x1", 64, 60, 80, 128, 256, 1, 1, 0, 1),
    ("seg l4 down 1x1", 64, 60, 80, 256, 512, 1, 1, 0, 1),
    ("seg psp bott 1x1", 64, 60, 80, 512, 1024, 1, 1, 0, 1),
    ("seg up1 mix 1x1", 64, 60, 80, 1024, 2304, 1, 1, 0, 1),
    ("seg up2 mix 1x1", 64, 120, 160, 256, 576, 1, 1, 0, 1),
    ("pose l1 3x3", 64, 40, 40, 64, 64, 3, 1, 1, 1),
    ("pose l2 3x3", 64, 20, 20, 128, 128, 3, 1, 1, 1),
    ("pose l3 3x3 d2", 64, 20, 20, 256, 256, 3, 1, 2, 2),
    ("pose l4 3x3 d4", 64, 20, 20, 512, 512, 3, 1, 4, 4),
    ("pose psp bott", 64, 20, 20, 512, 1024, 1, 1, 0, 1),
    ("pose up1 mix", 64, 20, 20, 1024, 2304, 1, 1, 0, 1),
    ("pose up2 mix", 64, 40, 40, 256, 576, 1, 1, 0, 1),
    ("pose conv2 64->128", 64, 1000, 1, 64, 128, 1, 1, 0, 1),
    ("pose conv5 256->512", 64, 1000, 1, 256, 512, 1, 1, 0, 1),
    ("pose conv6 512->1024", 64, 1000, 1, 512, 1024, 1, 1, 0, 1),
    ("pose heads l1 384->1920", 64, 1000, 1, 384, 1920, 1, 1, 0, 1),
    ("pose heads l2 640->256", 64, 1000, 1, 640, 256, 1, 1, 0, 1),
    ("pose heads l3 256->128", 64, 1000, 1, 256, 128, 1, 1, 0, 1),
]
arms = [("old", False, 0), ("g256", True, 1), ("g128", True, 2), ("g256x64", True, 3), ("g256x192", True, 4), ("g256pp", True, 5)]
for name, b, h, w, cin, cout, k, s, p, d in shapes:
    x = torch.randn(b, h, w, cin, device="cuda")
    wt = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    conv = E.Conv(wt, None, s, p, d, E.ACT_RELU, device="cuda", precision=PREC)
    ho, wo = conv.out_hw(h, w)
    out = torch.empty(b, ho, wo, cout, device="cuda")
    times = {a[0]: [] for a in arms}
    for rnd in range(6):
        for aname, use, var in arms:
            E.USE_GEMM_KERNEL, E.GEMM_VARIANT = use, var
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): conv(x, out=out)
            e1.record(); torch.cuda.synchronize()
            if rnd: times[aname].append(e0.elapsed_time(e1) / 3)
    flop = 2.0 * b * ho * wo * cout * cin * k * k
    line = "%-24s M=%8d N=%5d K=%5d " % (name, b * ho * wo, cout, cin * k * k)
    for aname, _, _ in arms:
        ms = sorted(times[aname])[len(times[aname]) // 2]
        line += " %s %7.3f ms %5.0f TF |" % (aname, ms, flop / ms / 1e9)
    print(line, flush=True)
