"""conv_gemm.hip block shapes (variant 1: 256x256, 2: 128x128, 3: 256x64, 4: 256x192, 6: 128x192) on the fp32-input layers of one bench step
that the S32 kernels do not take: which block shape is fastest per layer shape?  (engine.GEMM_VARIANT is read at call time.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E

# (name, B, H, W, Cin, Cout, k, stride, pad, dil)
SHAPES = [("layer2.0 conv1 (seg)", 64, 120, 160, 64, 128, 3, 2, 1, 1), ("layer2.0 down (seg)", 64, 120, 160, 64, 128, 1, 2, 0, 1),
          ("crops layer2 128->128", 64, 20, 20, 128, 128, 3, 1, 1, 1), ("crops layer3 256->256 d2", 64, 20, 20, 256, 256, 3, 1, 2, 2),
          ("crops layer3 128->256", 64, 20, 20, 128, 256, 3, 1, 1, 1), ("crops layer4 512->512 d4", 64, 20, 20, 512, 512, 3, 1, 4, 4),
          ("crops layer4 256->512", 64, 20, 20, 256, 512, 3, 1, 2, 2), ("crops psp 512->1024", 64, 20, 20, 512, 1024, 1, 1, 0, 1),
          ("crops up_1 mix 1024->2304", 64, 20, 20, 1024, 2304, 1, 1, 0, 1), ("crops up_2 mix 256->576", 64, 40, 40, 256, 576, 1, 1, 0, 1),
          ("crops layer1 64->64", 64, 40, 40, 64, 64, 3, 1, 1, 1)]
torch.manual_seed(0)
for name, b, h, w, cin, cout, k, st, pad, dil in SHAPES:
    x = torch.randn(b, h, w, cin, device="cuda")
    conv = E.Conv(torch.randn(cout, cin, k, k) / (k * k * cin) ** 0.5, None, st, pad, dil, E.ACT_RELU, device="cuda", precision="bf16x3")
    E.USE_HALO_KERNEL = False
    res = {}
    ref = None
    for v in (0, 1, 2, 3, 4, 6):
        E.GEMM_VARIANT = v
        try:
            y = conv(x)
        except Exception as e:      # noqa: BLE001
            res[v] = "n/a"
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        same = bool(torch.equal(y, ref))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            conv(x)
        e1.record()
        torch.cuda.synchronize()
        res[v] = "%.1f us%s" % (e0.elapsed_time(e1) * 100, "" if same else " (bits differ)")
    E.GEMM_VARIANT = 0
    print("%-28s" % name, "  ".join("v%d %s" % (v, r) for v, r in res.items()), flush=True)
