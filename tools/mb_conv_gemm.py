"""conv_gemm.hip on the pose stage's crop-map shapes (fp32 activations, register staging): per-shape medians.  A/B of a library variant:
run once with the product library and once with APE_HIP_LIB=autoposeestimation_amd/libape_hip_<variant>.so on the same box; also prints a
checksum of every output so that two arms can be compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
# (name, B, H, W, Cin, Cout, k, dil)
shapes = [("crop layer4 256->512 3x3", 64, 20, 20, 256, 512, 3, 1), ("crop layer4 512->512 3x3 d1", 64, 20, 20, 512, 512, 3, 1), ("crop layer4 512->512 3x3 d4", 64, 20, 20, 512, 512, 3, 4),
          ("crop psp 512->1024 1x1", 64, 20, 20, 512, 1024, 1, 1), ("crop up_1 mix 1024->2304 1x1", 64, 20, 20, 1024, 2304, 1, 1), ("crop up_2 mix 256->576", 64, 40, 40, 256, 576, 1, 1),
          ("crop layer3 256->256 3x3 d2", 64, 20, 20, 256, 256, 3, 2), ("crop layer2 128->128 3x3", 64, 20, 20, 128, 128, 3, 1),
          # the 4-wave blocks (128 x 128, 256 x 64): the crop's layers 1-2, the PointNet layers, the segmentor's 1/4-resolution stride-2 entry
          ("crop layer1 64->64 3x3", 64, 40, 40, 64, 64, 3, 1), ("crop layer2 64->128 3x3 (s1 here)", 64, 40, 40, 64, 128, 3, 1), ("pose e_conv2 64->128", 64, 1000, 1, 64, 128, 1, 1),
          ("pose conv2 64->128 / e_conv1 32->64", 64, 1000, 1, 32, 64, 1, 1), ("psp stage 512->512 6x6", 64, 6, 6, 512, 512, 1, 1), ("seg layer2 entry 64->128 3x3", 64, 60, 80, 64, 128, 3, 1)]
torch.manual_seed(0)
tot = 0.0
for name, b, h, w, cin, cout, k, dil in shapes:
    x = torch.randn(b, h, w, cin, device="cuda")
    wt = torch.randn(cout, cin, k, k) / (k * k * cin) ** 0.5 if k > 1 else torch.randn(cout, cin) / cin ** 0.5
    conv = E.Conv(wt, torch.randn(cout), 1, dil * (k // 2), dil, E.ACT_RELU, device="cuda", precision="bf16x3") if k > 1 else E.Conv(wt, torch.randn(cout), act=E.ACT_RELU, device="cuda", precision="bf16x3")
    out = torch.empty(b, h, w, cout, device="cuda")
    E.USE_HALO_KERNEL = False if k > 1 else E.USE_HALO_KERNEL      # (the crops' maps are what conv_gemm takes; keep the halo kernel out of this measurement)
    f = lambda: conv(x, out=out)
    f(); torch.cuda.synchronize()
    ts = []
    for rnd in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ts.sort()
    flop = 2.0 * b * h * w * cin * cout * k * k
    tot += ts[4]
    print("%-34s median %7.1f us  %4.0f TF/s (%.3f of 833)   checksum %016x" % (name, ts[4] * 1e3, flop / ts[4] / 1e9, flop / ts[4] / 1e9 / 833.3,
                                                                                  int(out.view(torch.int32).to(torch.int64).sum().item()) & 0xFFFFFFFFFFFFFFFF), flush=True)
print("sum of medians %.1f us" % (tot * 1e3))
