"""Where a tile of upconv_fused.hip spends its cycles: the stamped diagnostic build (make -C autoposeestimation_amd/csrc stamps), one launch at bench size.

    APE_HIP_LIB=autoposeestimation_amd/libape_hip_stamps.so python tools/stamp_upfuse.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E

B = 64
torch.manual_seed(0)
xs = E.S32.from_f32(torch.randn(B, 240, 320, 64, device="cuda"))
up = E.UpConv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), 0.25, device="cuda", precision="bf16x3", fma=True)
hw, hb = torch.randn(13, 64, device="cuda") / 8, torch.randn(13, device="cuda")
up.seg_head(xs, hw, hb, True, fused=True)
buf = torch.zeros(256 * 12 * 16, dtype=torch.int64, device="cuda")
_lib.lib().ape_upconv3x3_fused_stamps(buf.data_ptr())
up.seg_head(xs, hw, hb, True, fused=True)
torch.cuda.synchronize()
_lib.lib().ape_upconv3x3_fused_stamps(None)
t = buf.view(256, 12, 16)[:, :, :14].double().cpu()
names = ["loop head (decode, weights of the rows)", "row interpolation, top", "wait + barrier 1", "pixel requests A", "gather top: soft-max + stores", "matrix rows 6..9",
         "wait + barrier 2", "row interpolation, bottom", "wait + barrier 3", "requests B + gather bottom: soft-max + stores", "matrix rows 0..5 (next tile)", "wait + barrier 0",
         "both gathers: lane set-up (weights, offsets)", "both gathers: four 16-channel blocks (S reads, lerp, bias, act, head matrix instr)"]
tiles = 64 * 30 * 27 / 256.0
tot = t.sum(2).mean()
print("cycles per tile and wave (mean over 256 workgroups x 12 waves; %.1f tiles per workgroup), total %.0f" % (tiles, tot / tiles))
for i, n in enumerate(names):
    v = t[:, :, i]
    print("  %-42s %8.0f   (%4.1f %%)   min wave %8.0f  max wave %8.0f" % (n, v.mean() / tiles, 100 * v.mean() / tot, v.min() / tiles, v.max() / tiles))
