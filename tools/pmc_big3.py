"""The three largest kernels of a step launched a few times each, for `rocprofv3 --pmc ...` passes (counters in their own runs):
halo_s32<4> (layer4 512->512 d4), gemm_s32<256> (up_1 mix 1024->2304), up_3 + head in its low-resolution one-kernel form (upconv_fused.hip)
and, for reference, the direct form it replaced (conv3x3_halo_kernel<3,1,64,true,true>).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \\
              --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_big3.py
    python tools/pmc_big3.py --reduce <dir>/*/*_counter_collection.csv"""
import collections
import csv
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for path in sys.argv[2:]:
        seen = set()
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if not any(k in name for k in ("halo_s32", "gemm_s32", "conv3x3_halo_kernel", "upconv_fused")):
                continue
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (name, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                cnt[name] += 1
    for name, c in agg.items():
        n = cnt[name]
        print(name, "launches", n)
        for k, v in sorted(c.items()):
            print("    %-28s %.4g per launch" % (k, v / n))
        if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print("    matrix pipe busy %.3f of the kernel's cycles (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs); "
                  "vector instructions per matrix instruction %.2f" % ((c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (c["GRBM_GUI_ACTIVE"] / 8),
                                                                       c["SQ_INSTS_VALU"] / max(c["SQ_INSTS_MFMA"], 1)))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from autoposeestimation_amd import engine as E  # noqa: E402

torch.manual_seed(0)
B = 64
x = E.S32.from_f32(torch.randn(B, 60, 80, 512, device="cuda"))
c4 = E.Conv(torch.randn(512, 512, 3, 3) / 68, torch.randn(512), 1, 4, 4, E.ACT_RELU, device="cuda", precision="bf16x3")
o4 = torch.empty(B, 60, 80, 512, device="cuda")
x1 = E.S32.from_f32(torch.randn(B, 60, 80, 1024, device="cuda"))
cg = E.Conv(torch.randn(2304, 1024) / 32, torch.randn(2304), act=E.ACT_NONE, device="cuda", precision="bf16x3")
og = torch.empty(B, 60, 80, 2304, device="cuda")
xu = torch.randn(B, 240, 320, 64, device="cuda")
cu = E.Conv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), 1, 1, 1, E.ACT_PRELU, 0.25, device="cuda", precision="bf16x3")
hw, hb = torch.randn(13, 64, device="cuda") / 8, torch.randn(13, device="cuda")
uf = E.UpConv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), 0.25, device="cuda", precision="bf16x3", fma=True)
xus = E.S32.from_f32(xu)
for _ in range(3):
    c4(x, out=o4, out_fmt=E.FMT_S32)
    cg(x1, out=og)
    uf.seg_head(xus, hw, hb, True, fused=True)
    E.conv_seg_head(cu, xu, hw, hb, True, upsample2x=True)
torch.cuda.synchronize()
