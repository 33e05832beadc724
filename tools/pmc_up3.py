"""up_3 + head (the one-role kernel or, APE_UP3_BITS=2, up3_head_ws.hip; +4 / +8 / +16: its timing-only ablations) launched a few times for `rocprofv3 --pmc` passes:

    APE_UP3_BITS=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \\
        --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_up3.py
    python tools/pmc_up3.py --reduce <dir>/*/*_counter_collection.csv"""
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
            if not any(k in name for k in ("up3_head_ws", "conv3x3_halo_kernel", "halo_s32", "gemm_s32")):
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
            print("    matrix pipe busy %.3f of the kernel's cycles (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs)"
                  % ((c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (c["GRBM_GUI_ACTIVE"] / 8)))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from autoposeestimation_amd import _lib, engine as E  # noqa: E402

torch.manual_seed(0)
B = 64
xu = torch.randn(B, 240, 320, 64, device="cuda")
cu = E.Conv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), 1, 1, 1, E.ACT_PRELU, 0.25, device="cuda", precision="bf16x3")
hw, hb = torch.randn(13, 64, device="cuda") / 8, torch.randn(13, device="cuda")
_lib.lib().ape_up3_seghead_debug(int(os.environ.get("APE_UP3_BITS", "0")))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
E.conv_seg_head(cu, xu, hw, hb, True, upsample2x=True)
torch.cuda.synchronize()
e0.record()
for _ in range(3):
    E.conv_seg_head(cu, xu, hw, hb, True, upsample2x=True)
e1.record()
torch.cuda.synchronize()
print("bits %s: %.3f ms per launch" % (os.environ.get("APE_UP3_BITS", "0"), e0.elapsed_time(e1) / 3))
