"""upconv_fused.hip (up_3 + head at bench size) launched a few times for `rocprofv3 --pmc` passes, and its timing-only ablations:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \\
        --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_upfuse.py
    python tools/pmc_upfuse.py --reduce <dir>/*/*_counter_collection.csv
    python tools/pmc_upfuse.py --ablate          (no profiler: one line per ablation arm)"""
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
            if "upconv_fused" not in name:
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
xs = E.S32.from_f32(torch.randn(B, 240, 320, 64, device="cuda"))
up = E.UpConv(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), 0.25, device="cuda", precision="bf16x3", fma=True)
hw, hb = torch.randn(13, 64, device="cuda") / 8, torch.randn(13, device="cuda")


def timed(bits, n=5):
    _lib.lib().ape_upconv3x3_fused_debug(bits)
    up.seg_head(xs, hw, hb, True, fused=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        up.seg_head(xs, hw, hb, True, fused=True)
    e1.record()
    torch.cuda.synchronize()
    _lib.lib().ape_upconv3x3_fused_debug(0)
    return e0.elapsed_time(e1) / n


if len(sys.argv) > 1 and sys.argv[1] == "--ablate":
    arms = [(0, "full"), (1, "no matrix instructions"), (2, "no row interpolation / S stores"), (4, "no gather phases"), (8, "no soft-max / arg-max"),
            (16, "no pixel DMA"), (32, "no weight loads"), (1 | 32, "no matrix instr, no weight loads"), (2 | 4, "matrix phase + DMA only"),
            (1 | 2 | 4 | 32, "DMA + barriers only"), (1 | 2 | 32 | 16, "gather phases only"), (1 | 2 | 32 | 16 | 8, "gather phases without soft-max"),
            (1 | 4 | 32 | 16, "row interpolation only"), (51 | 64, "gather only, no S reads"), (51 | 128, "gather only, no head matrix instr"),
            (51 | 8 | 64 | 128, "gather only, no S reads / head matrix / soft-max"), (51 | 8 | 128, "gather only, no head matrix / soft-max")]
    for rnd in range(2):
        for bits, what in arms:
            print("round %d  bits %3d  %-40s %.3f ms" % (rnd, bits, what, timed(bits)), flush=True)
else:
    print("%.3f ms per launch" % timed(0, 3))
