"""One conv shape, one kernel variant, N launches: the process to put under `rocprofv3 --pmc ...` (development aid).
    python3 tools/pmc_gemm.py <variant 0..5 | old> [B H W Cin Cout k stride pad dil]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E
var = sys.argv[1]
shape = [int(v) for v in sys.argv[2:11]] if len(sys.argv) >= 11 else [64, 60, 80, 1024, 2304, 1, 1, 0, 1]
b, h, w, cin, cout, k, s, p, d = shape
E.USE_HALO_KERNEL = os.environ.get("APE_PMC_HALO", "0") == "1"
E.USE_GEMM_KERNEL = var != "old"
E.GEMM_VARIANT = 0 if var == "old" else int(var)
x = torch.randn(b, h, w, cin, device="cuda")
wt = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
conv = E.Conv(wt, None, s, p, d, E.ACT_RELU, device="cuda", precision="bf16x3")
ho, wo = conv.out_hw(h, w)
out = torch.empty(b, ho, wo, cout, device="cuda")
for _ in range(3):
    conv(x, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    conv(x, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("variant %s dbg %d: %.3f ms  %.0f TF/s" % (var, E.GEMM_VARIANT >> 4, ms, 2.0 * b * ho * wo * cout * cin * k * k / ms / 1e9))
