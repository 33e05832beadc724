"""timing ablations of halo_mx on the 512 -> 512 shape (results wrong under them)"""
import ctypes, os, sys
# the switches exist in the ablation build only: make -C autoposeestimation_amd/csrc ablations
os.environ.setdefault("APE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autoposeestimation_amd", "libape_hip_abl.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E
def t(f, n=5, rounds=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[len(ts) // 2]
b, h, w, cin, cout, dil = 64, 60, 80, 512, 512, int(os.environ.get("DIL", "4"))
conv = E.Conv(torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5), torch.randn(cout), pad=dil, dil=dil, act=E.ACT_RELU, device="cuda", precision="bf16x3")
xs = E.S32.from_f32(torch.relu(torch.randn(b, h, w, cin, device="cuda")))
out = E.S32(torch.empty(b, h, w, cout, device="cuda"))
xq = torch.empty_like(xs.t)
_lib.lib().ape_s32_to_f16m6(_lib.dptr(xs.t, torch.float32), _lib.dptr(xq, torch.float32), b * h * w, cin, None)
p = E.ConvParams(B=b, H=h, W=w, Cin=cin, ldx=cin, xoff=0, Ho=h, Wo=w, Cout=cout, ldy=cout, yoff=0, KH=3, KW=3, stride=1, pad=dil, dil=dil, act=E.ACT_RELU,
                 alpha=0.0, bias_bstride=0, ldr=0, roff=0, ups=0)
def run():
    _lib.check(_lib.lib().ape_conv3x3_halo_mx(_lib.dptr(xq, torch.float32), _lib.dptr(conv.mx6k()), _lib.dptr(conv.bias), None, 0, _lib.dptr(out.t, torch.float32),
                                              E.FMT_S32, ctypes.byref(p), None), "mx")
for bits, name in ((0, "as built"), (512, "first layout (chunks 4,5 | 6,7)"), (0, "as built again"), (512, "first layout again"), (32, "no cross terms"), (128, "no cross MFMAs (reads stay)"), (64, "no main MFMAs"), (64 | 128, "no MFMAs at all"), (64 | 32, "main reads only"),
                   (256, "no in-loop DMA"), (256 | 64 | 128, "reads + barriers only"), (4, "no stores"), (2, "one tile per workgroup"), (16, "four rows everywhere")):
    _lib.lib().ape_conv3x3_halo_mx_debug(bits)
    print("%-32s %.3f ms" % (name, t(run)))
_lib.lib().ape_conv3x3_halo_mx_debug(0)
