"""timing ablations of halo_s32's PP schedule on the 512 -> 512 shape (results wrong under them; ablation build only)"""
import ctypes, os, sys
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
p = E.ConvParams(B=b, H=h, W=w, Cin=cin, ldx=cin, xoff=0, Ho=h, Wo=w, Cout=cout, ldy=cout, yoff=0, KH=3, KW=3, stride=1, pad=dil, dil=dil, act=E.ACT_RELU,
                 alpha=0.0, bias_bstride=0, ldr=0, roff=0, ups=0)
def run():
    _lib.check(_lib.lib().ape_conv3x3_halo_s32(_lib.dptr(xs.t, torch.float32), _lib.dptr(conv.s32k()), _lib.dptr(conv.bias), None, 0, _lib.dptr(out.t, torch.float32),
                                               E.FMT_S32, ctypes.byref(p), None), "s32")
for bits, name in ((0, "pp as built"), (4096, "old schedule"), (64, "no fragment-read waits"), (256, "no DMA wait"), (512, "no in-loop DMA"), (2048, "no fragment reads"),
                   (2048 | 512, "no reads, no DMA (C + bookkeeping + barriers)"), (1024, "no MFMAs"), (1024 | 512, "no MFMAs, no DMA"), (1024 | 2048, "no MFMAs, no reads"),
                   (1024 | 2048 | 512, "bookkeeping + waits + barriers only"), (128, "no barriers (races; timing only)"), (32, "half the MFMAs (rows 0, 1 only)"), (32 | 512, "half the MFMAs, no DMA"), (4, "no stores"), (0, "pp as built again")):
    _lib.lib().ape_conv3x3_halo_s32_debug(bits)
    print("%-50s %.3f ms" % (name, t(run)), flush=True)
_lib.lib().ape_conv3x3_halo_s32_debug(0)
