import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from autoposeestimation_amd import _lib, engine as E
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ab_s32 import bind, OLD
old, new = bind(OLD), _lib.lib()
torch.manual_seed(0)
for (b, h, w, cin, cout, dil) in [(1, 16, 16, 32, 128, 1), (2, 24, 40, 64, 128, 1), (8, 60, 80, 64, 512, 1)]:
    x = torch.randn(b, h, w, cin, device="cuda")
    xs = E.S32.from_f32(x)
    conv = E.Conv(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout), 1, dil, dil, E.ACT_RELU, device="cuda", precision="bf16x3")
    resf = torch.randn(b, h, w, cout, device="cuda")
    ress = E.S32.from_f32(resf)
    p = E.ConvParams(B=b, H=h, W=w, Cin=cin, ldx=cin, xoff=0, Ho=h, Wo=w, Cout=cout, ldy=cout, yoff=0, KH=3, KW=3, stride=1, pad=dil, dil=dil, act=E.ACT_RELU,
                     alpha=0.0, bias_bstride=0, ldr=cout, roff=0, ups=0)
    for res, rfmt, rname in ((None, 0, "no res"), (resf, 0, "f32 res"), (ress.t, 1, "s32 res")):
        for ofmt in (0, 1):
            outs = []
            for lib in (old, new):
                o = torch.zeros(b, h, w, cout, device="cuda")
                rc = lib.ape_conv3x3_halo_s32(_lib.dptr(xs.t, torch.float32), _lib.dptr(conv.s32k()), _lib.dptr(conv.bias), _lib.dptr(res) if res is not None else None, rfmt,
                                              _lib.dptr(o), ofmt, ctypes.byref(p), _lib.stream_ptr())
                assert rc == 0
                outs.append(o)
            torch.cuda.synchronize()
            eq = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
            nd = (outs[0].view(torch.int32) != outs[1].view(torch.int32)).sum().item()
            print((b, h, w, cin, cout, dil), rname, "out", "s32" if ofmt else "f32", "equal" if eq else "DIFF %d of %d" % (nd, outs[0].numel()))
            if not eq and ofmt == 0:
                d = (outs[0] != outs[1]).nonzero()[:6]
                print("   first diffs (b,y,x,c):", d.tolist(), outs[0][tuple(d[0])].item(), outs[1][tuple(d[0])].item())
