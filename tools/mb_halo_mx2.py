import ctypes, os, sys
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
b, h, w, cin, cout = 64, 60, 80, 512, 512
for dil in (1, 4):
    conv = E.Conv(torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5), torch.randn(cout), pad=dil, dil=dil, act=E.ACT_RELU, device="cuda", precision="bf16x3")
    for name, x in (("relu(randn)", torch.relu(torch.randn(b, h, w, cin, device="cuda"))), ("zeros", torch.zeros(b, h, w, cin, device="cuda"))):
        xs = E.S32.from_f32(x)
        res = E.S32.from_f32(torch.randn(b, h, w, cout, device="cuda"))
        out = E.S32(torch.empty(b, h, w, cout, device="cuda"))
        xq = torch.empty_like(xs.t)
        _lib.lib().ape_s32_to_f16m6(_lib.dptr(xs.t, torch.float32), _lib.dptr(xq, torch.float32), b * h * w, cin, None)
        p = E.ConvParams(B=b, H=h, W=w, Cin=cin, ldx=cin, xoff=0, Ho=h, Wo=w, Cout=cout, ldy=cout, yoff=0, KH=3, KW=3, stride=1, pad=dil, dil=dil, act=E.ACT_RELU,
                         alpha=0.0, bias_bstride=0, ldr=cout, roff=0, ups=0)
        def mx(r):
            _lib.check(_lib.lib().ape_conv3x3_halo_mx(_lib.dptr(xq, torch.float32), _lib.dptr(conv.mx6k()), _lib.dptr(conv.bias), _lib.dptr(r.t, torch.float32) if r is not None else None,
                                                      E.FMT_S32, _lib.dptr(out.t, torch.float32), E.FMT_S32, ctypes.byref(p), None), "mx")
        def s32(r):
            _lib.check(_lib.lib().ape_conv3x3_halo_s32(_lib.dptr(xs.t, torch.float32), _lib.dptr(conv.s32k()), _lib.dptr(conv.bias), _lib.dptr(r.t, torch.float32) if r is not None else None,
                                                       E.FMT_S32, _lib.dptr(out.t, torch.float32), E.FMT_S32, ctypes.byref(p), None), "s32")
        print("d%d %-12s halo_s32 %.3f / %.3f ms (residual / none)   halo_mx %.3f / %.3f ms" % (dil, name, t(lambda: s32(res)), t(lambda: s32(None)), t(lambda: mx(res)), t(lambda: mx(None))))
