"""same-process A/B of upconv_fused builds: APE_LIBS=a.so,b.so python tools/ab_upfuse.py  (interleaved rounds, median / min per build)"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E
libs = os.environ["APE_LIBS"].split(",")
hs = [ctypes.CDLL(os.path.abspath(p)) for p in libs]
g = torch.Generator().manual_seed(5)
w = torch.randn(64, 64, 3, 3, generator=g) / 24
b = torch.randn(64, generator=g)
up = E.UpConv(w, b, 0.25, device="cuda", precision="bf16x3", fma=True)
B, h, wd = 64, 240, 320
xs = E.S32.from_f32((torch.randn(B, h, wd, 64, generator=g) * 2).cuda())
hw = (torch.randn(13, 64, generator=g) / 8).cuda()
hb = torch.randn(13, generator=g).cuda()
label = torch.empty(B, 2 * h, 2 * wd, dtype=torch.uint8, device="cuda")
score = torch.empty(B, 2 * h, 2 * wd, dtype=torch.float32, device="cuda")
ws = up.mix.s32k()
P = ctypes.c_void_p
def call(hl):
    f = hl.ape_upconv3x3_fused_seghead_s32
    f.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, P, P, ctypes.c_int, P, P, ctypes.c_int, P]
    rc = f(xs.t.data_ptr(), ws.data_ptr(), up.bias.data_ptr(), B, h, wd, 64, E.ACT_PRELU, 0.25, 1, hw.data_ptr(), hb.data_ptr(), 13, label.data_ptr(), score.data_ptr(), 1, _lib.stream_ptr())
    assert rc == 0
times = [[] for _ in hs]
outs = []
for i, hl in enumerate(hs):
    call(hl); torch.cuda.synchronize(); outs.append((label.clone(), score.clone()))
for rnd in range(12):
    for i, hl in enumerate(hs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call(hl)
        e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 5)
for p, t, o in zip(libs, times, outs):
    print("%-55s median %.4f ms  min %.4f ms   equal to first: %s" % (os.path.basename(p), statistics.median(t), min(t), bool(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]))))
