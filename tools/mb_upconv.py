"""ape_upconv3x3_gather_f32 at the up_1 / up_2 shapes (development aid)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E
L = _lib.lib()
for name, b, h, w, c in [("up_1", 64, 60, 80, 256), ("up_2", 64, 120, 160, 64), ("crop up_1", 64, 20, 20, 256), ("crop up_2", 64, 40, 40, 64)]:
    z = torch.randn(b, h, w, 9 * c, device="cuda")
    bias = torch.randn(c, device="cuda")
    out = torch.empty(b, 2 * h, 2 * w, c, device="cuda")
    fn = L.ape_upconv3x3_gather_f32
    for _ in range(2): fn(_lib.dptr(z), _lib.dptr(bias), _lib.dptr(out), b, h, w, c, 2, ctypes.c_float(0.25), _lib.stream_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn(_lib.dptr(z), _lib.dptr(bias), _lib.dptr(out), b, h, w, c, 2, ctypes.c_float(0.25), _lib.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-10s %.3f ms (%.2f TB/s)" % (name, ms, (z.numel() + out.numel()) * 4 / 1e9 / ms))
