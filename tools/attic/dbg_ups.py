"""Diagnostic for the fused x2 up-sampling halo kernels (development aid, not product).

Round 1 saw the per-tile LDS table `ups_tbl` return wrong blend weights when ups_lerp re-read its entry at the END of a tap on
grids larger than the chip.  `make -C autoposeestimation_amd/csrc dbgups` builds libape_hip_dbgups.so, in which ups_lerp does
that second read again and records every disagreement with the register-carried first read.  This script drives it:

    APE_HIP_LIB=autoposeestimation_amd/libape_hip_dbgups.so python tools/dbg_ups.py

and, with the product library, only checks fused == materialised bit for bit on the same grids."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from autoposeestimation_amd import engine as E, _lib

lib = _lib.lib()
dbg = hasattr(lib, "ape_ups_debug_read")
print("library:", _lib.LIB_PATH, "debug build" if dbg else "product build", flush=True)


def read_dbg(reset=True):
    buf = (ctypes.c_uint * (1 + 16 * 64))()
    lib.ape_ups_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.ape_ups_debug_read(buf, int(reset)) == 0
    a = np.frombuffer(buf, dtype=np.uint32).copy()
    return int(a[0]), a[1:].reshape(64, 16)


torch.manual_seed(0)
bad = 0
for cin, cout in [(64, 64), (64, 128)]:
    for prec in ("bf16x3", "bf16"):
        conv = E.Conv(torch.randn(cout, cin, 3, 3) / 24, torch.randn(cout), 1, 1, 1, E.ACT_PRELU, 0.25, device="cuda", precision=prec)
        hw, hb = torch.randn(13, 64, device="cuda") / 8, torch.randn(13, device="cuda")
        for B, h, w in [(1, 24, 40), (1, 120, 160), (8, 240, 320), (3, 136, 104), (16, 240, 320)]:
            x = torch.randn(B, h, w, cin, device="cuda")
            ref = conv(E.bilinear(x, 2 * h, 2 * w, True))
            if cout == 64:
                rl, rs = E.seg_head(ref, hw, hb, True)
            for it in range(3):
                if dbg:
                    read_dbg(True)
                y = conv(x, upsample2x=True)
                ok = torch.equal(y, ref)
                if cout == 64:
                    l, s = E.conv_seg_head(conv, x, hw, hb, True, upsample2x=True)
                    ok = ok and torch.equal(l, rl) and torch.equal(s, rs)
                torch.cuda.synchronize()
                msg = ""
                if dbg:
                    n, rec = read_dbg(True)
                    msg = " table re-read mismatches: %d" % n
                    for r in rec[:min(n, 6)]:
                        msg += ("\n      blk %d tid %d j %d px %d | 2nd read: x=%d y=%d lx=%g ly=%g | 1st read: lx=%g ly=%g ok=%g | tile y0=%d x0=%d b=%d hw_id=%08x xcc=%d"
                                % (r[0], r[1], r[2], r[3], r[4], r[5], r[6:7].view(np.float32)[0], r[7:8].view(np.float32)[0],
                                   r[8:9].view(np.float32)[0], r[9:10].view(np.float32)[0], r[10:11].view(np.float32)[0], r[11], r[12], r[13], r[14], r[15]))
                bad += 0 if ok else 1
                print(cin, cout, prec, (B, h, w), it, "OK" if ok else "MISMATCH", msg, flush=True)
print("output mismatches:", bad)
