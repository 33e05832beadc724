"""ape_upconv3x3_gather_ex at the benchmark's two shapes (up_1: 64x60x80 -> 120x160 C256, up_2: 64x120x160 -> 240x320 C64, S32 outputs):
the one-row kernel against the strip walk at several strip heights.  HIP-event time per launch, GB/s of algorithmic bytes (z once + out once)."""
import sys

import torch

sys.path.insert(0, ".")
from autoposeestimation_amd import _lib, engine as E  # noqa: E402

lib = _lib.lib()
B = 64
for (h, w, c) in ((60, 80, 256), (120, 160, 64)):
    z = torch.randn(B, h, w, 9 * c, device="cuda")
    bias = torch.randn(c, device="cuda")
    out = torch.empty(B, 2 * h, 2 * w, c, dtype=torch.float32, device="cuda")
    gb = 4.0 * (z.numel() + out.numel()) / 1e9
    for rows in (0, 8, 16, 24, 30, 40, 60, 120, 240):
        if rows > 2 * h:
            continue
        lib.ape_upconv3x3_gather_strip_rows(rows)
        for fmt in (E.FMT_S32,):
            def call():
                rc = lib.ape_upconv3x3_gather_ex(_lib.dptr(z), _lib.dptr(bias), _lib.dptr(out), fmt, B, h, w, c, E.ACT_PRELU, 0.25, 0, _lib.stream_ptr())
                assert rc == 0
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("h%d w%d C%d rows=%3d: %.3f ms  %.0f GB/s" % (h, w, c, rows, ms, gb / ms * 1e3), flush=True)
    del z, out
lib.ape_upconv3x3_gather_strip_rows(30)
