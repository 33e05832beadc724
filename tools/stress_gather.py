"""The strip-walk gather (compiler-generated v_pk_mul_f32 / v_pk_add_f32 with an op_sel broadcast in SRC0; the form behind round 5's faults was a swizzled
src1) under the conditions that exposed those faults: hundreds of launches beside a second stream that keeps the chip busy, every output compared bit
for bit with the ROW kernel's (same arithmetic, no packed instructions: ape_upconv3x3_gather_strip_rows(0))."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "40")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
rounds = int(os.environ.get("ROUNDS", "20"))
xb = torch.randn(64, 60, 80, 512, generator=g).cuda()
busy_conv = E.Conv(torch.randn(512, 512, 3, 3, generator=g) / 68, torch.randn(512, generator=g), 1, 1, 1, E.ACT_RELU, device="cuda", precision="bf16x3")
xs = E.S32.from_f32(torch.relu(xb))
busy_out = E.S32(torch.empty(64, 60, 80, 512, device="cuda"))
xc = torch.randn(64, 20, 20, 512, generator=g).cuda()
crop_out = torch.empty(64, 20, 20, 512, device="cuda")
# (the persistent halo_s32 takes every CU: the two streams mostly alternate; the crop-sized conv_gemm launches -- 200 workgroups -- run BESIDE the gather)
neigh = {"halo_s32 512->512 on a second stream": lambda: busy_conv(xs, out=busy_out, out_fmt=E.FMT_S32),
         "crop-sized conv_gemm 3x3 x 4": lambda: [busy_conv(xc, out=crop_out) for _ in range(4)], "nothing": lambda: None}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
total_bad = 0
for (B, h, w, c, fmt, fma) in ((64, 60, 80, 256, E.FMT_S32, 0), (16, 120, 160, 64, E.FMT_S32, 0), (16, 60, 80, 256, E.FMT_F32, 0), (16, 60, 80, 256, E.FMT_S32, 1)):
    z = torch.randn(B, h, w, 9 * c, generator=g).cuda()
    bias = torch.randn(c, generator=g).cuda()
    def gather(out):
        rc = lib.ape_upconv3x3_gather_ex(_lib.dptr(z), _lib.dptr(bias), _lib.dptr(out), fmt, B, h, w, c, E.ACT_PRELU, 0.25, fma, _lib.stream_ptr())
        assert rc == 0
        return out
    old = lib.ape_upconv3x3_gather_strip_rows(0)
    want = gather(torch.empty(B, 2 * h, 2 * w, c, device="cuda")).clone()
    lib.ape_upconv3x3_gather_strip_rows(old)
    outs = [torch.empty(B, 2 * h, 2 * w, c, device="cuda") for _ in range(4)]
    for name, f in neigh.items():
        f(); torch.cuda.synchronize()
        bad = n = 0
        for r in range(rounds):
            with torch.cuda.stream(s2):
                for _ in range(2): f()
            with torch.cuda.stream(s1):
                for o in outs: gather(o)
            with torch.cuda.stream(s2):
                f()
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(o, want)) for o in outs)
            n += len(outs)
        total_bad += bad
        print("%dx%dx%d C%d fmt %d fma %d  beside %-40s wrong outputs: %d of %d" % (B, 2 * h, 2 * w, c, fmt, fma, name, bad, n), flush=True)
print("TOTAL wrong", total_bad)
sys.exit(1 if total_bad else 0)
