"""Same-process A/B of the S32 conv kernels: the library of the previous round (tools/attic/probes/libape_hip_r2.so, built from the round-2
commit by `git archive dc986cc autoposeestimation_amd/csrc include` + make) against the current one, interleaved rounds, the segmentor's
own layer shapes at bench size.  Both libraries are driven through the same C ABI entry points with the same buffers; outputs are
compared bit for bit before timing."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from autoposeestimation_amd import _lib, engine as E  # noqa: E402

OLD = os.environ.get("APE_AB_OLD") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "attic", "probes", "libape_hip_r2.so")


def bind(path):
    lib = ctypes.CDLL(path)
    for name in ("ape_conv3x3_halo_s32", "ape_conv_gemm_s32"):
        getattr(lib, name).argtypes = _lib.SIGNATURES[name]
        getattr(lib, name).restype = ctypes.c_int
    return lib


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "halo"
    old, new = bind(OLD), _lib.lib()
    b = 64
    if which == "halo":
        shapes = [("layer2 128->128 d1", 60, 80, 128, 128, 3, 1), ("layer3 128->256 d1", 60, 80, 128, 256, 3, 1), ("layer3 256->256 d1", 60, 80, 256, 256, 3, 1),
                  ("layer3 256->256 d2", 60, 80, 256, 256, 3, 2), ("layer4 256->512 d1", 60, 80, 256, 512, 3, 1), ("layer4 512->512 d1", 60, 80, 512, 512, 3, 1),
                  ("layer4 512->512 d4", 60, 80, 512, 512, 3, 4)]
    else:
        shapes = [("ds 128->256", 60, 80, 128, 256, 1, 1), ("ds 256->512", 60, 80, 256, 512, 1, 1), ("psp 512->1024 +res", 60, 80, 512, 1024, 1, 1),
                  ("up_1 mix 1024->2304", 60, 80, 1024, 2304, 1, 1), ("up_2 mix 256->576", 120, 160, 256, 576, 1, 1)]
    torch.manual_seed(0)
    for name, h, w, cin, cout, k, dil in shapes:
        x = torch.randn(b, h, w, cin, device="cuda")
        xs = E.S32.from_f32(x)
        wt = torch.randn(cout, cin, k, k) / (k * k * cin) ** 0.5
        conv = E.Conv(wt, torch.randn(cout), 1, dil if k == 3 else 0, dil, E.ACT_RELU, device="cuda", precision="bf16x3")
        res = E.S32.from_f32(torch.randn(b, h, w, cout, device="cuda"))
        out_fmt = E.FMT_S32 if cout % 32 == 0 and "mix" not in name else E.FMT_F32
        use_res = k == 3 or "+res" in name
        outs = [torch.empty(b, h, w, cout, device="cuda") for _ in range(2)]
        p = E.ConvParams(B=b, H=h, W=w, Cin=cin, ldx=cin, xoff=0, Ho=h, Wo=w, Cout=cout, ldy=cout, yoff=0, KH=k, KW=k, stride=1, pad=dil if k == 3 else 0,
                         dil=dil, act=E.ACT_RELU, alpha=0.0, bias_bstride=0, ldr=cout if use_res else 0, roff=0, ups=0)
        fn = "ape_conv3x3_halo_s32" if k == 3 else "ape_conv_gemm_s32"

        def run(lib, out):
            rc = getattr(lib, fn)(_lib.dptr(xs.t, torch.float32), _lib.dptr(conv.s32k()), _lib.dptr(conv.bias), _lib.dptr(res.t) if use_res else None, E.FMT_S32,
                                  _lib.dptr(out), out_fmt, ctypes.byref(p), _lib.stream_ptr())
            assert rc == 0, rc
        arms = {"round 2": lambda: run(old, outs[0]), "now": lambda: run(new, outs[1])}
        for f in arms.values():
            f()
        torch.cuda.synchronize()
        same = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
        times = {kk: [] for kk in arms}
        for rnd in range(9):
            for kk, f in arms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    f()
                e1.record()
                torch.cuda.synchronize()
                times[kk].append(e0.elapsed_time(e1) / 3)
        flop = 2.0 * b * h * w * cin * cout * k * k
        line = "%-22s bitwise equal: %s" % (name, same)
        for kk, t in times.items():
            t = sorted(t)
            line += "   %s median %.3f min %.3f ms (%.2f of 833)" % (kk, t[len(t) // 2], t[0], flop / t[len(t) // 2] / 1e9 / 833.3)
        print(line, flush=True)


if __name__ == "__main__":
    main()
