"""Stress run of the one-kernel up_3 (+ head) against its two- / three-call form: many back-to-back launches at bench-like sizes while a second
stream keeps the chip busy with other work (the default bench runs the pose stage beside it), every repetition compared BIT FOR BIT.
    APE_HIP_LIB=autoposeestimation_amd/libape_hip_asmmath.so python tools/stress_upfuse.py --reps 50
prints one line per (arithmetic, shape): launches, launches with a wrong pixel, wrong pixels in all, worst launch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from autoposeestimation_amd import engine as E  # noqa: E402


def busy_work(stream, stop_after):
    """enqueue `stop_after` medium-size convolutions on `stream` (a 3x3 256 -> 256 layer on a 4 x 60 x 80 map: ~0.1 ms each, 150 workgroups --
    it leaves CUs for the kernel under test and keeps changing which ones)"""
    g = torch.Generator().manual_seed(99)
    conv = E.Conv(torch.randn(256, 256, 3, 3, generator=g) / 48, None, 1, 1, 1, E.ACT_RELU, precision="bf16x3")
    x = torch.randn(4, 60, 80, 256, generator=g).cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(stop_after):
            conv(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--shapes", default="2x240x320,64x240x320")
    ap.add_argument("--fma", default="1,0")
    ap.add_argument("--no-busy", action="store_true")
    args = ap.parse_args()
    side = torch.cuda.Stream()
    bad_total = 0
    for fma in [bool(int(v)) for v in args.fma.split(",")]:
        g = torch.Generator().manual_seed(5)
        w = torch.randn(64, 64, 3, 3, generator=g) / 24
        b = torch.randn(64, generator=g)
        up = E.UpConv(w, b, 0.25, device="cuda", precision="bf16x3", fma=fma)
        for shp in args.shapes.split(","):
            B, h, wd = [int(v) for v in shp.split("x")]
            g = torch.Generator().manual_seed(7 + 131 * h + wd)
            xs = E.S32.from_f32((torch.randn(B, h, wd, 64, generator=g) * 2).cuda())
            g = torch.Generator().manual_seed(13)
            hw = (torch.randn(13, 64, generator=g) / 8).cuda()
            hb = torch.randn(13, generator=g).cuda()
            want_l, want_s = up.seg_head(xs, hw, hb, True, fused=False)
            want_a = up(xs, fused=False) if B <= 8 else None
            torch.cuda.synchronize()
            if not args.no_busy:
                busy_work(side, 40 * args.reps)
            outs = []
            for rep in range(args.reps):
                gl, gs = up.seg_head(xs, hw, hb, True, fused=True)
                ga = up(xs, fused=True) if want_a is not None else None
                nbad = ((gs != want_s) | (gl != want_l)).sum()
                abad = (ga != want_a).sum() if ga is not None else torch.zeros((), dtype=torch.int64, device="cuda")
                outs.append(torch.stack([nbad, abad]))
            res = torch.stack(outs).cpu()
            torch.cuda.synchronize()
            hb_, ab_ = res[:, 0], res[:, 1]
            bad_total += int(hb_.sum()) + int(ab_.sum())
            print("fma %d  %-12s launches %d  head: %d wrong launches, %d wrong pixels (worst %d)   store: %d wrong launches, %d wrong values" %
                  (fma, shp, args.reps, int((hb_ > 0).sum()), int(hb_.sum()), int(hb_.max()), int((ab_ > 0).sum()), int(ab_.sum())), flush=True)
    print("TOTAL wrong", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
