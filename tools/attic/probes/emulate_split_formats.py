"""Numerical emulation (CPU, no GPU needed) of cheaper operand splits for the dense contractions, against an fp64 dot product:

  bf16x3   : x = xh + xl (bf16 each), acc += xh*wh + xh*wl + xl*wh                                  3 bf16 MFMA products  (today)
  f16+mx8  : x = x1 + x2 with x1 = fp16(x); main term x1*w1 exact in fp32; the two cross terms x1*w2 + x2*w1 with both factors
             rounded to MX-FP8 (e4m3 elements, one power-of-two scale per 32-element block along K)  1 f16 + 2 fp8 products (fp8 runs
             at twice the bf16 rate through v_mfma_scale_f32_16x16x128_f8f6f4: 2/3 of today's MFMA cycles)
  f16x3    : the same split with all three products in fp16                                            3 f16 products (reference point)

Reports the error of a K-long dot product relative to sqrt(sum (x*w)^2) (the scale random rounding errors accumulate at).
    python tools/probes/emulate_split_formats.py"""
import numpy as np
import torch

torch.manual_seed(0)
M, K, N = 512, 1024, 256


def bf16(v):
    return v.to(torch.bfloat16).to(torch.float32)


def f16(v):
    return v.to(torch.float16).to(torch.float32)


def mx8(v):
    """MX-FP8: blocks of 32 along the last axis share a power-of-two scale that maps the block maximum just under e4m3's 448"""
    shp = v.shape
    b = v.reshape(*shp[:-1], shp[-1] // 32, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)          # max / scale in [256, 512) -> saturates above 448: clamp
    q = (b / scale).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) * scale
    return q.reshape(shp)


def case(name, x, w):
    exact = x.double() @ w.double().T
    norm = torch.sqrt((x.double() ** 2) @ (w.double() ** 2).T).clamp_min(1e-300)
    out = {}
    xh, wh = bf16(x), bf16(w)
    xl, wl = bf16(x - xh), bf16(w - wh)
    out["bf16x3"] = (xh.double() @ wh.double().T + xh.double() @ wl.double().T + xl.double() @ wh.double().T)
    x1, w1 = f16(x), f16(w)
    x2, w2 = x - x1, w - w1
    out["f16+mx8"] = x1.double() @ w1.double().T + mx8(x1).double() @ mx8(w2).double().T + mx8(x2).double() @ mx8(w1).double().T
    out["f16x3"] = x1.double() @ w1.double().T + x1.double() @ f16(w2).double().T + f16(x2).double() @ w1.double().T
    out["bf16"] = xh.double() @ wh.double().T
    print(name)
    for k, v in out.items():
        e = ((v - exact) / norm)
        print("   %-8s rms %.3e   max %.3e" % (k, float(e.pow(2).mean().sqrt()), float(e.abs().max())))


w = torch.randn(N, K) * 0.03
case("gaussian activations", torch.randn(M, K), w)
x = torch.relu(torch.randn(M, K)) * torch.exp(torch.randn(M, K))           # post-ReLU, heavy-tailed magnitudes
case("post-ReLU, log-normal magnitudes", x, w)
x = torch.relu(torch.randn(M, K)) * torch.exp(2.0 * torch.randn(M, 1))     # per-row scale spread over ~4 decades
case("post-ReLU, rows of very different scale", x, w * torch.exp(1.5 * torch.randn(N, 1)))
