"""Thin host layer over the C ABI (include/ape_hip.h): weight repacking into the device layout and one Python
function per kernel.  Tensors are torch CUDA(HIP) tensors used purely as device buffers; every function enqueues on
torch's current stream and returns without synchronising.  No function here has a CPU path.

Device layout (DESIGN.md "Data layout in HBM"): activations NHWC fp32 `[B,H,W,C]` (points are `[B,N,1,C]`),
weights `[Cout][KH][KW][Cin4]` with Cin zero-padded to a multiple of 4.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_PRELU, ACT_RELU, ACT_SIGMOID, ConvParams  # noqa: F401


def _st():
    return _lib.stream_ptr()


class LaunchProfile:
    """Optional per-launch HIP-event timing of the heavy kernels on torch's current stream (bench.py's roofline leg).
    Events are recorded on the very stream the kernel is enqueued on; durations are read after the final sync.
    A record = (kernel label as rocprofv3 spells it, shape string, algorithmic flop, algorithmic bytes, start, end): the flop are
    2 * M * Cout * KH * KW * Cin of the convolution the launch evaluates, the bytes its activations in + weights + residual +
    output, each counted once (what a perfect kernel would move)."""

    def __init__(self, only=None):
        self.records = []
        self.only = only        # None: time every profiled launch; else a set of labels (the others run un-timed, without
                                # the two event packets per launch that cost ~4 us of dispatch gap each)

    def wants(self, label):
        return self.only is None or label in self.only

    def begin(self, label):
        """-> a start event, or None when this label is not being timed"""
        if not self.wants(label):
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def end(self, e0, label, shape, flop, nbytes):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((label, shape, float(flop), float(nbytes), e0, e1))

    def summary(self, by_shape=False):
        out = {}
        for name, shape, flop, nbytes, e0, e1 in self.records:
            key = (name, shape) if by_shape else name
            d = out.setdefault(key, {"launches": 0, "flop": 0.0, "bytes": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flop"] += flop
            d["bytes"] += nbytes
            d["ms"] += e0.elapsed_time(e1)
        return out


def _prof_begin(label):
    prof = PROFILE
    return None if prof is None else prof.begin(label)


def _prof_end(e0, label, shape, flop, nbytes):
    if e0 is not None and PROFILE is not None:
        PROFILE.end(e0, label, shape, flop, nbytes)


def _conv_cost(conv, b, h_in, w_in, ho, wo, residual):
    """(algorithmic flop, algorithmic bytes) of one conv launch: see LaunchProfile"""
    flop = 2.0 * b * ho * wo * conv.cout * conv.kh * conv.kw * conv.cin_real
    nbytes = 4.0 * (b * h_in * w_in * conv.cin_real + conv.cout * conv.kh * conv.kw * conv.cin_real + b * ho * wo * conv.cout * (2 if residual else 1))
    return flop, nbytes


PROFILE = None   # set to a LaunchProfile() to time conv launches

# operand precision of the dense contractions (accumulation is always fp32):
#   "f32"    exact fp32 MFMA (v_mfma_f32_32x32x2_f32)                 157 TFLOP/s peak
#   "bf16x3" split-bf16, 3 products per term, ~2^-16 operand error      833 TFLOP/s effective peak
#   "bf16"   plain bf16 operands (2^-8)                                 2.5 PFLOP/s peak
PRECISIONS = ("f32", "bf16x3", "bf16")
SPLITK_SMALL_M = False     # split-K for the pose networks' layers whose grid would not fill the chip: FramePipeline(low_latency=True) sets it around its pose stage (round 6)
USE_HALO_KERNEL = True    # route eligible 3x3 convs of the bf16 paths to the LDS-halo kernel (conv3x3_halo.hip)
USE_GEMM_KERNEL = os.environ.get("APE_USE_GEMM_KERNEL", "1") != "0"    # route Cin % 32 == 0 layers the halo kernel does not take to conv_gemm.hip (else conv_bf16.hip)
USE_CONV_MULTI = os.environ.get("APE_USE_CONV_MULTI", "1") != "0"     # the PSP stage convolutions in one launch (conv1x1_multi); 0: one launch each (A/B)
GEMM_VARIANT = int(os.environ.get("APE_GEMM_VARIANT", "0"))          # 0 = chosen from the shape; 1..4 force a block shape (tools/microbench_generic.py)


FMT_F32, FMT_S32 = 0, 1
USE_S32 = os.environ.get("APE_USE_S32", "1") != "0"    # pre-split activations between the segmentor's bf16x3 layers (conv_gemm_s32.hip)


class S32:
    """A pre-split activation tensor (include/ape_hip.h "S32"): per pixel and 32-channel group 128 bytes [hi 32 x bf16 | lo 32 x bf16].
    Same bytes per channel as fp32, so it lives in a float32 torch tensor `t` of the logical shape [B,H,W,C]; only kernels that
    declare S32 operands may read it (the element values of `t` are NOT the activations)."""
    __slots__ = ("t",)

    def __init__(self, t):
        if t.dtype != torch.float32 or t.shape[-1] % 32:
            raise ValueError("S32 needs a float32 buffer with a multiple of 32 channels")
        self.t = t

    @property
    def shape(self):
        return self.t.shape

    @property
    def device(self):
        return self.t.device

    def __getitem__(self, idx):          # batch slicing only
        return S32(self.t[idx])

    def to_f32(self):
        y = torch.empty_like(self.t)
        c = self.t.shape[-1]
        _lib.check(_lib.lib().ape_convert_s32(_lib.dptr(self.t, torch.float32), _lib.dptr(y), self.t.numel() // c, c, 0, _st()), "ape_convert_s32")
        return y

    @staticmethod
    def from_f32(x):
        y = torch.empty_like(x)
        c = x.shape[-1]
        _lib.check(_lib.lib().ape_convert_s32(_lib.dptr(x, torch.float32), _lib.dptr(y), x.numel() // c, c, 1, _st()), "ape_convert_s32")
        return S32(y)


def pack_conv_weight(w, device):
    """[Cout,Cin,KH,KW] | [Cout,Cin,1] | [Cout,Cin]  ->  contiguous f32 [Cout,KH,KW,Cin4] on `device`."""
    w = w.detach()
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif w.dim() == 3:
        w = w[:, :, :, None]
    cout, cin, kh, kw = w.shape
    cin4 = (cin + 3) // 4 * 4
    out = torch.zeros(cout, kh, kw, cin4, dtype=torch.float32, device=device)
    out[..., :cin] = w.to(device=device, dtype=torch.float32).permute(0, 2, 3, 1)
    return out.contiguous()


def allow_splitk(obj, _seen=None):
    """mark every Conv reachable from `obj` (a plan object, a list / tuple / dict of them, an UpConv) as free to take the split-K form at small M"""
    _seen = set() if _seen is None else _seen
    if id(obj) in _seen or obj is None or isinstance(obj, (int, float, str, bytes, torch.Tensor)):
        return
    _seen.add(id(obj))
    if isinstance(obj, Conv):
        obj.allow_splitk = True
        return
    if isinstance(obj, (list, tuple)):
        for o in obj:
            allow_splitk(o, _seen)
    elif isinstance(obj, dict):
        for o in obj.values():
            allow_splitk(o, _seen)
    elif hasattr(obj, "__dict__") and type(obj).__module__.startswith("autoposeestimation_amd") or type(obj).__name__ == "Plan":
        for o in vars(obj).values():
            allow_splitk(o, _seen)


class Conv:
    """One conv / 1x1 / Linear layer bound to ape_conv2d_nhwc_f32."""

    def __init__(self, weight, bias=None, stride=1, pad=0, dil=1, act=ACT_NONE, alpha=0.0, device="cuda", precision="f32"):
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (PRECISIONS,))
        self.w = pack_conv_weight(weight, device)
        self.cout, self.kh, self.kw, self.cin = self.w.shape
        self.cin_real = weight.shape[1]
        self.precision = precision
        self.nsplit = {"f32": 0, "bf16x3": 3, "bf16": 1}[precision]
        if self.nsplit:
            k = self.kh * self.kw * self.cin
            n = _lib.lib().ape_packed_weights_bf16_elems(self.cout, k)
            self.wp = torch.empty(n, dtype=torch.bfloat16, device=device)
            _lib.check(_lib.lib().ape_pack_weights_bf16(_lib.dptr(self.w), _lib.dptr(self.wp), self.cout, k, _st()),
                       "ape_pack_weights_bf16")
            self.variant = "conv_bf16_kernel<%d,%s>" % (self.nsplit, "128,128,2,2,64,false" if self.cout > 64 else "128,64,4,1,32,true")
        else:
            self.variant = "conv_f32_kernel<%s>" % ("128,2,2" if self.cout > 64 else "64,4,1" if self.cout > 32 else "32,4,1")
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        self.stride, self.pad, self.dil, self.act, self.alpha = stride, pad, dil, act, float(alpha)

    @classmethod
    def from_packed(cls, w, wp, cin_real, stride=1, pad=0, dil=1, act=ACT_NONE, alpha=0.0, precision="f32"):
        """a layer over operands that are ALREADY packed (w: f32 [Cout,KH,KW,Cin4]; wp: the split-bf16 planes of the same weights, needed
        unless precision == 'f32'): the training tape keeps them per parameter and re-packs them after each optimizer step
        (autograd.WeightBank), instead of packing per call"""
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (PRECISIONS,))
        self = cls.__new__(cls)
        self.w = w
        self.cout, self.kh, self.kw, self.cin = w.shape
        self.cin_real = cin_real
        self.precision = precision
        self.nsplit = {"f32": 0, "bf16x3": 3, "bf16": 1}[precision]
        if self.nsplit:
            if wp is None:
                raise ValueError("precision %s needs the packed bf16 planes" % precision)
            self.wp = wp
            self.variant = "conv_bf16_kernel<%d,%s>" % (self.nsplit, "128,128,2,2,64,false" if self.cout > 64 else "128,64,4,1,32,true")
        else:
            self.variant = "conv_f32_kernel<%s>" % ("128,2,2" if self.cout > 64 else "64,4,1" if self.cout > 32 else "32,4,1")
        self.bias = None
        self.stride, self.pad, self.dil, self.act, self.alpha = stride, pad, dil, act, float(alpha)
        return self

    def s32k(self):
        """the weights in the S32K grouping ([Cout][K/32][hi 32 | lo 32] bf16) for the S32 consumers; built on first use"""
        ws = getattr(self, "_ws32", None)
        if ws is None:
            k = self.kh * self.kw * self.cin
            ws = torch.empty(self.cout * k * 2, dtype=torch.bfloat16, device=self.w.device)
            _lib.check(_lib.lib().ape_pack_weights_s32k(_lib.dptr(self.w), _lib.dptr(ws), self.cout, k, _st()), "ape_pack_weights_s32k")
            self._ws32 = ws
        return ws

    def mx6k(self):
        """the weights as F16M6 lines ([Cout][9 * Cin / 32][128 B], mx6.pack_conv_weights) for ape_conv3x3_halo_mx; built on first use"""
        wq = getattr(self, "_wmx6", None)
        if wq is None:
            from . import mx6
            w = self.w.detach().float().cpu().numpy().reshape(self.cout, self.kh, self.kw, self.cin)       # (the unpacked layout: [Cout][kh][kw][Cin])
            lines = mx6.pack_lines(w.reshape(self.cout, self.kh * self.kw * (self.cin // 32), 32))
            wq = self._wmx6 = torch.from_numpy(lines).to(self.w.device)
        return wq

    def _generic_variant(self, m):
        """name of the template instantiation ape_conv2d_nhwc_* dispatches to (mirrors the C++ rule; profiling label only)"""
        if self.nsplit and self.cout >= 256 and ((-(-self.cout // 256)) * 256 - self.cout) * 8 <= self.cout and m >= 65536:
            return "conv_bf16_kernel<%d,256,256,4,2,32,true>" % self.nsplit
        return self.variant

    def _gemm_variant(self, m):
        """name of the conv_gemm.hip instantiation ape_conv_gemm_bf16 dispatches to (mirrors the C++ rule; profiling label only)"""
        pure = "true" if (self.kh == 1 and self.kw == 1 and self.stride == 1 and self.pad == 0) else "false"
        v = GEMM_VARIANT & 15
        if v == 0:
            k = self.kh * self.kw * self.cin
            if self.cout <= 64:
                v = 3
            elif (self.cout >= 256 and ((-(-self.cout // 256)) * 256 - self.cout) * 8 <= self.cout and k >= 256
                  and -(-m // 256) * -(-self.cout // 256) >= 192):
                v = 1
            elif (self.cout >= 192 and ((-(-self.cout // 192)) * 192 - self.cout) * 8 <= self.cout and k >= 256
                  and -(-m // 256) * -(-self.cout // 192) >= 192):
                v = 4
            else:
                v = 2
        # the same spelling as rocprofv3's kernel name (template arguments NSPLIT, BM, BN, WM, WN, PURE, PP), so that bench.py finds
        # this kernel's HBM traffic in profiles/*_pmc_traffic.json
        return "conv_gemm_kernel<%d,%s,%s,%s>" % (self.nsplit, {1: "256,256,2,4", 2: "128,128,2,2", 3: "256,64,4,1", 4: "256,192,2,4",
                                                               5: "256,256,2,4", 6: "128,192,2,2"}[v], pure, "true" if v == 5 else "false")

    def out_hw(self, h, w):
        ho = (h + 2 * self.pad - self.dil * (self.kh - 1) - 1) // self.stride + 1
        wo = (w + 2 * self.pad - self.dil * (self.kw - 1) - 1) // self.stride + 1
        return ho, wo

    def __call__(self, x, out=None, xoff=0, yoff=0, residual=None, roff=0, bias=None, bias_bstride=0, act=None, upsample2x=False,
                 out_fmt=FMT_F32, splitk=False):
        """x[B,H,W,ldx] (reads channels xoff..xoff+cin) -> out[B,Ho,Wo,ldy] (writes yoff..yoff+cout).
        upsample2x: the conv runs on the bilinear x2 (align_corners=True) up-sampling of x, fused into the LDS-halo kernel
        when it applies, otherwise materialised by ape_bilinear_nhwc_f32 first.
        x / residual may be `S32` (pre-split) tensors and out_fmt=FMT_S32 returns one: only on the split-bf16 kernels that
        declare S32 operands (this never converts silently -- an unsupported combination raises)."""
        if isinstance(x, S32) or isinstance(residual, S32):
            return self._call_s32(x, out, xoff, yoff, residual, roff, bias, bias_bstride, act, out_fmt)
        if out_fmt == FMT_S32 and isinstance(out, S32):
            out = out.t
        if upsample2x:
            can_fuse = (self.nsplit and USE_HALO_KERNEL and self.kh == 3 and self.kw == 3 and self.stride == 1 and self.pad == 1
                        and self.dil == 1 and self.cin % 32 == 0 and x.shape[3] == self.cin and xoff == 0)
            if not can_fuse:
                return self(bilinear(x, 2 * x.shape[1], 2 * x.shape[2], True), out, 0, yoff, residual, roff, bias, bias_bstride, act)
        b, h, w, ldx = x.shape
        if upsample2x:
            h, w = 2 * h, 2 * w
        ho, wo = self.out_hw(h, w)
        if out is None:
            out = torch.empty(b, ho, wo, self.cout, dtype=torch.float32, device=x.device)
        if tuple(out.shape[:3]) != (b, ho, wo):
            raise ValueError("conv output buffer %s does not match %s" % (tuple(out.shape), (b, ho, wo)))
        if residual is not None and tuple(residual.shape[:3]) != (b, ho, wo):
            raise ValueError("residual shape mismatch")
        bias = self.bias if bias is None else bias
        # small-M layers of the POSE networks (one crop's 20 x 20 maps: 4..16 output tiles walking K = 4608 alone on a 256-CU chip) take the split-K
        # form whenever the library finds it worth it (ape_conv_gemm_splitk_workspace_bytes > 0: fewer than 96 tiles and >= 8 k-tiles) -- the
        # batch-1 live loop of main.py:517-553; batches that fill the chip (the bench's 64 crops: 200+ tiles per layer) never do.  Opt-in per layer
        # (`allow_splitk`, set by PoseNet / PoseRefineNet for their plans): the split changes the fp32 summation order, and the SEGMENTOR's class
        # maps must stay bit-identical between a frame run alone and the same frame inside a batch (tests/test_gpu_bench_parity.py)
        splitk = splitk or (SPLITK_SMALL_M and self.__dict__.get("allow_splitk", False))
        # the parameter block and the kernel choice depend on shapes only: kept per call signature (the training tape calls every layer
        # with the same shapes step after step; building the ctypes struct and asking the library twice cost ~15 us per call)
        key = (b, h, w, ldx, xoff, out.shape[3], yoff, self.act if act is None else act, bias_bstride, 0 if residual is None else residual.shape[3], roff,
               bool(upsample2x), splitk, USE_HALO_KERNEL, USE_GEMM_KERNEL)
        plans = self.__dict__.get("_plans")
        if plans is None:
            plans = self.__dict__["_plans"] = {}
        plan = plans.get(key)
        if plan is None:
            p = ConvParams(B=b, H=h, W=w, Cin=self.cin, ldx=ldx, xoff=xoff, Ho=ho, Wo=wo, Cout=self.cout,
                           ldy=out.shape[3], yoff=yoff, KH=self.kh, KW=self.kw, stride=self.stride, pad=self.pad,
                           dil=self.dil, act=self.act if act is None else act, alpha=self.alpha,
                           bias_bstride=bias_bstride, ldr=0 if residual is None else residual.shape[3], roff=roff,
                           ups=int(bool(upsample2x)))
            # the halo kernel tiles the image in 16x16 pixels: use it only when those tiles are mostly full (crop feature maps of
            # 20x20 / 40x40 would waste 30..60 % of the MFMAs; the flattened-M generic kernel has no such edge effect)
            halo = bool(self.nsplit and USE_HALO_KERNEL and (upsample2x or (h * w) >= 0.8 * (-(-h // 16) * -(-w // 16) * 256))
                        and _lib.lib().ape_conv3x3_halo_supported(ctypes.byref(p)))
            gemm = bool(not halo and self.nsplit and USE_GEMM_KERNEL and _lib.lib().ape_conv_gemm_supported(ctypes.byref(p)))
            sk_bytes = int(_lib.lib().ape_conv_gemm_splitk_workspace_bytes(ctypes.byref(p))) if (gemm and splitk) else 0
            if len(plans) > 64:
                plans.clear()
            plan = plans[key] = (p, halo, gemm, sk_bytes)
        p, halo, gemm, sk_bytes = plan
        if p.alpha != self.alpha:
            p.alpha = self.alpha
        if out_fmt == FMT_S32 and not gemm:
            raise ValueError("an S32 output from an fp32 input exists only on the conv_gemm kernel (Cin % 32 == 0, not a halo layer)")
        e0 = None
        if PROFILE is not None:
            label = ("conv3x3_halo_kernel<%d,%d,%d,%s,false>" % (self.nsplit, self.dil, 64 if self.cout <= 64 else 128, "true" if upsample2x else "false")
                     if halo else self._gemm_variant(b * ho * wo) if gemm else self._generic_variant(b * ho * wo))
            e0 = _prof_begin(label)
        if halo:
            rc = _lib.lib().ape_conv3x3_halo_bf16(_lib.dptr(x, torch.float32), _lib.dptr(self.wp), _lib.dptr(bias),
                                                  _lib.dptr(residual), _lib.dptr(out, torch.float32), ctypes.byref(p),
                                                  self.nsplit, _st())
            _lib.check(rc, "ape_conv3x3_halo_bf16")
        elif gemm and sk_bytes and out_fmt == FMT_F32:
            # the training tape's batch-1 layers (autograd.ConvFn): k-tiles dealt to several workgroups per output tile
            ws = torch.empty(sk_bytes, dtype=torch.uint8, device=x.device)
            rc = _lib.lib().ape_conv_gemm_bf16_splitk(_lib.dptr(x, torch.float32), _lib.dptr(self.wp), _lib.dptr(bias), _lib.dptr(residual),
                                                      _lib.dptr(out, torch.float32), ctypes.byref(p), self.nsplit, _lib.dptr(ws), ws.numel(), _st())
            _lib.check(rc, "ape_conv_gemm_bf16_splitk")
        elif gemm:
            rc = _lib.lib().ape_conv_gemm_bf16_fmt(_lib.dptr(x, torch.float32), _lib.dptr(self.wp), _lib.dptr(bias),
                                                   _lib.dptr(residual), _lib.dptr(out, torch.float32), out_fmt, ctypes.byref(p),
                                                   self.nsplit, GEMM_VARIANT, _st())
            _lib.check(rc, "ape_conv_gemm_bf16_fmt")
        elif self.nsplit:
            rc = _lib.lib().ape_conv2d_nhwc_bf16(_lib.dptr(x, torch.float32), _lib.dptr(self.wp), _lib.dptr(bias),
                                                 _lib.dptr(residual), _lib.dptr(out, torch.float32), ctypes.byref(p),
                                                 self.nsplit, _st())
            _lib.check(rc, "ape_conv2d_nhwc_bf16")
        else:
            rc = _lib.lib().ape_conv2d_nhwc_f32(_lib.dptr(x, torch.float32), _lib.dptr(self.w), _lib.dptr(bias),
                                                _lib.dptr(residual), _lib.dptr(out, torch.float32), ctypes.byref(p), _st())
            _lib.check(rc, "ape_conv2d_nhwc_f32")
        if e0 is not None:
            hin, win = (h // 2, w // 2) if upsample2x else (h, w)
            _prof_end(e0, label, "%dx%dx%d %d->%d k%d s%d d%d%s" % (b, ho, wo, self.cin_real, self.cout, self.kh, self.stride, self.dil, " ups" if upsample2x else ""),
                      *_conv_cost(self, b, hin, win, ho, wo, residual is not None))
        return S32(out) if out_fmt == FMT_S32 else out


# 3x3 layers on S32 inputs with these input channel counts take the half-pass kernel (fp16 main product + block-scaled e2m3 cross terms,
# conv3x3_halo_mx.hip) instead of halo_s32's three bf16 products.  Off unless asked for: not bit-identical to bf16x3.
USE_MX6 = os.environ.get("APE_USE_MX6", "0") != "0"
MX6_CIN = tuple(int(v) for v in os.environ.get("APE_MX6_CIN", "512").split(","))


def _conv_call_s32(self, x, out, xoff, yoff, residual, roff, bias, bias_bstride, act, out_fmt):
    if self.nsplit != 3:
        raise ValueError("S32 operands exist only in the split-bf16 ('bf16x3') precision")
    if not isinstance(x, S32):
        raise ValueError("S32 output / residual needs an S32 input on this path")
    xt = x.t
    b, h, w, ldx = xt.shape
    ho, wo = self.out_hw(h, w)
    if out is None:
        out = torch.empty(b, ho, wo, self.cout, dtype=torch.float32, device=xt.device)
    out_t = out.t if isinstance(out, S32) else out
    if tuple(out_t.shape[:3]) != (b, ho, wo):
        raise ValueError("conv output buffer %s does not match %s" % (tuple(out_t.shape), (b, ho, wo)))
    res_t, res_fmt = None, FMT_F32
    if residual is not None:
        res_t, res_fmt = (residual.t, FMT_S32) if isinstance(residual, S32) else (residual, FMT_F32)
        if tuple(res_t.shape[:3]) != (b, ho, wo):
            raise ValueError("residual shape mismatch")
    bias = self.bias if bias is None else bias
    p = ConvParams(B=b, H=h, W=w, Cin=self.cin, ldx=ldx, xoff=xoff, Ho=ho, Wo=wo, Cout=self.cout, ldy=out_t.shape[3], yoff=yoff,
                   KH=self.kh, KW=self.kw, stride=self.stride, pad=self.pad, dil=self.dil, act=self.act if act is None else act,
                   alpha=self.alpha, bias_bstride=bias_bstride, ldr=0 if res_t is None else res_t.shape[3], roff=roff, ups=0)
    is3 = self.kh == 3
    if is3 and USE_MX6 and self.cin in MX6_CIN and self.cout >= 128 and self.stride == 1 and xoff == 0 and ldx == self.cin \
            and _lib.lib().ape_conv3x3_halo_mx_supported(ctypes.byref(p)):
        # half the matrix passes (conv3x3_halo_mx.hip): the S32 input is re-expressed once as F16M6 lines
        xq = torch.empty_like(xt)
        _lib.check(_lib.lib().ape_s32_to_f16m6(_lib.dptr(xt, torch.float32), _lib.dptr(xq, torch.float32), b * h * w, ldx, _st()), "ape_s32_to_f16m6")
        label = "halo_mx_kernel<%d>" % self.dil
        e0 = _prof_begin(label)
        rc = _lib.lib().ape_conv3x3_halo_mx(_lib.dptr(xq, torch.float32), _lib.dptr(self.mx6k()), _lib.dptr(bias), _lib.dptr(res_t), res_fmt,
                                            _lib.dptr(out_t, torch.float32), out_fmt, ctypes.byref(p), _st())
        _lib.check(rc, "ape_conv3x3_halo_mx")
        if e0 is not None:
            _prof_end(e0, label, "%dx%dx%d %d->%d k%d s%d d%d" % (b, ho, wo, self.cin_real, self.cout, self.kh, self.stride, self.dil),
                      *_conv_cost(self, b, h, w, ho, wo, residual is not None))
        return S32(out_t) if out_fmt == FMT_S32 else out_t
    if is3:
        if not _lib.lib().ape_conv3x3_halo_s32_supported(ctypes.byref(p)):
            raise ValueError("no S32 kernel for this layer geometry (3x3, stride %d, dil %d, Cin %d, Cout %d)" % (self.stride, self.dil, self.cin, self.cout))
        label = "halo_s32_kernel<%d, true>" % self.dil          # (the kernel's name in a rocprofv3 trace: <dilation, ping-pong schedule>)
    else:
        if not _lib.lib().ape_conv_gemm_s32_supported(ctypes.byref(p)):
            raise ValueError("no S32 kernel for this layer geometry (%dx%d, stride %d, Cin %d, Cout %d)" % (self.kh, self.kw, self.stride, self.cin, self.cout))
        label = ("gemm_s32_res_kernel<%d, false>" if res_t is not None else "gemm_s32_kernel<%d, false>") % (128 if self.cout <= 128 else 192 if (-(-self.cout // 192) * 192 - self.cout) < (-(-self.cout // 256) * 256 - self.cout) else 256)
    e0 = _prof_begin(label)
    fn = _lib.lib().ape_conv3x3_halo_s32 if is3 else _lib.lib().ape_conv_gemm_s32
    rc = fn(_lib.dptr(xt, torch.float32), _lib.dptr(self.s32k()), _lib.dptr(bias), _lib.dptr(res_t), res_fmt,
            _lib.dptr(out_t, torch.float32), out_fmt, ctypes.byref(p), _st())
    _lib.check(rc, "ape_conv3x3_halo_s32" if is3 else "ape_conv_gemm_s32")
    if e0 is not None:
        _prof_end(e0, label, "%dx%dx%d %d->%d k%d s%d d%d" % (b, ho, wo, self.cin_real, self.cout, self.kh, self.stride, self.dil),
                  *_conv_cost(self, b, h, w, ho, wo, residual is not None))
    return S32(out_t) if out_fmt == FMT_S32 else out_t


Conv._call_s32 = _conv_call_s32


def conv_seg_head(conv, x, head_w, head_b, double_softmax=True, upsample2x=False):
    """`seg_head(conv(x))` for a 64-channel 3x3 layer: fused into the LDS-halo kernel's epilogue when that kernel applies
    (split-bf16 / bf16 precision, Cin % 32 == 0), otherwise the two calls.  -> label[B,H,W] u8, score[B,H,W] f32"""
    b, h, w, ldx = x.shape
    if upsample2x:
        h, w = 2 * h, 2 * w
    c = head_w.shape[0]
    fusable = (conv.nsplit and USE_HALO_KERNEL and conv.cout == 64 and conv.kh == 3 and conv.kw == 3 and conv.stride == 1
               and conv.pad == 1 and conv.dil == 1 and conv.cin % 32 == 0 and ldx == conv.cin and c <= 16
               and (upsample2x or (h * w) >= 0.8 * (-(-h // 16) * -(-w // 16) * 256)))
    if not fusable:
        return seg_head(conv(x, upsample2x=upsample2x), head_w, head_b, double_softmax)
    p = ConvParams(B=b, H=h, W=w, Cin=conv.cin, ldx=ldx, xoff=0, Ho=h, Wo=w, Cout=conv.cout, ldy=conv.cout, yoff=0, KH=3, KW=3,
                   stride=1, pad=1, dil=1, act=conv.act, alpha=conv.alpha, bias_bstride=0, ldr=0, roff=0, ups=int(bool(upsample2x)))
    label = torch.empty(b, h, w, dtype=torch.uint8, device=x.device)
    score = torch.empty(b, h, w, dtype=torch.float32, device=x.device)
    hlabel = "conv3x3_halo_kernel<%d,1,64,%s,true>" % (conv.nsplit, "true" if upsample2x else "false")
    e0 = _prof_begin(hlabel)
    rc = _lib.lib().ape_conv3x3_halo_seghead_bf16(_lib.dptr(x, torch.float32), _lib.dptr(conv.wp), _lib.dptr(conv.bias), ctypes.byref(p),
                                                  conv.nsplit, _lib.dptr(head_w, torch.float32), _lib.dptr(head_b), c, _lib.dptr(label),
                                                  _lib.dptr(score), int(bool(double_softmax)), _st())
    _lib.check(rc, "ape_conv3x3_halo_seghead_bf16")
    if e0 is not None:
        hin, win = (h // 2, w // 2) if upsample2x else (h, w)
        # the 64-channel activation is never written: out = label (1 B) + score (4 B) per pixel
        _prof_end(e0, hlabel, "%dx%dx%d %d->%d k3 s1 d1%s +head" % (b, h, w, conv.cin_real, conv.cout, " ups" if upsample2x else ""),
                  2.0 * b * h * w * conv.cout * 9 * conv.cin_real,
                  4.0 * (b * hin * win * conv.cin_real + conv.cout * 9 * conv.cin_real) + 5.0 * b * h * w)
    return label, score


USE_FUSED_STEM = os.environ.get("APE_USE_FUSED_STEM", "1") != "0"
USE_U8_STEM = os.environ.get("APE_USE_U8_STEM", "1") != "0"      # the stem reads the uint8 frames itself (ToTensor / Normalize fused into its patch load)


class U8Frames:
    """The network input `ToTensor + Normalize` of uint8 frames or crops (pipeline/utils.py:421-427, 556-560), NOT yet materialised:
    crop o = the hc x wc window of frame rects[o,0] at (rects[o,1], rects[o,2]) of rgb[F,H,W,3] u8.  `stem_pool` consumes it directly (the
    fp32 NHWC4 image -- 16 bytes per pixel -- is then never written); `materialise()` gives the tensor `preprocess_u8` would."""
    __slots__ = ("rgb", "rects", "hc", "wc", "div255", "_x4")

    def __init__(self, rgb, rects, hc, wc, div255):
        self.rgb, self.rects, self.hc, self.wc, self.div255, self._x4 = rgb, rects, int(hc), int(wc), bool(div255), None

    @property
    def shape(self):
        return (self.rects.shape[0], self.hc, self.wc, 4)

    @property
    def is_cuda(self):
        return self.rgb.is_cuda

    @property
    def device(self):
        return self.rgb.device

    def __getitem__(self, idx):          # batch slicing only
        return U8Frames(self.rgb, self.rects[idx].contiguous(), self.hc, self.wc, self.div255)

    def materialise(self):
        if self._x4 is None:
            self._x4 = preprocess_u8(self.rgb, self.rects, self.hc, self.wc, self.div255)
        return self._x4


def stem_pool(conv, x):
    """maxpool3x3s2(conv(x)) for the ResNet stem (7x7 / stride 2 / pad 3, 4 -> 64 channels, ReLU): one fused kernel on the bf16
    paths (the half-resolution activation is never stored), the two calls otherwise.  x: the normalised image [B,H,W,4], or `U8Frames`."""
    if isinstance(x, U8Frames):
        fus = (USE_U8_STEM and USE_FUSED_STEM and conv.nsplit and conv.cout == 64 and conv.cin == 4 and conv.kh == 7 and conv.kw == 7
               and conv.stride == 2 and conv.pad == 3 and conv.dil == 1 and conv.act == ACT_RELU)
        if not fus:
            return stem_pool(conv, x.materialise())
        n, h, w, _ = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty(n, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1, 64, dtype=torch.float32, device=x.device)
        rc = _lib.lib().ape_stem_conv_pool_u8(_lib.dptr(x.rgb, torch.uint8), x.rgb.shape[0], _lib.dptr(x.rects, torch.int32), _lib.dptr(conv.w, torch.float32),
                                              _lib.dptr(conv.bias), _lib.dptr(y), n, x.rgb.shape[1], x.rgb.shape[2], h, w, int(x.div255), conv.nsplit, _st())
        _lib.check(rc, "ape_stem_conv_pool_u8")
        return y
    b, h, w, ldx = x.shape
    fusable = (USE_FUSED_STEM and conv.nsplit and conv.cout == 64 and conv.cin == 4 and ldx == 4 and conv.kh == 7 and conv.kw == 7
               and conv.stride == 2 and conv.pad == 3 and conv.dil == 1 and conv.act == ACT_RELU)
    if not fusable:
        return maxpool3x3s2(conv(x))
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty(b, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1, 64, dtype=torch.float32, device=x.device)
    rc = _lib.lib().ape_stem_conv_pool_bf16(_lib.dptr(x, torch.float32), _lib.dptr(conv.w, torch.float32), _lib.dptr(conv.bias),
                                            _lib.dptr(y), b, h, w, conv.nsplit, _st())
    _lib.check(rc, "ape_stem_conv_pool_bf16")
    return y


def maxpool3x3s2(x):
    b, h, w, c = x.shape
    y = torch.empty(b, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ape_maxpool3x3s2_nhwc_f32(_lib.dptr(x, torch.float32), _lib.dptr(y), b, h, w, c, _st()),
               "ape_maxpool3x3s2_nhwc_f32")
    return y


def adaptive_avgpool(x, s):
    b, h, w, c = x.shape
    y = torch.empty(b, s, s, c, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ape_adaptive_avgpool_nhwc_f32(_lib.dptr(x, torch.float32), _lib.dptr(y), b, h, w, c, s, _st()),
               "ape_adaptive_avgpool_nhwc_f32")
    return y


def adaptive_avgpool_multi(x, sizes, channels=None):
    """{s: AdaptiveAvgPool2d((s, s))(x[..., :channels])} for several sizes in one pass over x[B,H,W,C]; per-size launches when the bin edges
    of the sizes cut an axis into more than 12 atoms.  `channels` (default: all) pools the leading channels only and gives [B,s,s,channels]."""
    fmt = FMT_S32 if isinstance(x, S32) else FMT_F32
    xt = x.t if fmt == FMT_S32 else x
    b, h, w, ld = xt.shape
    c = ld if channels is None else int(channels)
    sizes = list(sizes)
    ys = [torch.empty(b, s, s, c, dtype=torch.float32, device=xt.device) for s in sizes]
    ws = _workspace(_lib.lib().ape_adaptive_avgpool_multi_workspace_bytes(b, c), xt.device)
    ptrs = (ctypes.c_void_p * len(sizes))(*[y.data_ptr() for y in ys])
    szs = (ctypes.c_int * len(sizes))(*sizes)
    rc = _lib.lib().ape_adaptive_avgpool_multi_nhwc_ld(_lib.dptr(xt, torch.float32), fmt, ptrs, szs, len(sizes), b, h, w, c, ld, _lib.dptr(ws),
                                                       ws.numel() * ws.element_size(), _st())
    if rc == -1:      # APE_EINVAL: too many atoms for this geometry
        xf = x.to_f32() if fmt == FMT_S32 else x
        xf = xf if c == ld else xf[..., :c].contiguous()
        return {s: adaptive_avgpool(xf, s) for s in sizes}
    _lib.check(rc, "ape_adaptive_avgpool_multi_nhwc_ld")
    return dict(zip(sizes, ys))


def conv1x1_multi(convs, xs):
    """[conv(x) for conv, x in zip(convs, xs)] for up to four independent 1x1 / stride-1 convolutions of fp32 maps in ONE launch
    (ape_conv_gemm_bf16_multi): the PSP module's stage convolutions (pspnet.py:15-18, 22), problems of 1 .. 18 tiles each whose separate
    launches ran one after the other with the chip idle.  Bit for bit what the separate calls give; they are what runs where the
    launch does not apply (more than four problems, other precisions, geometries the GEMM kernel does not take)."""
    convs, xs = list(convs), list(xs)
    n = len(convs)
    ok = (1 <= n <= 4 and USE_CONV_MULTI and USE_GEMM_KERNEL and not (GEMM_VARIANT & 15) and
          all(c.nsplit and c.nsplit == convs[0].nsplit and c.kh == 1 and c.kw == 1 and c.stride == 1 and c.pad == 0 and not isinstance(x, S32)
              and x.shape[3] == c.cin and x.is_contiguous() for c, x in zip(convs, xs)))
    if ok:
        outs = [torch.empty(x.shape[0], x.shape[1], x.shape[2], c.cout, dtype=torch.float32, device=x.device) for c, x in zip(convs, xs)]
        ps = (ConvParams * n)(*[ConvParams(B=x.shape[0], H=x.shape[1], W=x.shape[2], Cin=c.cin, ldx=x.shape[3], xoff=0, Ho=x.shape[1], Wo=x.shape[2],
                                           Cout=c.cout, ldy=c.cout, yoff=0, KH=1, KW=1, stride=1, pad=0, dil=1, act=c.act, alpha=c.alpha, bias_bstride=0,
                                           ldr=0, roff=0, ups=0) for c, x in zip(convs, xs)])
        ok = all(_lib.lib().ape_conv_gemm_supported(ctypes.byref(ps[i])) for i in range(n))
    if not ok:
        return [c(x) for c, x in zip(convs, xs)]
    arr = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731
    rc = _lib.lib().ape_conv_gemm_bf16_multi(n, arr(xs), arr([c.wp for c in convs]), arr([c.bias for c in convs]), arr(outs), ps, convs[0].nsplit, _st())
    _lib.check(rc, "ape_conv_gemm_bf16_multi")
    return outs


def bilinear(x, ho, wo, align_corners, out=None, yoff=0, accumulate=False):
    b, h, w, c = x.shape
    if out is None:
        out = torch.empty(b, ho, wo, c, dtype=torch.float32, device=x.device)
    if tuple(out.shape[:3]) != (b, ho, wo):
        raise ValueError("bilinear output buffer mismatch")
    rc = _lib.lib().ape_bilinear_nhwc_f32(_lib.dptr(x, torch.float32), _lib.dptr(out, torch.float32), b, h, w, c, c, ho, wo,
                                          out.shape[3], yoff, int(bool(align_corners)), int(bool(accumulate)), _st())
    _lib.check(rc, "ape_bilinear_nhwc_f32")
    return out


def psp_prior_sum(zs, h, w):
    """zs = [z1[B,1,1,C], z2[B,2,2,C], z3[B,3,3,C], z6[B,6,6,C]] -> sum of their bilinear (align_corners=False) resizes [B,h,w,C]"""
    b, c = zs[0].shape[0], zs[0].shape[3]
    out = torch.empty(b, h, w, c, dtype=torch.float32, device=zs[0].device)
    rc = _lib.lib().ape_psp_prior_sum_f32(*[_lib.dptr(z, torch.float32) for z in zs], _lib.dptr(out), b, h, w, c, _st())
    _lib.check(rc, "ape_psp_prior_sum_f32")
    return out


USE_PSP_FOLD = os.environ.get("APE_USE_PSP_FOLD", "1") != "0"    # the PSP prior sum as 64 more K-columns of the bottleneck's contraction (S32 graph)
PSP_FOLD_K = 64


def psp_bottleneck_folded(conv, x576, zs, out_fmt=FMT_S32):
    """relu(W_f . f + sum_s upsample(z_s) + bias) (pspnet.py:22-24 with the prior sum of :12-17 folded in) as ONE contraction over
    K = Cin + 64: `x576` is the S32 map f[B,h,w,Cin + 64] whose last 64 channels this call fills with the pixels' bilinear coefficients,
    `zs` = [z1[B,1,1,C], z2[B,2,2,C], z3[B,3,3,C], z6[B,6,6,C]] fp32 (the merged prior convs of the pooled maps), `conv` the bottleneck's
    feats columns (Cin -> C, bias, ReLU).  No [B,h,w,C] prior-sum tensor is written or read (include/ape_hip.h ape_psp_fold_operands)."""
    xt = x576.t
    b, h, w, ld = xt.shape
    if conv.nsplit != 3 or conv.kh != 1 or ld != conv.cin + PSP_FOLD_K or [tuple(z.shape) for z in zs] != [(b, s, s, conv.cout) for s in (1, 2, 3, 6)]:
        raise ValueError("psp_bottleneck_folded: a split-bf16 1x1 layer, an S32 map of Cin + 64 channels and the four prior maps [B,s,s,Cout]")
    groups = conv.cin // 32 + 2
    wimg = torch.empty(b, conv.cout, groups * 64, dtype=torch.bfloat16, device=xt.device)
    _lib.check(_lib.lib().ape_psp_fold_operands(_lib.dptr(conv.s32k()), *[_lib.dptr(z, torch.float32) for z in zs], _lib.dptr(wimg),
                                                _lib.dptr(xt, torch.float32), b, h, w, ld, conv.cin, conv.cout, _st()), "ape_psp_fold_operands")
    out = torch.empty(b, h, w, conv.cout, dtype=torch.float32, device=xt.device)
    p = ConvParams(B=b, H=h, W=w, Cin=ld, ldx=ld, xoff=0, Ho=h, Wo=w, Cout=conv.cout, ldy=conv.cout, yoff=0, KH=1, KW=1, stride=1, pad=0, dil=1,
                   act=conv.act, alpha=conv.alpha, bias_bstride=0, ldr=0, roff=0, ups=0)
    label = "gemm_s32_kernel<%d, false>" % (128 if conv.cout <= 128 else 192 if (-(-conv.cout // 192) * 192 - conv.cout) < (-(-conv.cout // 256) * 256 - conv.cout) else 256)
    e0 = _prof_begin(label)
    rc = _lib.lib().ape_conv_gemm_s32_per_image(_lib.dptr(xt, torch.float32), _lib.dptr(wimg), conv.cout * groups * 128, _lib.dptr(conv.bias),
                                                _lib.dptr(out), out_fmt, ctypes.byref(p), _st())
    _lib.check(rc, "ape_conv_gemm_s32_per_image")
    if e0 is not None:      # algorithmic: the layer's own contraction (K = Cin) + the prior sum's 50 coefficient columns; bytes: f in, out, per-frame weights
        _prof_end(e0, label, "%dx%dx%d %d+%d->%d k1 s1 d1 psp-fold" % (b, h, w, conv.cin, PSP_FOLD_K, conv.cout), 2.0 * b * h * w * conv.cout * ld,
                  4.0 * (b * h * w * ld + b * conv.cout * ld + b * h * w * conv.cout))
    return S32(out) if out_fmt == FMT_S32 else out


USE_UPFUSE = os.environ.get("APE_USE_UPFUSE", "1") != "0"    # 64-channel PSPUpsample layers on S32 inputs as ONE kernel (upconv_fused.hip)


class UpConv:
    """PSPUpsample (pspnet.py:27-37) as  1x1 conv at low resolution (9*Cout channels) + ape_upconv3x3_gather_f32 -- or, for a 64 -> 64
    layer on an S32 input (up_3), as the ONE kernel of upconv_fused.hip that keeps the 9*Cout-channel tensor on the chip (bit-identical
    to the two calls).  fma: both interpolation steps as chained fused multiply-adds instead of separately rounded products."""

    def __init__(self, weight, bias, alpha, device="cuda", precision="f32", fma=False):
        cout, cin, kh, kw = weight.shape
        assert (kh, kw) == (3, 3)
        w9 = weight.detach().permute(2, 3, 0, 1).reshape(9 * cout, cin)            # row = (ky*3+kx)*Cout + co
        self.mix = Conv(w9, None, device=device, precision=precision)
        self.bias = bias.detach().to(device=device, dtype=torch.float32).contiguous()
        self.alpha, self.cout, self.cin, self.fma = float(alpha), cout, cin, bool(fma)

    def fusable(self, x):
        """the one-kernel form applies: split-bf16 operands, an S32 input, and a geometry upconv_fused.hip serves"""
        return bool(USE_UPFUSE and isinstance(x, S32) and self.mix.nsplit == 3 and x.shape[3] == self.cin
                    and _lib.lib().ape_upconv3x3_fused_supported(x.shape[1], x.shape[2], self.cin, self.cout))

    def _gather(self, z, out_fmt):
        b, h, w, _ = z.shape
        out = torch.empty(b, 2 * h, 2 * w, self.cout, dtype=torch.float32, device=z.device)
        strip = self.cout % 64 == 0 and _lib.lib().ape_upconv3x3_gather_strip_rows(-1) > 0      # the library's own routing rule
        glabel = "upconv_gather_%skernel<%s,%s>" % ("strip_" if strip else "", "true" if out_fmt == FMT_S32 else "false", "true" if self.fma else "false")
        e0 = _prof_begin(glabel)
        rc = _lib.lib().ape_upconv3x3_gather_ex(_lib.dptr(z, torch.float32), _lib.dptr(self.bias), _lib.dptr(out), out_fmt, b, h, w,
                                                self.cout, ACT_PRELU, self.alpha, int(self.fma), _st())
        _lib.check(rc, "ape_upconv3x3_gather_ex")
        if e0 is not None:      # HBM-bound: z read once (9 * Cout channels at low resolution) + the output written once
            _prof_end(e0, glabel, "%dx%dx%d C%d" % (b, 2 * h, 2 * w, self.cout), 0.0, 4.0 * (b * h * w * 9 * self.cout + b * 4 * h * w * self.cout))
        return S32(out) if out_fmt == FMT_S32 else out

    def __call__(self, x, out_fmt=FMT_F32, fused=None):
        """x fp32 or S32 -> [B,2h,2w,Cout] in `out_fmt` (two-call form: the 9*Cout-channel intermediate z stays fp32, the gather is
        VALU work).  fused: None = the one-kernel form whenever it applies, False = never, True = required."""
        if fused is None:
            fused = self.fusable(x)
        if not fused:
            return self._gather(self.mix(x), out_fmt)
        if not self.fusable(x):
            raise ValueError("no fused PSPUpsample kernel for this layer / input (needs split-bf16 operands, an S32 input, 64 -> 64 channels)")
        b, h, w, _ = x.shape
        out = torch.empty(b, 2 * h, 2 * w, self.cout, dtype=torch.float32, device=x.device)
        label = "upconv_fused_kernel<%d,false,%s>" % (self.cin // 32, "true" if self.fma else "false")
        e0 = _prof_begin(label)
        rc = _lib.lib().ape_upconv3x3_fused_s32(_lib.dptr(x.t, torch.float32), _lib.dptr(self.mix.s32k()), _lib.dptr(self.bias), _lib.dptr(out),
                                                out_fmt, b, h, w, self.cin, ACT_PRELU, self.alpha, int(self.fma), _st())
        _lib.check(rc, "ape_upconv3x3_fused_s32")
        if e0 is not None:
            _prof_end(e0, label, "%dx%dx%d %d->%d up (executed flop; direct form x4)" % (b, 2 * h, 2 * w, self.cin, self.cout), *self._fused_cost(b, h, w, 4.0 * self.cout))
        return S32(out) if out_fmt == FMT_S32 else out

    def _fused_cost(self, b, h, w, out_bytes_per_pixel):
        """(flop, algorithmic bytes) of the one-kernel form.  The flop are the ones the kernel EXECUTES on the matrix cores -- the channel
        mixing at LOW resolution, 2 * B * h * w * (9 * Cout) * Cin -- a quarter of the reference's convolution at high resolution
        (pspnet.py:30-33: 2 * B * 4 h w * Cout * 9 * Cin): pricing the layer's direct form against the matrix peak would overstate the
        matrix-core throughput four times over (the kernel is bound by vector-instruction issue, not by the matrix pipe).  Bytes: the input
        read + the output written once."""
        return (2.0 * b * h * w * 9 * self.cout * self.cin,
                4.0 * (b * h * w * self.cin + 9 * self.cout * self.cin) + out_bytes_per_pixel * b * 4 * h * w)

    def seg_head(self, x, head_w, head_b, double_softmax=True, fused=None):
        """`seg_head(self(x))`: label[B,2h,2w] u8, score f32 -- in the one-kernel form the 64-channel activation is never stored"""
        if fused is None:
            fused = self.fusable(x) and head_w.shape[0] <= 16 and self.cout == 64
        if not fused:
            return seg_head(self(x, fused=False), head_w, head_b, double_softmax)
        b, h, w, _ = x.shape
        c = head_w.shape[0]
        label = torch.empty(b, 2 * h, 2 * w, dtype=torch.uint8, device=x.device)
        score = torch.empty(b, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
        klabel = "upconv_fused_kernel<%d,true,%s>" % (self.cin // 32, "true" if self.fma else "false")
        e0 = _prof_begin(klabel)
        rc = _lib.lib().ape_upconv3x3_fused_seghead_s32(_lib.dptr(x.t, torch.float32), _lib.dptr(self.mix.s32k()), _lib.dptr(self.bias), b, h, w,
                                                        self.cin, ACT_PRELU, self.alpha, int(self.fma), _lib.dptr(head_w, torch.float32),
                                                        _lib.dptr(head_b), c, _lib.dptr(label), _lib.dptr(score), int(bool(double_softmax)), _st())
        _lib.check(rc, "ape_upconv3x3_fused_seghead_s32")
        if e0 is not None:
            _prof_end(e0, klabel, "%dx%dx%d %d->%d up +head (executed flop; direct form x4)" % (b, 2 * h, 2 * w, self.cin, self.cout), *self._fused_cost(b, h, w, 5.0))
        return label, score


def gather_rows(x, index):
    """x[B,R,C], index[B,n] i64 -> [B,n,C]"""
    b, r, c = x.shape
    n = index.shape[1]
    y = torch.empty(b, n, c, dtype=torch.float32, device=x.device)
    rc = _lib.lib().ape_gather_rows_f32(_lib.dptr(x, torch.float32), _lib.dptr(index, torch.int64), _lib.dptr(y), b, r, n, c, _st())
    _lib.check(rc, "ape_gather_rows_f32")
    return y


def ups_patch_gather(x, index):
    """x[B,h,w,C], index[B,n] i64 (pixel of the x2 up-sampled image) -> [B*n, 1, 1, 9*C] patches of the up-sampled image"""
    b, h, w, c = x.shape
    n = index.shape[1]
    out = torch.empty(b * n, 1, 1, 9 * c, dtype=torch.float32, device=x.device)
    rc = _lib.lib().ape_ups_patch_gather_f32(_lib.dptr(x, torch.float32), _lib.dptr(index, torch.int64), _lib.dptr(out), b, h, w, c, n, _st())
    _lib.check(rc, "ape_ups_patch_gather_f32")
    return out


def conv3x3_as_matrix(conv):
    """A 3x3 Conv viewed as the 1x1 contraction over its 9*Cin patch columns: shares the (already [Cout][KH][KW][Cin]) weights."""
    assert conv.kh == 3 and conv.kw == 3 and conv.cin == conv.cin_real
    m = object.__new__(Conv)
    m.__dict__.update(conv.__dict__)
    m.kh = m.kw = 1
    m.cin = m.cin_real = 9 * conv.cin
    m.stride, m.pad, m.dil = 1, 0, 1
    if conv.nsplit == 0:
        m.w = conv.w.view(conv.cout, 1, 1, 9 * conv.cin)
    return m


def log_softmax_rows(x):
    c = x.shape[-1]
    rows = x.numel() // c
    y = torch.empty_like(x)
    _lib.check(_lib.lib().ape_log_softmax_rows_f32(_lib.dptr(x, torch.float32), _lib.dptr(y), rows, c, _st()),
               "ape_log_softmax_rows_f32")
    return y


def mean_rows(x):
    """x[B,n,C] -> [B,C]"""
    b, n, c = x.shape
    y = torch.empty(b, c, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ape_mean_rows_f32(_lib.dptr(x, torch.float32), _lib.dptr(y), b, n, c, _st()), "ape_mean_rows_f32")
    return y


def pad3to4(x):
    """x[...,3] -> [...,4]"""
    rows = x.numel() // 3
    y = torch.empty(*x.shape[:-1], 4, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ape_pad3to4_f32(_lib.dptr(x, torch.float32), _lib.dptr(y), rows, _st()), "ape_pad3to4_f32")
    return y


def head_select(h, off_r, off_t, off_c, wr, br, wt, bt, wc, bc, obj, b, n, k):
    """h[B*n, ldh] -> out[B,n,8]"""
    out = torch.empty(b, n, 8, dtype=torch.float32, device=h.device)
    rc = _lib.lib().ape_head_select_f32(_lib.dptr(h, torch.float32), h.shape[-1], off_r, off_t, off_c, _lib.dptr(wr),
                                        _lib.dptr(br), _lib.dptr(wt), _lib.dptr(bt), _lib.dptr(wc), _lib.dptr(bc),
                                        _lib.dptr(obj, torch.int64), _lib.dptr(out), b, n, k, _st())
    _lib.check(rc, "ape_head_select_f32")
    return out


def pose_select(heads, points4, want_new_points=True):
    """heads[B,n,8], points4[B,n,4] -> pose[B,7] f64, which[B] i32, new_points4[B,n,4] | None"""
    b, n, _ = heads.shape
    pose = torch.empty(b, 7, dtype=torch.float64, device=heads.device)
    which = torch.empty(b, dtype=torch.int32, device=heads.device)
    newp = torch.empty(b, n, 4, dtype=torch.float32, device=heads.device) if want_new_points else None
    rc = _lib.lib().ape_pose_select_f32(_lib.dptr(heads, torch.float32), _lib.dptr(points4, torch.float32), _lib.dptr(pose),
                                        _lib.dptr(which), _lib.dptr(newp), b, n, _st())
    _lib.check(rc, "ape_pose_select_f32")
    return pose, which, newp


def pose_compose(pose, ref_r, ref_t):
    """in-place: pose[B,7] f64 <- pose o (ref_r[B,>=4], ref_t[B,>=3]) ; ref_* are (possibly strided) row views"""
    b = pose.shape[0]
    rc = _lib.lib().ape_pose_compose_f64(_lib.dptr(pose, torch.float64), ctypes.c_void_p(ref_r.data_ptr()), ref_r.stride(0),
                                         ctypes.c_void_p(ref_t.data_ptr()), ref_t.stride(0), b, _st())
    _lib.check(rc, "ape_pose_compose_f64")
    return pose


def pose_recentre(points4, pose):
    b, n, _ = points4.shape
    out = torch.empty_like(points4)
    rc = _lib.lib().ape_pose_recentre_f32(_lib.dptr(points4, torch.float32), _lib.dptr(pose, torch.float64), _lib.dptr(out), b, n, _st())
    _lib.check(rc, "ape_pose_recentre_f32")
    return out


# ---- segmentation post-processing / selection --------------------------------------------------------------------
def preprocess_u8(rgb, rects, hc, wc, div255):
    """rgb[B,H,W,3] u8, rects[n,3] i32 (frame,row0,col0) -> [n,hc,wc,4] f32"""
    b, h, w, _ = rgb.shape
    n = rects.shape[0]
    out = torch.empty(n, hc, wc, 4, dtype=torch.float32, device=rgb.device)
    rc = _lib.lib().ape_preprocess_u8_nhwc4(_lib.dptr(rgb, torch.uint8), _lib.dptr(rects, torch.int32), _lib.dptr(out), n, h, w,
                                            hc, wc, int(bool(div255)), _st())
    _lib.check(rc, "ape_preprocess_u8_nhwc4")
    return out


def seg_argmax(logits, n_classes, double_softmax=True):
    """logits[B,H,W,ld] -> label[B,H,W] u8, score[B,H,W] f32"""
    b, h, w, ld = logits.shape
    label = torch.empty(b, h, w, dtype=torch.uint8, device=logits.device)
    score = torch.empty(b, h, w, dtype=torch.float32, device=logits.device)
    rc = _lib.lib().ape_seg_argmax_f32(_lib.dptr(logits, torch.float32), ld, n_classes, _lib.dptr(label), _lib.dptr(score),
                                       b * h * w, int(bool(double_softmax)), _st())
    _lib.check(rc, "ape_seg_argmax_f32")
    return label, score


def seg_head(feat, w, bias, double_softmax=True):
    """feat[B,H,W,64], w[C,64], bias[C] -> label[B,H,W] u8, score[B,H,W] f32 (fused final conv + softmax^2 + argmax)"""
    b, h, wd, c = feat.shape
    if c != 64:
        raise ValueError("seg_head expects the 64-channel up_3 activation")
    label = torch.empty(b, h, wd, dtype=torch.uint8, device=feat.device)
    score = torch.empty(b, h, wd, dtype=torch.float32, device=feat.device)
    rc = _lib.lib().ape_seg_head_f32(_lib.dptr(feat, torch.float32), _lib.dptr(w, torch.float32), _lib.dptr(bias), w.shape[0],
                                     _lib.dptr(label), _lib.dptr(score), b * h * wd, int(bool(double_softmax)), _st())
    _lib.check(rc, "ape_seg_head_f32")
    return label, score


_ws_cache = {}
CAPTURING = False       # set while a HIP graph is being captured (pipeline.FramePipeline pose graphs): workspaces then come from the graph's own pool


def _workspace(nbytes, device):
    if CAPTURING:
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    key = (str(device), "seg", torch.cuda.current_stream().cuda_stream)   # one workspace per stream: sub-batches may overlap
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


SEG_SCORE_MEAN, SEG_SCORE_SUM = 0, 1


def seg_components(label, score, n_classes, min_pixels=100, score_mode=SEG_SCORE_MEAN):
    """-> objmap[B,H,W] u8, det[B,C,5] i32 (valid,rmin,rmax,cmin,cmax); score_mode: best component by mean (full_prediction)
    or summed (do_cca) probability"""
    b, h, w = label.shape
    objmap = torch.empty_like(label)
    det = torch.empty(b, n_classes, 5, dtype=torch.int32, device=label.device)
    nbytes = _lib.lib().ape_seg_components_workspace_bytes(b, h, w, n_classes)
    ws = _workspace(nbytes, label.device)
    rc = _lib.lib().ape_seg_components_scored(_lib.dptr(label, torch.uint8), _lib.dptr(score, torch.float32), _lib.dptr(objmap),
                                              _lib.dptr(det), b, h, w, n_classes, min_pixels, score_mode, _lib.dptr(ws),
                                              ws.numel(), _st())
    _lib.check(rc, "ape_seg_components_scored")
    return objmap, det


def bgsub_features(f_rgb, b_rgb, f_depth, b_depth, gate, mean, std, want_diff=False):
    """f_rgb/b_rgb[B,H,W,3] u8, f_depth/b_depth[B,H,W] u16, gate[B,2] f64 (min,max) -> x8[B,H,W,8] f32 (7 normalised difference
    channels + a zero), optionally the uint8 channels diff[B,H,W,7]."""
    import ctypes
    b, h, w, _ = f_rgb.shape
    out = torch.empty(b, h, w, 8, dtype=torch.float32, device=f_rgb.device)
    diff = torch.empty(b, h, w, 7, dtype=torch.uint8, device=f_rgb.device) if want_diff else None
    m = (ctypes.c_float * 7)(*[float(v) for v in mean])
    s = (ctypes.c_float * 7)(*[float(v) for v in std])
    rc = _lib.lib().ape_bgsub_features_f32(_lib.dptr(f_rgb, torch.uint8), _lib.dptr(b_rgb, torch.uint8),
                                           _lib.dptr(f_depth, torch.uint16), _lib.dptr(b_depth, torch.uint16),
                                           _lib.dptr(gate, torch.float64), ctypes.cast(m, ctypes.c_void_p),
                                           ctypes.cast(s, ctypes.c_void_p), _lib.dptr(out),
                                           _lib.dptr(diff) if want_diff else None, b, h, w, _st())
    _lib.check(rc, "ape_bgsub_features_f32")
    return (out, diff) if want_diff else out


def choose_points(objmap, depth, objects, n_points, seed=0):
    """objects[n,6] i32 (frame,cls,rmin,rmax,cmin,cmax) -> choose[n,N] i64, n_cand[n] i32.  `seed`: an int, or a device tensor (int32[1]) that the
    kernel reads when it runs -- the form a captured launch needs to follow a seed that changes between replays"""
    b, h, w = objmap.shape
    n = objects.shape[0]
    choose = torch.zeros(n, n_points, dtype=torch.int64, device=objmap.device)
    n_cand = torch.zeros(n, dtype=torch.int32, device=objmap.device)
    stride = h * w
    cand = torch.empty(n, stride, dtype=torch.int32, device=objmap.device)
    if isinstance(seed, torch.Tensor):
        rc = _lib.lib().ape_choose_points_dseed(_lib.dptr(objmap, torch.uint8), _lib.dptr(depth, torch.uint16), _lib.dptr(objects, torch.int32),
                                                n, h, w, n_points, _lib.dptr(seed, torch.int32), _lib.dptr(cand), stride, _lib.dptr(choose),
                                                _lib.dptr(n_cand), _st())
    else:
        rc = _lib.lib().ape_choose_points(_lib.dptr(objmap, torch.uint8), _lib.dptr(depth, torch.uint16), _lib.dptr(objects, torch.int32),
                                          n, h, w, n_points, seed & 0xFFFFFFFF, _lib.dptr(cand), stride, _lib.dptr(choose),
                                          _lib.dptr(n_cand), _st())
    _lib.check(rc, "ape_choose_points")
    return choose, n_cand


def backproject(depth, objects, choose, intr, depth_scale):
    b, h, w = depth.shape
    n, npts = choose.shape
    pts = torch.empty(n, npts, 4, dtype=torch.float32, device=depth.device)
    rc = _lib.lib().ape_backproject_f32(_lib.dptr(depth, torch.uint16), _lib.dptr(objects, torch.int32), _lib.dptr(choose, torch.int64),
                                        _lib.dptr(pts), n, h, w, npts, float(intr["fx"]), float(intr["fy"]), float(intr["ppx"]),
                                        float(intr["ppy"]), float(depth_scale), _st())
    _lib.check(rc, "ape_backproject_f32")
    return pts


# ---- ADD / ADD-S ------------------------------------------------------------------------------------------------------
def adds_dis(pred_r, pred_t, points, model, target, symmetric, want_pred=False, want_std=True):
    """pred_r[N,4], pred_t[N,3], points[N,3]|None, model[M,3], target[M,3] -> dis[N], std[N]|None, pred[N,M,3]|None"""
    n, m = pred_r.shape[0], model.shape[0]
    dev = pred_r.device
    dis = torch.empty(n, dtype=torch.float32, device=dev)
    std = torch.empty(n, dtype=torch.float32, device=dev) if want_std else None
    pred = torch.empty(n, m, 3, dtype=torch.float32, device=dev) if want_pred else None
    rc = _lib.lib().ape_adds_dis_f32(_lib.dptr(pred_r, torch.float32), _lib.dptr(pred_t, torch.float32), _lib.dptr(points),
                                     _lib.dptr(model, torch.float32), _lib.dptr(target, torch.float32), n, m,
                                     int(bool(symmetric)), _lib.dptr(pred), _lib.dptr(dis), _lib.dptr(std), _st())
    _lib.check(rc, "ape_adds_dis_f32")
    return dis, std, pred


def adds_dis_batched(pred_r, pred_t, model, target, symmetric):
    """pred_r[B,4], pred_t[B,3], model[B,M,3], target[B,M,3] -> dis[B]: ADD / ADD-S of B objects, each against its own clouds, in one launch"""
    b, m = model.shape[0], model.shape[1]
    dis = torch.empty(b, dtype=torch.float32, device=model.device)
    ws = torch.empty(b, m, dtype=torch.float32, device=model.device)          # per-point distances (the mean is a second launch)
    rc = _lib.lib().ape_adds_dis_batched_f32(_lib.dptr(pred_r, torch.float32), _lib.dptr(pred_t, torch.float32), _lib.dptr(model, torch.float32),
                                             _lib.dptr(target, torch.float32), b, m, int(bool(symmetric)), _lib.dptr(ws), _lib.dptr(dis), _st())
    _lib.check(rc, "ape_adds_dis_batched_f32")
    return dis


def adds_select(dis, std, pred_c, pred_r, pred_t, points, w):
    """-> out9 (loss, dis[which], q[4], t[3]) f32 on device, which i32[1]"""
    out = torch.empty(9, dtype=torch.float32, device=dis.device)
    which = torch.empty(1, dtype=torch.int32, device=dis.device)
    rc = _lib.lib().ape_adds_select_f32(_lib.dptr(dis), _lib.dptr(std), _lib.dptr(pred_c, torch.float32), _lib.dptr(pred_r),
                                        _lib.dptr(pred_t), _lib.dptr(points), dis.shape[0], float(w), _lib.dptr(out),
                                        _lib.dptr(which), _st())
    _lib.check(rc, "ape_adds_select_f32")
    return out, which


def recentre_qt(pts, qt7):
    """pts[n,3], qt7 (device f32 [7]: unnormalised quaternion + translation) -> (pts - t) . R(q)"""
    out = torch.empty_like(pts)
    rc = _lib.lib().ape_recentre_qt_f32(_lib.dptr(pts, torch.float32), _lib.dptr(qt7, torch.float32), _lib.dptr(out),
                                        pts.shape[0], _st())
    _lib.check(rc, "ape_recentre_qt_f32")
    return out
