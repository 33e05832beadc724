"""Tape for the training step (DenseFusion/tools/train.py:205-238).

The reference trains through torch autograd over cuDNN.  Here torch.autograd is only the TAPE (which op ran, what it saved,
in which order to call the backward rules); every forward and backward rule below is a hand-written gfx950 kernel behind the
C ABI (include/ape_hip.h, csrc/backward.hip, csrc/adds.hip).  Tensors are NHWC fp32 as on the inference path; convolution
weights stay in the reference's parameter layout ([Cout,Cin,KH,KW] / [Cout,Cin,1] / [Cout,Cin]) so state dicts and optimizers
see the reference's tensors.  No CPU fallback: host tensors raise in _lib.dptr.
"""
import ctypes
import math

import torch

from autoposeestimation_amd import _lib
from autoposeestimation_amd import engine as E

_st = _lib.stream_ptr


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def _c(t):
    return t.contiguous()


def act_bwd(dy, ref, act, alpha=0.0):
    dx = torch.empty_like(dy)
    rc = _lib.lib().ape_act_bwd_f32(_lib.dptr(dy, torch.float32), _lib.dptr(ref), _lib.dptr(dx), dy.numel(), act, float(alpha), _st())
    _lib.check(rc, "ape_act_bwd_f32")
    return dx


def colsum(x2d_rows, c, ld, off, like):
    out = torch.empty(c, dtype=torch.float32, device=like.device)
    scratch = torch.empty(64 * c, dtype=torch.float32, device=like.device)
    rc = _lib.lib().ape_colsum_f32(_lib.dptr(like, torch.float32), _lib.dptr(out), x2d_rows, c, ld, off, _lib.dptr(scratch), _st())
    _lib.check(rc, "ape_colsum_f32")
    return out


def conv_wgrad(x, dy, cout, kh, kw, stride, pad, dil, param_shape=None):
    """x[B,H,W,Cx] (Cx % 4 == 0), dy[B,Ho,Wo,Cout] -> dw[Cout,KH,KW,Cx], or -- with `param_shape` ([Cout,Cin,...], the reference's
    parameter) -- directly the parameter's gradient in ITS layout (the reduction pass writes it, no permute copy)"""
    b, h, w, cx = x.shape
    _, ho, wo, ldy = dy.shape
    p = _lib.ConvParams(B=b, H=h, W=w, Cin=cx, ldx=cx, xoff=0, Ho=ho, Wo=wo, Cout=cout, ldy=ldy, yoff=0, KH=kh, KW=kw,
                        stride=stride, pad=pad, dil=dil, act=0, alpha=0.0, bias_bstride=0, ldr=0, roff=0, ups=0)
    nbytes = _lib.lib().ape_conv2d_wgrad_workspace_bytes(ctypes.byref(p))
    ws = _ws(nbytes, x.device)
    if param_shape is not None:
        dw = torch.empty(param_shape, dtype=torch.float32, device=x.device)
        rc = _lib.lib().ape_conv2d_wgrad_param_f32(_lib.dptr(x, torch.float32), _lib.dptr(dy, torch.float32), _lib.dptr(dw), ctypes.byref(p),
                                                   int(param_shape[1]), _lib.dptr(ws), ws.numel(), _st())
        _lib.check(rc, "ape_conv2d_wgrad_param_f32")
        return dw
    dw = torch.empty(cout, kh, kw, cx, dtype=torch.float32, device=x.device)
    rc = _lib.lib().ape_conv2d_wgrad_nhwc_f32(_lib.dptr(x, torch.float32), _lib.dptr(dy, torch.float32), _lib.dptr(dw), ctypes.byref(p),
                                              _lib.dptr(ws), ws.numel(), _st())
    _lib.check(rc, "ape_conv2d_wgrad_nhwc_f32")
    return dw


def _rows_dense(w):
    """every row w[i] is laid out like a contiguous tensor (rows may be further apart than their length)"""
    expect = 1
    for size, stride in zip(reversed(w.shape[1:]), reversed(w.stride()[1:])):
        if size != 1 and stride != expect:
            return False
        expect *= size
    return w.dim() >= 2 and (w.shape[0] == 1 or w.stride(0) >= expect)


class WeightBank:
    """The conv kernels' operands of ONE parameter, kept while the parameter does not change ("keep packed weights
    until optimizer.step()"): the forward form ([Cout,KH,KW,Cin4] f32 + split-bf16 planes) and the input-gradient form (flipped taps,
    channels transposed).  Both are written by ONE launch of ape_pack_train_weights from the parameter's own storage; `Adam.step()`
    re-packs the banks of all its parameters in one further launch right after the update.  A bank is rebuilt on the next use when the
    parameter's torch version counter moved (a torch optimizer, any in-place op on the parameter itself).  Writes the counter does not see
    -- `p.data.mul_()`, `dist.broadcast(p.data)`, a raw-pointer kernel -- are covered one level up: the modules' training forward re-packs
    every bank of the module first (`_HipModule.sync_banks`, one launch), and load_state_dict / .to() / train() drop the banks
    (`invalidate_banks`).  Code that calls the tape's conv functions directly after such a write calls `refresh_banks` / `invalidate_banks`
    itself."""

    def __init__(self, weight, precision):
        w = weight.detach()
        if not w.is_cuda:
            raise ValueError("the training path needs its parameters in device memory (no CPU fallback)")
        shape = tuple(w.shape) + (1,) * (4 - w.dim())
        self.cout, self.cin, self.kh, self.kw = shape
        taps = self.kh * self.kw
        # packed straight from the parameter's storage: the whole tensor, or a block of rows / columns of it whose rows stay dense
        own = w.dtype == torch.float32 and _rows_dense(w)
        if not own:                    # anything else (the zero-padded head rows built per call): packed from a dense copy, per call
            w = w.float().contiguous()
        self.src = w
        self.version = weight._version if own else None
        self.precision = precision
        nsplit = {"f32": 0, "bf16x3": 3, "bf16": 1}[precision]
        dev = w.device
        self.ops = []
        jobs = (_lib.PackJob * 2)()
        self.max_elems = 0
        for tr, (n, c) in enumerate(((self.cout, self.cin), (self.cin, self.cout))):
            c4 = (c + 3) // 4 * 4
            kp = (taps * c4 + 7) // 8 * 8
            wf = torch.empty(n, self.kh, self.kw, c4, dtype=torch.float32, device=dev)
            wp = torch.empty(2 * n * kp, dtype=torch.bfloat16, device=dev) if nsplit else None
            self.ops.append((wf, wp, c))
            jobs[tr] = _lib.PackJob(src=w.data_ptr(), dst_f32=wf.data_ptr(), dst_bf16=wp.data_ptr() if nsplit else None, cout=self.cout,
                                    cin=self.cin, taps=taps, transpose=tr, src_ld=w.stride(0) if self.cout > 1 else self.cin * taps, reserved=0)
            self.max_elems = max(self.max_elems, n * kp)
        self.jobs = jobs
        self.convs = {}
        self.table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self.refresh()

    def refresh(self):
        _lib.check(_lib.lib().ape_pack_train_weights(2, _lib.dptr(self.table), self.max_elems, _st()), "ape_pack_train_weights")

    def conv(self, transposed, stride, pad, dil, act):
        key = (transposed, stride, pad, dil, act)
        c = self.convs.get(key)
        if c is None:
            wf, wp, c_real = self.ops[1 if transposed else 0]
            c = self.convs[key] = E.Conv.from_packed(wf, wp, c_real, stride, pad, dil, act, precision=self.precision)
        return c


def weight_bank(weight, precision):
    """the (cached) bank of a parameter, or of a dense-row block of one (kept on the parameter it is a view of)"""
    base = weight._base if weight._is_view() else None
    if base is not None:
        key = (weight.storage_offset(), tuple(weight.shape), tuple(weight.stride()))
        slices = getattr(base, "_ape_slices", None)
        bank = slices.get(key) if slices is not None else None
    else:
        bank = getattr(weight, "_ape_bank", None)
    if (bank is None or bank.version is None or bank.version != weight._version or bank.precision != precision
            or bank.src.data_ptr() != weight.data_ptr()):
        bank = WeightBank(weight, precision)
        try:
            if base is None:
                weight._ape_bank = bank
            elif bank.version is not None:
                if slices is None:
                    slices = base._ape_slices = {}
                slices[key] = bank
        except AttributeError:
            pass
    return bank


def invalidate_banks(params):
    """forget the kept operands of `params` (and of their row / column blocks): the next use packs them again from the parameter"""
    for p in params:
        for name in ("_ape_bank", "_ape_slices"):
            if getattr(p, name, None) is not None:
                try:
                    delattr(p, name)
                except AttributeError:
                    pass
    refresh_banks.cache.clear()


def refresh_banks(params):
    """re-pack the operands of every parameter in `params` that has banks (its own and those of its row / column blocks): one launch (their
    job tables concatenated once and kept)"""
    banks = []
    for p in params:
        b = getattr(p, "_ape_bank", None)
        if b is not None and b.version is not None:
            banks.append(b)
        banks.extend(getattr(p, "_ape_slices", {}).values())
    if not banks:
        return
    key = tuple(id(b) for b in banks)
    cache = refresh_banks.cache
    hit = cache.get(key)
    if hit is None:
        if len(cache) >= 8:
            cache.clear()
        hit = cache[key] = (torch.cat([b.table for b in banks]), max(b.max_elems for b in banks), banks)
    _lib.check(_lib.lib().ape_pack_train_weights(2 * len(banks), _lib.dptr(hit[0]), hit[1], _st()), "ape_pack_train_weights")


refresh_banks.cache = {}


def _w4(weight):
    """parameter layout -> [Cout,Cin,KH,KW]"""
    if weight.dim() == 2:
        return weight[:, :, None, None]
    if weight.dim() == 3:
        return weight[:, :, :, None]
    return weight


class ConvFn(torch.autograd.Function):
    """y = act(conv(x, W) + bias + residual): nn.Conv2d / nn.Conv1d (1 x 1 over points) / nn.Linear of the reference networks.
    x[B,H,W,Cx] with Cx = Cin rounded up to a multiple of 4 (zero channels), y[B,Ho,Wo,Cout]."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, pad, dil, act, precision):
        bank = weight_bank(weight, precision)
        conv = bank.conv(False, stride, pad, dil, act)
        if x.shape[3] != conv.cin:
            raise ValueError("input has %d channels, the packed weight expects %d" % (x.shape[3], conv.cin))
        bflat = None if bias is None else _c(bias.detach().reshape(-1).float())
        y = conv(_c(x), residual=None if residual is None else _c(residual), bias=bflat, splitk=True)
        ctx.cfg = (stride, pad, dil, act, precision, tuple(weight.shape), None if bias is None else tuple(bias.shape))
        ctx.bank = bank
        ctx.save_for_backward(x, weight, y if act != E.ACT_NONE else None)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, pad, dil, act, precision, wshape, bshape = ctx.cfg
        x, weight, y = ctx.saved_tensors
        bank = ctx.bank
        dy = _c(dy)
        dpre = act_bwd(dy, y, act) if act != E.ACT_NONE else dy
        b, ho, wo, cout = dpre.shape
        cin, kh, kw = bank.cin, bank.kh, bank.kw
        dx = dw = db = dres = None
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(_c(x), dpre, cout, kh, kw, stride, pad, dil, param_shape=wshape)       # the parameter's layout
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(b * ho * wo, cout, cout, 0, dpre).reshape(bshape)
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dpre
        if ctx.needs_input_grad[0]:
            # input gradient = forward conv of (zero-stuffed) dpre with the flipped, transposed weights (the bank's second operand)
            _, h, w_, cx = x.shape
            lh, lw = h - dil * (kh - 1) + 2 * pad, w_ - dil * (kw - 1) + 2 * pad
            if stride == 1 and cout % 4 == 0:
                src = dpre
            else:
                c4 = (cout + 3) // 4 * 4
                src = torch.zeros(b, lh, lw, c4, dtype=torch.float32, device=dy.device)
                src[:, ::stride, ::stride][:, :ho, :wo, :cout] = dpre
            back = bank.conv(True, 1, dil * (kh - 1) - pad, dil, E.ACT_NONE)
            dx = torch.zeros(b, h, w_, cx, dtype=torch.float32, device=dy.device) if cx != cin else None
            dx = back(src, out=dx, splitk=True)
        return dx, dw, db, dres, None, None, None, None, None


def conv(x, weight, bias=None, residual=None, stride=1, pad=0, dil=1, act=E.ACT_NONE, precision="f32"):
    return ConvFn.apply(x, weight, bias, residual, stride, pad, dil, act, precision)


class PReLUFn(torch.autograd.Function):
    """nn.PReLU() with a single slope (pspnet.py:33)"""

    @staticmethod
    def forward(ctx, x, alpha):
        x = _c(x)
        y = torch.empty_like(x)
        a = float(alpha.detach().reshape(-1)[0])
        _lib.check(_lib.lib().ape_prelu_f32(_lib.dptr(x, torch.float32), _lib.dptr(y), x.numel(), a, _st()), "ape_prelu_f32")
        ctx.save_for_backward(x, alpha)
        ctx.a = a
        return y

    @staticmethod
    def backward(ctx, dy):
        x, alpha = ctx.saved_tensors
        dy = _c(dy)
        dx = act_bwd(dy, x, E.ACT_PRELU, ctx.a) if ctx.needs_input_grad[0] else None
        da = None
        if ctx.needs_input_grad[1]:
            da = torch.empty(1, dtype=torch.float32, device=dy.device)
            scratch = torch.empty(1024, dtype=torch.float32, device=dy.device)
            rc = _lib.lib().ape_prelu_dalpha_f32(_lib.dptr(dy, torch.float32), _lib.dptr(x), _lib.dptr(da), dy.numel(), _lib.dptr(scratch), _st())
            _lib.check(rc, "ape_prelu_dalpha_f32")
            da = da.reshape(alpha.shape)
        return dx, da


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (extractors.py:91)"""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return E.maxpool3x3s2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        b, h, w, c = x.shape
        dx = torch.empty_like(x)
        rc = _lib.lib().ape_maxpool3x3s2_bwd_nhwc_f32(_lib.dptr(x, torch.float32), _lib.dptr(_c(dy), torch.float32), _lib.dptr(dx), b, h, w, c, _st())
        _lib.check(rc, "ape_maxpool3x3s2_bwd_nhwc_f32")
        return dx


class AdaptiveAvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d((S, S)) (pspnet.py:16)"""

    @staticmethod
    def forward(ctx, x, s):
        ctx.shape, ctx.s = tuple(x.shape), s
        return E.adaptive_avgpool(_c(x), s)

    @staticmethod
    def backward(ctx, dy):
        b, h, w, c = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        rc = _lib.lib().ape_adaptive_avgpool_bwd_nhwc_f32(_lib.dptr(_c(dy), torch.float32), _lib.dptr(dx), b, h, w, c, ctx.s, _st())
        _lib.check(rc, "ape_adaptive_avgpool_bwd_nhwc_f32")
        return dx, None


class BilinearFn(torch.autograd.Function):
    """F.upsample(x, size / scale_factor, mode='bilinear', align_corners) (pspnet.py:22,37)"""

    @staticmethod
    def forward(ctx, x, ho, wo, align_corners):
        ctx.shape, ctx.o, ctx.ac = tuple(x.shape), (ho, wo), int(bool(align_corners))
        return E.bilinear(_c(x), ho, wo, align_corners)

    @staticmethod
    def backward(ctx, dy):
        b, h, w, c = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        rc = _lib.lib().ape_bilinear_bwd_nhwc_f32(_lib.dptr(_c(dy), torch.float32), _lib.dptr(dx), b, h, w, c, ctx.o[0], ctx.o[1], ctx.ac, _st())
        _lib.check(rc, "ape_bilinear_bwd_nhwc_f32")
        return dx, None, None, None


class LogSoftmaxRowsFn(torch.autograd.Function):
    """nn.LogSoftmax over the channel (last NHWC) axis (pspnet.py:55)"""

    @staticmethod
    def forward(ctx, x):
        shape = x.shape
        y = E.log_softmax_rows(_c(x).view(-1, shape[-1])).view(shape)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        c = y.shape[-1]
        dx = torch.empty_like(y)
        rc = _lib.lib().ape_log_softmax_bwd_rows_f32(_lib.dptr(_c(dy), torch.float32), _lib.dptr(y), _lib.dptr(dx), y.numel() // c, c, _st())
        _lib.check(rc, "ape_log_softmax_bwd_rows_f32")
        return dx


class GatherRowsFn(torch.autograd.Function):
    """torch.gather(emb, 2, choose) of network.py:100-102 in NHWC: x[B,R,C], index[B,n] -> [B,n,C]"""

    @staticmethod
    def forward(ctx, x, index):
        ctx.shape = tuple(x.shape)
        index = _c(index)
        ctx.save_for_backward(index)
        return E.gather_rows(_c(x), index)

    @staticmethod
    def backward(ctx, dy):
        (index,) = ctx.saved_tensors
        b, r, c = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        rc = _lib.lib().ape_scatter_add_rows_f32(_lib.dptr(_c(dy), torch.float32), _lib.dptr(index, torch.int64), _lib.dptr(dx), b, r,
                                                 index.shape[1], c, _st())
        _lib.check(rc, "ape_scatter_add_rows_f32")
        return dx, None


class MeanRowsFn(torch.autograd.Function):
    """torch.nn.AvgPool1d(num_points) (network.py:64,164): x[B,n,C] -> [B,C]"""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return E.mean_rows(_c(x))

    @staticmethod
    def backward(ctx, dy):
        b, n, c = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        _lib.check(_lib.lib().ape_mean_rows_bwd_f32(_lib.dptr(_c(dy), torch.float32), _lib.dptr(dx), b, n, c, _st()), "ape_mean_rows_bwd_f32")
        return dx


def _adds_grad(pred_r, pred_t, points, model, target, pred_c, dis, std, g, symmetric, full, w):
    n, m = pred_r.shape[0], model.shape[0]
    d_r, d_t = torch.empty_like(pred_r), torch.empty_like(pred_t)
    d_c = torch.empty(n, dtype=torch.float32, device=pred_r.device) if full else None
    g = _c(g.detach().float().reshape(1))
    rc = _lib.lib().ape_adds_grad_f32(_lib.dptr(pred_r, torch.float32), _lib.dptr(pred_t, torch.float32), _lib.dptr(points),
                                      _lib.dptr(model, torch.float32), _lib.dptr(target, torch.float32), _lib.dptr(pred_c),
                                      _lib.dptr(dis, torch.float32), _lib.dptr(std), _lib.dptr(g), n, m, int(bool(symmetric)),
                                      int(bool(full)), float(w), _lib.dptr(d_r), _lib.dptr(d_t), _lib.dptr(d_c), _st())
    _lib.check(rc, "ape_adds_grad_f32")
    return d_r, d_t, d_c


class PoseLossFn(torch.autograd.Function):
    """loss.py:12-53: r[N,4], t[N,3], c[N], points[N,3], model[M,3], target[M,3] -> (loss, out9, pred[N,M,3]); only `loss`
    is differentiable (dis[which], new_points, new_target are detached in the reference, loss.py:73)."""

    @staticmethod
    def forward(ctx, r, t, c, points, model, target, symmetric, w):
        r, t, c = _c(r.detach()), _c(t.detach()), _c(c.detach())
        dis, std, pred = E.adds_dis(r, t, points, model, target, symmetric, want_pred=True)
        out9, _ = E.adds_select(dis, std, c, r, t, points, w)
        ctx.save_for_backward(r, t, c, points, model, target, dis, std)
        ctx.cfg = (symmetric, w)
        loss = out9[0].clone()
        ctx.mark_non_differentiable(out9, pred)
        return loss, out9, pred

    @staticmethod
    def backward(ctx, g_loss, _g9, _gp):
        r, t, c, points, model, target, dis, std = ctx.saved_tensors
        symmetric, w = ctx.cfg
        d_r, d_t, d_c = _adds_grad(r, t, points, model, target, c, dis, std, g_loss, symmetric, True, w)
        return d_r, d_t, d_c, None, None, None, None, None


class RefineDisFn(torch.autograd.Function):
    """loss_refiner.py:12-47: r[1,4], t[1,3], model[M,3], target[M,3] -> (dis, pred[1,M,3]); dis is what train.py:222 back-propagates"""

    @staticmethod
    def forward(ctx, r, t, model, target, symmetric):
        r, t = _c(r.detach()), _c(t.detach())
        dis, _, pred = E.adds_dis(r, t, None, model, target, symmetric, want_pred=True)
        ctx.save_for_backward(r, t, model, target, dis)
        ctx.symmetric = symmetric
        ctx.mark_non_differentiable(pred)
        return dis[0].clone(), pred

    @staticmethod
    def backward(ctx, g_dis, _gp):
        r, t, model, target, dis = ctx.saved_tensors
        d_r, d_t, _ = _adds_grad(r, t, None, model, target, None, dis, None, g_dis, ctx.symmetric, False, 0.0)
        return d_r, d_t, None, None, None


class Adam:
    """optim.Adam(params, lr) of train.py:109,113 on ape_adam_step_f32 (torch defaults: betas (0.9, 0.999), eps 1e-8,
    weight_decay 0).  `step()` skips parameters whose .grad is None, `zero_grad()` drops the gradients."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.state = {}

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        jobs = (_lib.AdamJob * len(live))()
        keep = []
        b1, b2 = ctypes.c_float(self.betas[0]).value, ctypes.c_float(self.betas[1]).value      # the float32 values the kernel multiplies with
        for i, p in enumerate(live):
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p.data), "exp_avg_sq": torch.zeros_like(p.data)}
            st["step"] += 1
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = _c(g.float())
                keep.append(g)
            if not p.data.is_contiguous():
                raise RuntimeError("Adam needs contiguous parameters")
            jobs[i] = _lib.AdamJob(param=_lib.dptr(p.data, torch.float32), grad=_lib.dptr(g), exp_avg=_lib.dptr(st["exp_avg"]),
                                   exp_avg_sq=_lib.dptr(st["exp_avg_sq"]), n=p.numel(), bc1=1.0 - b1 ** st["step"],
                                   bc2_sqrt=math.sqrt(1.0 - b2 ** st["step"]))
        rc = _lib.lib().ape_adam_step_multi_f32(len(live), jobs, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                                float(self.weight_decay), _st())
        _lib.check(rc, "ape_adam_step_multi_f32")
        refresh_banks(live)          # the conv operands of the updated parameters, one launch
