"""Drop-in for DenseFusion/lib/network.py (+ pspnet.py, extractors.py): `PoseNet`, `PoseRefineNet`, `PSPNet`
with the reference's constructor/forward signatures and state-dict key names, executed by hand-written gfx950
kernels through the C ABI (include/ape_hip.h).  There is no CPU path: forward() on host tensors raises.

    estimator = PoseNet(num_points=1000, num_obj=12); estimator.load_state_dict(torch.load('pose_model.pth'))
    estimator.to('cuda').eval()
    pred_r, pred_t, pred_c, emb = estimator(img[1,3,Hc,Wc], points[1,N,3], choose[1,1,N], idx[1,1])   # network.py:95,132

Differences from the reference that do not change results beyond fp32 rounding:
  * the three heads' 1408-wide first layer is split into  W[:, :384] . pointfeat  +  (W[:, 384:] . ap_x + b)  -- the
    1024-vector broadcast of network.py:67-68 becomes a per-crop bias instead of a materialised 1408 x N tensor;
  * the PSP bottleneck (pspnet.py:22-24) is evaluated as  W_f . feats + sum_s up(W_s . prior_s)  (1x1 conv and bilinear
    resize commute), so the 2560-channel concat is never built;
  * up_1 / up_2 (bilinear x2 then 3x3 conv, pspnet.py:27-37) mix channels at LOW resolution (1x1 conv to 9*Cout channels) and
    then resize + shift + sum the nine taps (engine.UpConv): conv and resize are linear, so the order is free;
  * the heads evaluate only the selected object's output rows (network.py:119-126 computes all, then index_selects);
  * the 32-channel log-softmax embedding is evaluated only at the `choose`d pixels (network.py:100-102 gathers them
    from the full map);  `PSPNet.forward` still returns the full map;
  * `forward_batch` generalises the reference's batch-1-only forward (network.py:123, b = 0) to B independent crops.
"""
import os

import torch
import torch.nn as nn

from autoposeestimation_amd import engine as E

_BLOCKS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}


def _register(root, dotted, tensor):
    """Create `root.a.b.c` as nn.Parameter, making bare nn.Module containers on the way (state-dict key = dotted)."""
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: the MI355X path has no CPU fallback -- move the module and its inputs to 'cuda'"
                           % (what, t.device))


class _HipModule(nn.Module):
    """nn.Module holding reference-named parameters; the device plan (repacked weights) is rebuilt lazily whenever the
    parameters may have changed (load_state_dict / .to / .cuda)."""

    def __init__(self):
        super().__init__()
        self._plan = None
        self._drop = None
        self.precision = "f32"

    def set_precision(self, precision):
        """Operand precision of this network's dense contractions: 'f32' | 'bf16x3' | 'bf16' (see engine.PRECISIONS)."""
        if precision not in E.PRECISIONS:
            raise ValueError(precision)
        self.precision = precision
        self._plan = None
        return self

    def _apply(self, fn, *a, **k):
        self._plan = None
        self._pcache = None
        r = super()._apply(fn, *a, **k)
        self.invalidate_banks()
        return r

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Parameter, nn.Module)):      # a replaced parameter / sub-module: cached look-ups and the plan are stale
            self.__dict__["_pcache"] = None
            self.__dict__["_plan"] = None
        super().__setattr__(name, value)

    def register_parameter(self, name, param):
        self.__dict__["_pcache"] = None
        self.__dict__["_plan"] = None
        super().register_parameter(name, param)

    def invalidate_banks(self):
        """drop the packed conv operands kept on this module's parameters (autograd.WeightBank): they are rebuilt from the parameters on
        the next training forward"""
        from autoposeestimation_amd import autograd as A
        A.invalidate_banks(self.parameters())

    def sync_banks(self):
        """Top of every training forward: re-pack the kept conv operands of ALL this module's parameters from the parameters' storage (one
        launch).  A bank is keyed on the parameter's torch version counter, which writes through `.data`, raw pointers (the library's
        Adam) or collectives on `.data` never move -- so the training forward does not trust it: whatever wrote the parameters since the last
        forward, the convolutions below run on what is in memory now."""
        from autoposeestimation_amd import autograd as A
        A.refresh_banks(list(self.parameters()))

    def param(self, name):
        """get_parameter(name) through a dict built once (the training forward looks ~75 parameters up per step); an entry is checked
        against its owner's `_parameters` slot, so a parameter replaced by assignment -- on this module or on a holder below it -- is seen"""
        c = self.__dict__.get("_pcache")
        if c is None:
            c = self.__dict__["_pcache"] = {}
        hit = c.get(name)
        if hit is not None and hit[0]._parameters.get(hit[1]) is hit[2]:
            return hit[2]
        path, _, leaf = name.rpartition(".")
        owner = self.get_submodule(path) if path else self
        p = owner._parameters[leaf]
        c[name] = (owner, leaf, p)
        return p

    def train(self, mode=True):
        """train(True): parameters become leaves of the tape (autoposeestimation_amd/autograd.py) and forward() runs the
        unfused training graph; either way the cached inference plan is dropped (an optimizer may have changed the weights)."""
        self._plan = None
        if mode:
            for p in self.parameters():
                p.requires_grad_(True)
        self.invalidate_banks()
        return super().train(mode)

    def set_dropout_masks(self, masks):
        """Fix the Dropout2d channel multipliers of the next training forwards (parity tests): dict with 'drop_1' [B,1024],
        'drop_2a' [B,256], 'drop_2b' [B,64] holding 0 or 1/(1-p); None -> sampled per forward like nn.Dropout2d."""
        self._drop = masks
        return self

    def load_state_dict(self, state_dict, strict=True, **k):
        self._plan = None
        r = super().load_state_dict(state_dict, strict=strict, **k)
        self.invalidate_banks()
        return r

    def _sd(self):
        return {k: v for k, v in self.state_dict().items()}

    def plan(self):
        if self._plan is None:
            sd = self._sd()
            dev = next(iter(sd.values())).device
            _need_cuda(next(iter(sd.values())), type(self).__name__ + " parameters")
            self._plan = self._build_plan(sd, dev)
        return self._plan


# ----------------------------------------------------------------------------------------------------------------
# PSPNet  (pspnet.py:40-77 over the BN-free dilated ResNet of extractors.py:78-124)
# ----------------------------------------------------------------------------------------------------------------
def _pspnet_param_shapes(backend, n_classes, in_channels=3):
    from autoposeestimation_amd.synthetic import pspnet_spec
    return pspnet_spec(backend, n_classes, in_channels)


class _PSPPlan:
    def __init__(self, sd, prefix, backend, dev, precision="f32"):
        g = lambda k: sd[prefix + k]  # noqa: E731
        _Conv = lambda *a, **k: E.Conv(*a, precision=precision, **k)  # noqa: E731
        self.precision = precision
        self.stem = _Conv(g("feats.conv1.weight"), None, 2, 3, 1, E.ACT_RELU, device=dev)
        self.blocks = []
        inplanes = 64
        for li, (planes, nblk, stride, dil) in enumerate(zip((64, 128, 256, 512), _BLOCKS[backend], (1, 2, 1, 1), (1, 1, 2, 4)), 1):
            for b in range(nblk):
                first = b == 0
                s = stride if first else 1
                d = 1 if first else dil      # extractors.py:107 does not forward `dilation` to the first block
                p = f"feats.layer{li}.{b}."
                c1 = _Conv(g(p + "conv1.weight"), None, s, d, d, E.ACT_RELU, device=dev)
                c2 = _Conv(g(p + "conv2.weight"), None, 1, d, d, E.ACT_RELU, device=dev)   # relu after the residual add
                down = None
                if first and (stride != 1 or inplanes != planes):
                    down = _Conv(g(p + "downsample.0.weight"), None, s, 0, 1, E.ACT_NONE, device=dev)
                self.blocks.append((c1, c2, down))
            inplanes = planes
        wb = g("psp.bottleneck.weight")           # [1024, 2560, 1, 1] = [prior_1 | prior_2 | prior_3 | prior_6 | feats]
        # prior branch i: bottleneck columns i applied to stage conv i applied to the pooled map -- two 1x1 convs with nothing
        # between them (pspnet.py:15-16,22-24), i.e. ONE 512 -> 1024 map whose matrix is their product (formed once, in fp64)
        self.prior = []
        for i in range(4):
            ws = g(f"psp.stages.{i}.1.weight").detach().double().reshape(512, 512)
            wbi = wb[:, i * 512:(i + 1) * 512].detach().double().reshape(-1, 512)
            self.prior.append(_Conv((wbi @ ws).float(), None, device=dev))
        self.bott_feats = _Conv(wb[:, 2048:2560], g("psp.bottleneck.bias"), act=E.ACT_RELU, device=dev)
        # up_1 / up_2: channel mixing at low resolution + tap gather (4x fewer flops, no upsampled tensor); up_3 (64 -> 64 at
        # full resolution) would write a 9 x 64-channel half-resolution tensor larger than what it saves, so it stays direct
        self.up = [E.UpConv(g(f"{n}.conv.1.weight"), g(f"{n}.conv.1.bias"), float(g(f"{n}.conv.2.weight").reshape(-1)[0]),
                            device=dev, precision=precision) for n in ("up_1", "up_2")]
        self.up3 = _Conv(g("up_3.conv.1.weight"), g("up_3.conv.1.bias"), 1, 1, 1, E.ACT_PRELU,
                         alpha=float(g("up_3.conv.2.weight").reshape(-1)[0]), device=dev)
        # the segmentor's up_3 in up_1 / up_2's low-resolution form, ONE kernel with the head (engine.UpConv.seg_head, upconv_fused.hip):
        # 4x fewer matrix flops than the direct 3x3 conv on the up-sampled map
        self.up3_low = (E.UpConv(g("up_3.conv.1.weight"), g("up_3.conv.1.bias"), float(g("up_3.conv.2.weight").reshape(-1)[0]),
                                 device=dev, precision=precision, fma=True) if precision == "bf16x3" else None)
        self.final = _Conv(g("final.0.weight"), g("final.0.bias"), device=dev)

    def _s32_graph(self, x):
        """the S32 graph's 3x3 kernel tiles the 1/8-resolution map in 16x16 pixels: worth it only when those tiles are mostly full
        (the 480x640 segmentor: 60x80 -> 94 %; a 160x160 crop: 20x20 -> 39 %, which stays on the flattened-M kernels)"""
        h8, w8 = -(-x.shape[1] // 8), -(-x.shape[2] // 8)
        return self.precision == "bf16x3" and E.USE_S32 and h8 * w8 >= 0.8 * (-(-h8 // 16) * -(-w8 // 16) * 256)

    def label_score(self, x, head_w, head_b, double_softmax=True):
        """x[B,H,W,4] -> (label u8, score f32)[B,H,W]: features with the classification head (first C rows of the final 1x1 conv +
        softmax(+softmax) + arg-max) fused into up_3 -- the full-resolution 64-channel activation is never stored.  On the S32 graph
        up_2 hands up_3 a pre-split map and up_3 + head is the one low-resolution kernel; otherwise the head rides in the epilogue of the
        direct 3x3 conv on the (virtually) up-sampled map."""
        if self.up3_low is not None and E.USE_UPFUSE and self._s32_graph(x) and head_w.shape[0] <= 16:
            p2 = self._features_s32(x, None, True, up2_fmt=E.FMT_S32)
            if self.up3_low.fusable(p2):
                return self.up3_low.seg_head(p2, head_w, head_b, double_softmax)
            return E.conv_seg_head(self.up3, p2.to_f32(), head_w, head_b, double_softmax, upsample2x=True)
        return E.conv_seg_head(self.up3, self.features(x, stop_before_up3=True), head_w, head_b, double_softmax, upsample2x=True)

    def features(self, x, taps=None, stop_before_up3=False):
        """x[B,H,W,4] (RGB + zero pad) -> up_3 activation [B,H,W,64]"""
        if self._s32_graph(x):
            return self._features_s32(x, taps, stop_before_up3)
        y = E.stem_pool(self.stem, x)
        for c1, c2, down in self.blocks:
            res = y if down is None else down(y)
            y = c2(c1(y), residual=res)
        f = y
        b, h, w, _ = f.shape
        pools = E.adaptive_avgpool_multi(f, (2, 3, 6))
        # the 1x1 pool is the mean of the four 2x2 bins when they tile the map evenly (one workgroup per frame otherwise)
        pools[1] = E.adaptive_avgpool(pools[2], 1) if h % 2 == 0 and w % 2 == 0 else E.adaptive_avgpool(f, 1)
        zs = E.conv1x1_multi(self.prior, [pools[s] for s in (1, 2, 3, 6)])       # (the four stage convolutions in one launch)
        p = self.bott_feats(f, residual=E.psp_prior_sum(zs, h, w))
        if taps is not None:
            taps["feats"], taps["psp"] = f, p
        for i, up in enumerate(self.up):
            p = up(p)
            if taps is not None:
                taps["up_%d" % (i + 1)] = p
        if stop_before_up3:
            return p
        p = self.up3(p, upsample2x=True)     # bilinear x2 fused into the conv's halo load when the LDS-halo kernel applies
        if taps is not None:
            taps["up_3"] = p
        return p

    def _features_s32(self, x, taps, stop_before_up3, up2_fmt=E.FMT_F32):
        """The same graph with PRE-SPLIT ("S32", include/ape_hip.h) activations between the split-bf16 layers from layer 2 on: every
        producer writes the bf16 hi | lo pair its consumers' matrix cores take, the 3x3 and 1x1 layers stream it HBM -> LDS by
        LDS-DMA (conv3x3_halo_s32.hip, conv_gemm_s32.hip).  The MFMA operands are bit-identical to the fp32-activation graph; residual
        adds and the pools read hi + lo (2^-17 relative) instead of the fp32 value.  fp32 stays where a VALU kernel consumes the
        tensor: the stem / layer 1 (64 channels: the narrow halo kernel), the 9*Cout tap tensors of up_1 / up_2, up_2's output
        (bilinearly re-sampled inside up_3's halo load)."""
        S = E.FMT_S32
        y = E.stem_pool(self.stem, x)
        fold = E.USE_PSP_FOLD                             # the last layer-4 conv then writes a map with 64 spare channels for the prior coefficients
        for bi, (c1, c2, down) in enumerate(self.blocks):
            wide = c1.cout >= 128
            if not wide:                                  # layer 1: fp32 in and out
                res = y if down is None else down(y)
                y = c2(c1(y), residual=res)
                continue
            res = y if down is None else down(y)          # fp32 (a residual only) from either kernel
            t = c1(y, out_fmt=S)                          # stride-2 first conv of layer 2: conv_gemm.hip with an S32 epilogue
            out = None
            if fold and bi == len(self.blocks) - 1:
                out = E.S32(torch.empty(t.shape[0], t.shape[1], t.shape[2], c2.cout + E.PSP_FOLD_K, dtype=torch.float32, device=x.device))
            y = c2(t, residual=res, out=out, out_fmt=S)
        f = y
        b, h, w, _ = f.shape
        nf = self.bott_feats.cin
        pools = E.adaptive_avgpool_multi(f, (2, 3, 6), channels=nf)             # (the map's 64 spare channels of the folded form are not pooled)
        pools[1] = E.adaptive_avgpool(pools[2], 1) if h % 2 == 0 and w % 2 == 0 else E.adaptive_avgpool(f.to_f32()[..., :nf].contiguous(), 1)
        zs = E.conv1x1_multi(self.prior, [pools[s] for s in (1, 2, 3, 6)])       # (the four stage convolutions in one launch)
        if fold:
            p = E.psp_bottleneck_folded(self.bott_feats, f, zs, out_fmt=S)
        else:
            p = self.bott_feats(f, residual=E.psp_prior_sum(zs, h, w), out_fmt=S)
        if taps is not None:
            taps["feats"], taps["psp"] = f.to_f32()[..., :nf].contiguous(), p.to_f32()
        p = self.up[0](p, out_fmt=S)
        if taps is not None:
            taps["up_1"] = p.to_f32()
        p = self.up[1](p, out_fmt=up2_fmt)          # (S32 only for the fused up_3 + head of label_score)
        if taps is not None:
            taps["up_2"] = p
        if stop_before_up3:
            return p
        p = self.up3(p, upsample2x=True)
        if taps is not None:
            taps["up_3"] = p
        return p


class PSPNet(_HipModule):
    """pspnet.py:40-77.  forward(x[B,3,H,W]) -> log_softmax(final 1x1 conv) [B,32,H,W] (eval semantics)."""

    def __init__(self, n_classes=21, sizes=(1, 2, 3, 6), psp_size=512, deep_features_size=256, backend="resnet18",
                 pretrained=False, in_channels=3):
        super().__init__()
        if backend not in _BLOCKS or tuple(sizes) != (1, 2, 3, 6) or psp_size != 512:
            raise NotImplementedError("BasicBlock backends (resnet18/34) with sizes (1,2,3,6) only")
        self.backend = backend
        self.in_channels = in_channels
        for key, shape in _pspnet_param_shapes(backend, n_classes, in_channels):
            _register(self, key, torch.zeros(shape))

    def _build_plan(self, sd, dev):
        return _PSPPlan(sd, "", self.backend, dev, self.precision)

    def forward_nhwc(self, x4, logits_only=False):
        """x4[B,H,W,4] -> [B,H,W,32] log-softmax (or raw logits)"""
        pl = self.plan()
        logits = pl.final(pl.features(x4))
        return logits if logits_only else E.log_softmax_rows(logits)

    def forward(self, x):
        _need_cuda(x, "input")
        x4 = torch.zeros(x.shape[0], x.shape[2], x.shape[3], 4, dtype=torch.float32, device=x.device)
        x4[..., :3] = x.permute(0, 2, 3, 1)
        return self.forward_nhwc(x4).permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------------------------------------------
# Training graphs: the same networks written op by op over the tape of autograd.py, unfused, so that
# loss.backward() (train.py:221-225) reaches every reference parameter.  Dropout2d (pspnet.py:48,50: p = 0.3 / 0.15) is a
# per-(sample, channel) multiplier.
# ----------------------------------------------------------------------------------------------------------------
def _dropout2d(x, masks, key, p):
    if masks is not None and key in masks:
        m = masks[key].to(device=x.device, dtype=torch.float32)
    else:
        m = torch.bernoulli(torch.full((x.shape[0], x.shape[3]), 1.0 - p, device=x.device)) / (1.0 - p)
    return x * m[:, None, None, :]


def _pspnet_train(mod, prefix, backend, x, masks, precision):
    """x[B,H,W,4] -> log-softmax embedding [B,H,W,32] (pspnet.py:64-77 in train mode)"""
    from autoposeestimation_amd import autograd as A
    P = lambda k: mod.param(prefix + k)  # noqa: E731
    cv = lambda *a, **k: A.conv(*a, precision=precision, **k)  # noqa: E731
    y = A.MaxPoolFn.apply(cv(x, P("feats.conv1.weight"), stride=2, pad=3, act=E.ACT_RELU))
    inplanes = 64
    for li, (planes, nblk, stride, dil) in enumerate(zip((64, 128, 256, 512), _BLOCKS[backend], (1, 2, 1, 1), (1, 1, 2, 4)), 1):
        for b in range(nblk):
            first = b == 0
            s, d = (stride if first else 1), (1 if first else dil)
            k = "feats.layer%d.%d." % (li, b)
            res = cv(y, P(k + "downsample.0.weight"), stride=s) if first and (stride != 1 or inplanes != planes) else y
            h = cv(y, P(k + "conv1.weight"), stride=s, pad=d, dil=d, act=E.ACT_RELU)
            y = cv(h, P(k + "conv2.weight"), residual=res, pad=d, dil=d, act=E.ACT_RELU)
        inplanes = planes
    f = y
    _, h, w, _ = f.shape
    priors = [A.BilinearFn.apply(cv(A.AdaptiveAvgPoolFn.apply(f, s), P("psp.stages.%d.1.weight" % i)), h, w, False)
              for i, s in enumerate((1, 2, 3, 6))]
    p = cv(torch.cat(priors + [f], 3), P("psp.bottleneck.weight"), P("psp.bottleneck.bias"), act=E.ACT_RELU)
    p = _dropout2d(p, masks, "drop_1", 0.3)
    for name, dkey in (("up_1", "drop_2a"), ("up_2", "drop_2b"), ("up_3", None)):
        u = A.BilinearFn.apply(p, 2 * p.shape[1], 2 * p.shape[2], True)
        u = cv(u, P(name + ".conv.1.weight"), P(name + ".conv.1.bias"), pad=1)
        p = A.PReLUFn.apply(u, P(name + ".conv.2.weight"))
        if dkey is not None:
            p = _dropout2d(p, masks, dkey, 0.15)
    return A.LogSoftmaxRowsFn.apply(cv(p, P("final.0.weight"), P("final.0.bias")))


def _feat_train(mod, x4, emb, refine, precision):
    """PoseNetFeat / PoseRefineNetFeat (network.py:39-68, 136-168): x4[1,N,1,4], emb[1,N,1,32] -> pf[1,N,1,384], ap[1,1024]"""
    from autoposeestimation_amd import autograd as A
    g = lambda k: (mod.param("feat.%s.weight" % k), mod.param("feat.%s.bias" % k))  # noqa: E731
    cv = lambda x, k: A.conv(x, *g(k), act=E.ACT_RELU, precision=precision)  # noqa: E731
    c1, e1 = cv(x4, "conv1"), cv(emb, "e_conv1")
    c2, e2 = cv(c1, "conv2"), cv(e1, "e_conv2")
    pf = torch.cat([c1, e1, c2, e2], 3)
    x6 = cv(cv(pf if refine else torch.cat([c2, e2], 3), "conv5"), "conv6")
    return pf, A.MeanRowsFn.apply(x6.view(1, -1, 1024))


def _select_rows(w, b, obj, k):
    """rows obj*k..obj*k+k of the last layer (network.py:123-125 index_selects the object's outputs), padded to 4 rows"""
    w = w.reshape(w.shape[0], -1)[obj * k:(obj + 1) * k]
    b = b[obj * k:(obj + 1) * k]
    if k < 4:
        w = torch.cat([w, torch.zeros(4 - k, w.shape[1], dtype=w.dtype, device=w.device)], 0)
        b = torch.cat([b, torch.zeros(4 - k, dtype=b.dtype, device=b.device)], 0)
    return w, b


# ----------------------------------------------------------------------------------------------------------------
# PointNet trunks and heads
# ----------------------------------------------------------------------------------------------------------------
S32_MIN_ROWS = int(os.environ.get("APE_POSE_S32_MIN_ROWS", "16384"))      # rows (crops x points) from which the PointNet trunk / heads take the pre-split route (256-row tiles; below, the 128-row blocks of conv_gemm fill the chip better)


class _FeatPlan:
    """PoseNetFeat / PoseRefineNetFeat (network.py:39-68, 136-168) writing straight into the concatenated buffer
    pf[B*N, 384] = [conv1(x) 64 | e_conv1(emb) 64 | conv2 128 | e_conv2 128]."""

    def __init__(self, sd, dev, refine, precision="f32"):
        g = lambda k: (sd[f"feat.{k}.weight"], sd[f"feat.{k}.bias"])  # noqa: E731
        kw = dict(act=E.ACT_RELU, device=dev, precision=precision)
        self.conv1 = E.Conv(*g("conv1"), **kw)
        self.e_conv1 = E.Conv(*g("e_conv1"), **kw)
        self.conv2 = E.Conv(*g("conv2"), **kw)
        self.e_conv2 = E.Conv(*g("e_conv2"), **kw)
        self.conv5 = E.Conv(*g("conv5"), **kw)
        self.conv6 = E.Conv(*g("conv6"), **kw)
        self.refine = refine
        self.precision = precision
        self.pf_s32 = None

    def __call__(self, x4, emb):
        """x4[B,N,4], emb[B,N,32] -> pf[B,N,1,384], ap[B,1024]"""
        b, n, _ = x4.shape
        x4 = x4.view(b, n, 1, 4)
        emb = emb.view(b, n, 1, 32)
        pf = torch.empty(b, n, 1, 384, dtype=torch.float32, device=x4.device)
        self.conv1(x4, out=pf, yoff=0)
        self.e_conv1(emb, out=pf, yoff=64)
        self.conv2(pf, out=pf, xoff=0, yoff=128)
        self.e_conv2(pf, out=pf, xoff=64, yoff=256)
        # Enough rows to fill 256-row tiles: the 1x1 layers run on PRE-SPLIT activations (conv_gemm_s32.hip, LDS-DMA operands) -- the same
        # MFMA operands as the on-the-fly split of conv_gemm.hip, so every value below is bit for bit what the fp32 route gives
        # (tests/test_gpu_posenet.py); `self.pf_s32` hands the split point features to the estimator's heads
        self.pf_s32 = None
        if self.precision == "bf16x3" and E.USE_S32 and b * n >= S32_MIN_ROWS:
            pfs = self.pf_s32 = E.S32.from_f32(pf)
            x5 = self.conv5(pfs, xoff=0 if self.refine else 128, out_fmt=E.FMT_S32)
            x6 = self.conv6(x5)
        else:
            x5 = self.conv5(pf, xoff=0 if self.refine else 128)
            x6 = self.conv6(x5)
        return pf, E.mean_rows(x6.view(b, n, 1024))


class PoseNet(_HipModule):
    """DenseFusion/lib/network.py:70-132."""

    def __init__(self, num_points, num_obj):
        super().__init__()
        from autoposeestimation_amd.synthetic import posenet_state_dict
        self.num_points, self.num_obj = num_points, num_obj
        for key, t in posenet_state_dict(num_obj, seed=0).items():   # key names + shapes; values are placeholders
            _register(self, key, torch.zeros_like(t))

    def _build_plan(self, sd, dev):
        pl = type("Plan", (), {})()
        pr = self.precision
        pl.cnn = _PSPPlan(sd, "cnn.model.module.", "resnet18", dev, pr)
        pl.feat = _FeatPlan(sd, dev, refine=False, precision=pr)
        w1 = torch.cat([sd[f"conv1_{h}.weight"] for h in "rtc"], 0)[:, :, 0]      # [1920, 1408]
        b1 = torch.cat([sd[f"conv1_{h}.bias"] for h in "rtc"], 0)
        pl.l1_point = E.Conv(w1[:, :384], None, act=E.ACT_RELU, device=dev, precision=pr)           # per-point part
        pl.l1_global = E.Conv(w1[:, 384:], b1, act=E.ACT_NONE, device=dev, precision=pr)            # per-crop bias from ap_x
        pl.l2 = [E.Conv(sd[f"conv2_{h}.weight"], sd[f"conv2_{h}.bias"], act=E.ACT_RELU, device=dev, precision=pr) for h in "rtc"]
        pl.l3 = [E.Conv(sd[f"conv3_{h}.weight"], sd[f"conv3_{h}.bias"], act=E.ACT_RELU, device=dev, precision=pr) for h in "rtc"]
        pl.l4 = [t.detach().to(dev, torch.float32).reshape(t.shape[0], -1).contiguous()
                 for h in "rtc" for t in (sd[f"conv4_{h}.weight"], sd[f"conv4_{h}.bias"])]
        E.allow_splitk(pl)          # (batch-1 frames: the crop's small-M layers may split K; never the segmentor's)
        return pl

    def forward_batch(self, img4, points4, choose, obj, taps=None):
        """img4[B,Hc,Wc,4] f32 (normalised RGB + 0), points4[B,N,4], choose[B,N] i64, obj[B] i64
        -> heads[B,N,8] (qw,qx,qy,qz,tx,ty,tz,c), emb[B,N,32]"""
        pl = self.plan()
        for t, name in ((img4, "img"), (points4, "points"), (choose, "choose"), (obj, "obj")):      # (img4: the normalised crops or engine.U8Frames)
            _need_cuda(t, name)
        b, hc, wc, _ = img4.shape
        n = points4.shape[1]
        if taps is None and hc % 2 == 0 and wc % 2 == 0:
            # only N of the crop's Hc*Wc embedding pixels are kept (network.py:100-102): up_3 is evaluated at those alone, as
            # one contraction over the 3x3 patches of the (virtual) up-sampled up_2 output gathered at the chosen pixels
            p2 = pl.cnn.features(img4, stop_before_up3=True)
            if getattr(pl, "up3_matrix", None) is None:
                pl.up3_matrix = E.conv3x3_as_matrix(pl.cnn.up3)
            g = pl.up3_matrix(E.ups_patch_gather(p2, choose)).view(b, n, 64)
        else:
            up3 = pl.cnn.features(img4, taps)
            g = E.gather_rows(up3.view(b, hc * wc, 64), choose)             # [B,N,64]
        emb = E.log_softmax_rows(pl.cnn.final(g.view(b, n, 1, 64)).view(b, n, 32))
        pf, ap = pl.feat(points4, emb)
        gb = pl.l1_global(ap.view(b, 1, 1, 1024)).view(b, 1920)            # W[:, 384:] . ap_x + b
        h3 = torch.empty(b, n, 1, 384, dtype=torch.float32, device=img4.device)
        if pl.feat.pf_s32 is not None:                                          # pre-split route (see _FeatPlan): h1, h2 stay split
            h1 = pl.l1_point(pl.feat.pf_s32, bias=gb, bias_bstride=1920, out_fmt=E.FMT_S32)
            h2 = E.S32(torch.empty(b, n, 1, 768, dtype=torch.float32, device=img4.device))
            for i in range(3):
                pl.l2[i](h1, out=h2, xoff=640 * i, yoff=256 * i, out_fmt=E.FMT_S32)
                pl.l3[i](h2, out=h3, xoff=256 * i, yoff=128 * i)
        else:
            h1 = pl.l1_point(pf, bias=gb, bias_bstride=1920)                    # [B,N,1,1920]
            h2 = torch.empty(b, n, 1, 768, dtype=torch.float32, device=img4.device)
            for i in range(3):
                pl.l2[i](h1, out=h2, xoff=640 * i, yoff=256 * i)
                pl.l3[i](h2, out=h3, xoff=256 * i, yoff=128 * i)
        heads = E.head_select(h3.view(b * n, 384), 0, 128, 256, *pl.l4, obj, b, n, 128)
        if taps is not None:
            taps["pf"], taps["ap"], taps["emb"] = pf, ap, emb
        return heads, emb

    def forward(self, img, x, choose, obj):
        """Reference signature (network.py:95): img[1,3,Hc,Wc], x[1,N,3], choose[1,1,N] i64, obj[1,1] i64 ->
        (out_rx[1,N,4], out_tx[1,N,3], out_cx[1,N,1], emb[1,32,N])"""
        _need_cuda(img, "img")
        if img.shape[0] != 1:
            raise ValueError("reference forward is batch-1 (network.py:123); use forward_batch for B crops")
        o = int(obj.reshape(-1)[0])
        if not 0 <= o < self.num_obj:
            raise IndexError("obj index %d out of range" % o)
        img4 = torch.zeros(1, img.shape[2], img.shape[3], 4, dtype=torch.float32, device=img.device)
        img4[..., :3] = img.permute(0, 2, 3, 1)
        if self.training:
            self.sync_banks()
            return self._forward_train(img4, E.pad3to4(x.float().contiguous()), choose.reshape(1, -1).contiguous(), o)
        heads, emb = self.forward_batch(img4, E.pad3to4(x.float().contiguous()), choose.reshape(1, -1).contiguous(),
                                        obj.reshape(1).contiguous())
        return (heads[:, :, 0:4].contiguous(), heads[:, :, 4:7].contiguous(), heads[:, :, 7:8].contiguous(),
                emb.transpose(1, 2).contiguous())


def _posenet_forward_train(self, img4, points4, choose, o):
    """train-mode PoseNet.forward (network.py:95-132) on the tape"""
    from autoposeestimation_amd import autograd as A
    pr = self.precision
    _, hc, wc, _ = img4.shape
    n = points4.shape[1]
    emb_map = _pspnet_train(self, "cnn.model.module.", "resnet18", img4, self._drop, pr)
    emb = A.GatherRowsFn.apply(emb_map.view(1, hc * wc, 32), choose)                      # [1,N,32]
    pf, ap = _feat_train(self, points4.view(1, n, 1, 4), emb.view(1, n, 1, 32), False, pr)
    outs = []
    for h, k in (("r", 4), ("t", 3), ("c", 1)):
        w1 = self.param("conv1_%s.weight" % h)[:, :, 0]                              # [640,1408]
        gb = A.conv(ap.view(1, 1, 1, 1024), w1[:, 384:], self.param("conv1_%s.bias" % h), precision=pr)
        y = A.conv(pf, w1[:, :384], gb, act=E.ACT_RELU, precision=pr)
        for l in (2, 3):
            y = A.conv(y, self.param("conv%d_%s.weight" % (l, h)), self.param("conv%d_%s.bias" % (l, h)),
                       act=E.ACT_RELU, precision=pr)
        w4, b4 = _select_rows(self.param("conv4_%s.weight" % h), self.param("conv4_%s.bias" % h), o, k)
        y = A.conv(y, w4, b4, act=E.ACT_SIGMOID if h == "c" else E.ACT_NONE, precision=pr)
        outs.append(y.view(1, n, 4)[:, :, :k])
    return outs[0], outs[1], outs[2], emb.transpose(1, 2)


PoseNet._forward_train = _posenet_forward_train      # defined after _pspnet_train / _feat_train, which it calls


class PoseRefineNet(_HipModule):
    """DenseFusion/lib/network.py:170-206."""

    def __init__(self, num_points, num_obj):
        super().__init__()
        from autoposeestimation_amd.synthetic import refiner_state_dict
        self.num_points, self.num_obj = num_points, num_obj
        for key, t in refiner_state_dict(num_obj, seed=0).items():
            _register(self, key, torch.zeros_like(t))

    def _build_plan(self, sd, dev):
        pl = type("Plan", (), {})()
        pr = self.precision
        pl.feat = _FeatPlan(sd, dev, refine=True, precision=pr)
        pl.l1 = E.Conv(torch.cat([sd["conv1_r.weight"], sd["conv1_t.weight"]], 0),
                       torch.cat([sd["conv1_r.bias"], sd["conv1_t.bias"]], 0), act=E.ACT_RELU, device=dev, precision=pr)   # 1024 -> 512|512
        pl.l2 = [E.Conv(sd[f"conv2_{h}.weight"], sd[f"conv2_{h}.bias"], act=E.ACT_RELU, device=dev, precision=pr) for h in "rt"]
        pl.l3 = [t.detach().to(dev, torch.float32).contiguous()
                 for h in "rt" for t in (sd[f"conv3_{h}.weight"], sd[f"conv3_{h}.bias"])]
        E.allow_splitk(pl)
        return pl

    def forward_batch(self, points4, emb, obj):
        """points4[B,N,4], emb[B,N,32], obj[B] -> out[B,8] (qw,qx,qy,qz,tx,ty,tz,0)"""
        pl = self.plan()
        for t, name in ((points4, "points"), (emb, "emb"), (obj, "obj")):
            _need_cuda(t, name)
        b = points4.shape[0]
        _, ap = pl.feat(points4, emb)
        h1 = pl.l1(ap.view(b, 1, 1, 1024))                                   # [B,1,1,1024]
        h2 = torch.empty(b, 1, 1, 256, dtype=torch.float32, device=points4.device)
        pl.l2[0](h1, out=h2, xoff=0, yoff=0)
        pl.l2[1](h1, out=h2, xoff=512, yoff=128)
        return E.head_select(h2.view(b, 256), 0, 128, 0, *pl.l3, None, None, obj, b, 1, 128).view(b, 8)

    def forward(self, x, emb, obj):
        """Reference signature (network.py:187): x[1,N,3], emb[1,32,N], obj[1,1] -> (out_rx[1,4], out_tx[1,3])"""
        _need_cuda(x, "x")
        o = int(obj.reshape(-1)[0])
        if not 0 <= o < self.num_obj:
            raise IndexError("obj index %d out of range" % o)
        if self.training:
            self.sync_banks()
            return self._forward_train(E.pad3to4(x.float().contiguous()), emb.transpose(1, 2).contiguous(), o)
        out = self.forward_batch(E.pad3to4(x.float().contiguous()), emb.transpose(1, 2).contiguous(), obj.reshape(1).contiguous())
        return out[:, 0:4].contiguous(), out[:, 4:7].contiguous()

    def _forward_train(self, points4, emb, o):
        """train-mode PoseRefineNet.forward (network.py:187-206) on the tape: points4[1,N,4], emb[1,N,32]"""
        from autoposeestimation_amd import autograd as A
        pr = self.precision
        n = points4.shape[1]
        _, ap = _feat_train(self, points4.view(1, n, 1, 4), emb.reshape(1, n, 1, 32), True, pr)
        outs = []
        for h, k in (("r", 4), ("t", 3)):
            y = ap.view(1, 1, 1, 1024)
            for l in (1, 2):
                y = A.conv(y, self.param("conv%d_%s.weight" % (l, h)), self.param("conv%d_%s.bias" % (l, h)),
                           act=E.ACT_RELU, precision=pr)
            w3, b3 = _select_rows(self.param("conv3_%s.weight" % h), self.param("conv3_%s.bias" % h), o, k)
            outs.append(A.conv(y, w3, b3, precision=pr).view(1, 4)[:, :k])
        return outs[0], outs[1]


class ModifiedResnet(nn.Module):
    """network.py:27-37 -- kept for import compatibility; PoseNet above owns the PSPNet weights directly
    (the `cnn.model.module.` key prefix is what nn.DataParallel produced in the reference)."""

    def __init__(self, usegpu=True):
        super().__init__()
        self.model = nn.Module()
        self.model.add_module("module", PSPNet(backend="resnet18"))

    def forward(self, x):
        return self.model.module(x)
