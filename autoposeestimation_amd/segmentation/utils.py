"""Drop-in for the inference surface of segmentation/utils.py (reference :352-359): `nets`, `get_model`.

The reference builds its segmentor from the third-party `segmentation_models_pytorch==0.1.3` (smp.Unet / PSPNet /
Linknet), which is neither vendored in the reference tree nor installed here, so its arithmetic cannot be restated or
pinned.  DECISION: `get_model('PsPNet', cfg)` returns the reference's OWN in-repo PSPNet
(DenseFusion/lib/pspnet.py, BasicBlock encoder `cfg['encoder_name']` in {resnet18, resnet34}) executed by the gfx950
kernels; `predict()` returns the first `classes` channels of its `final` 1x1 conv with the configured activation.
smp-format Unet / LinkNet checkpoints cannot be honoured and raise NotImplementedError.

Training-only symbols of the reference module (jaccard_loss, IoU, ConfusionMatrix, transforms, animate*) are outside the
hot path and are not provided.
"""
import torch

from autoposeestimation_amd import engine as E
from autoposeestimation_amd.DenseFusion.lib.network import PSPNet, _need_cuda


class PsPNetSegmentor(PSPNet):
    """`model.predict(x[B,in_channels,H,W]) -> [B,classes,H,W]` like smp's SegmentationModel.predict (eval + no_grad + activation)."""

    def __init__(self, encoder_name="resnet18", encoder_weights=None, activation="softmax", in_channels=3, classes=2):
        if not 1 <= in_channels <= 8:
            raise NotImplementedError("in_channels must be in 1..8")
        if encoder_weights is not None:
            raise NotImplementedError("no pretrained encoder weights are available offline")
        if not 1 <= classes <= 32:
            raise ValueError("classes must be in 1..32 (the in-repo PSPNet's final conv has 32 channels, pspnet.py:54)")
        if activation not in (None, "softmax", "softmax2d", "identity"):
            raise NotImplementedError("activation %r" % (activation,))
        super().__init__(backend=encoder_name, in_channels=in_channels)
        self.classes, self.activation = classes, activation
        self._final_cls = None

    def _build_plan(self, sd, dev):
        pl = super()._build_plan(sd, dev)
        # only the first `classes` rows of the final 1x1 conv are ever needed
        self._final_cls = E.Conv(sd["final.0.weight"][:self.classes], sd["final.0.bias"][:self.classes], device=dev,
                                 precision=self.precision)
        self._head_w = sd["final.0.weight"][:self.classes].detach().to(dev, torch.float32).reshape(self.classes, 64).contiguous()
        self._head_b = sd["final.0.bias"][:self.classes].detach().to(dev, torch.float32).contiguous()
        return pl

    def logits_nhwc(self, x4):
        """x4[B,H,W,4] (ToTensor+Normalize'd RGB, zero 4th channel; [B,H,W,8] for in_channels > 4) -> logits[B,H,W,classes]"""
        pl = self.plan()
        return self._final_cls(pl.features(x4))

    def label_score_nhwc(self, x4, double_softmax=True):
        """x4[B,H,W,4] -> (label u8[B,H,W], score f32[B,H,W]): features -> fused head (final conv rows 0..classes-1 in exact
        fp32 + softmax(+softmax) + argmax); the logits tensor of `logits_nhwc` is never materialised."""
        pl = self.plan()
        b, h, w, _ = x4.shape
        # the kernels index a tensor with 32 bits: the largest of the pass is the full-resolution 64-channel map up_3 reads / the head
        # consumes (B x H x W x 64 elements; 109 frames of 480x640).  Larger batches go through in slices.
        max_b = max(1, ((1 << 31) - 1) // (h * w * 64))
        if b > max_b:
            parts = [self.label_score_nhwc(x4[i:i + max_b], double_softmax) for i in range(0, b, max_b)]
            return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        if self.classes > 16:
            return E.seg_argmax(self.logits_nhwc(x4), self.classes, double_softmax)
        return pl.label_score(x4, self._head_w, self._head_b, double_softmax)

    def predict(self, x):
        _need_cuda(x, "input")
        if x.shape[1] != self.in_channels:
            raise ValueError("expected %d input channels, got %d" % (self.in_channels, x.shape[1]))
        x4 = torch.zeros(x.shape[0], x.shape[2], x.shape[3], (self.in_channels + 3) // 4 * 4, dtype=torch.float32, device=x.device)
        x4[..., :self.in_channels] = x.permute(0, 2, 3, 1)
        logits = self.logits_nhwc(x4).permute(0, 3, 1, 2).contiguous()
        if self.activation in ("softmax", "softmax2d"):
            return torch.softmax(logits, dim=1)
        return logits


def _unavailable(name):
    def ctor(**cfg):
        raise NotImplementedError(
            "%s comes from segmentation_models_pytorch, which is not vendored in the reference and not installed; "
            "use get_model('PsPNet', cfg) (in-repo PSPNet on gfx950)" % name)
    return ctor


nets = {"Unet": _unavailable("smp.Unet"), "PsPNet": PsPNetSegmentor, "LinkNet": _unavailable("smp.Linknet")}


def get_model(name, segmentation_config):
    """reference segmentation/utils.py:356-359"""
    model = nets[name]
    model = model(**segmentation_config)
    return model
