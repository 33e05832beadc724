"""Generate tests/golden/bgsub.npz by RUNNING the reference's background_subtraction.utils.get_mask_prediction
(imported from /root/reference through tools/ref_shim.py) on a small synthetic directory.  Build container only.

What stands in for what (nothing here is copied reference code):
  * the smp Unet (third-party, not installed) is replaced by a RECORDING model: `.predict(x)` stores the 7-channel tensor the
    reference built and returns seeded logits-like probabilities, so the golden pins the reference's feature block
    (:721-819) and its do_cca (:199-222) on known network outputs;
  * torchvision's ToTensor / Normalize (stubbed by the shim) are given their documented definitions
    (HWC uint8 -> CHW float32 / 255; (x - mean) / std per channel);
  * cv2.connectedComponents (OpenCV, not installed) -> scipy.ndimage.label with the 8-neighbourhood (PARITY UNPINNED for
    this call);
  * numpy >= 2 refuses `reference_point != np.array([])` (:735): the reference point is passed as an object whose `!=`
    answers what numpy 1.x answered (True for a 3-vector, False for the empty default).
Also checks oracle.bgsub_oracle.pil_rgb_to_hsv against the installed Pillow over all 2^24 colours.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import ref_shim  # noqa: E402

H, W, NF = 48, 64, 3


class _RefPoint(np.ndarray):
    """3-vector whose `!= np.array([])` is True (numpy 1.x semantics of the reference's test at :735)."""

    def __ne__(self, other):
        return True


class _NoRefPoint:
    def __ne__(self, other):
        return False


class _ToTensor:
    def __call__(self, pic):
        return torch.from_numpy(np.ascontiguousarray(pic.transpose(2, 0, 1))).float().div(255)


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)

    def __call__(self, t):
        return t.clone().sub_(self.mean[:, None, None]).div_(self.std[:, None, None])


class _Recorder:
    def __init__(self, preds):
        self.preds, self.xs, self.i = preds, [], 0

    def to(self, device):
        return self

    def eval(self):
        return self

    def predict(self, x):
        self.xs.append(x[0].numpy().copy())
        p = torch.from_numpy(self.preds[self.i % len(self.preds)])[None]
        self.i += 1
        return p


def _frames(rng):
    """background + foreground frames: noise background, a coloured box in the foreground, depth with zeros and far pixels"""
    b_rgb = rng.integers(0, 256, (NF, H, W, 3), dtype=np.uint8)
    f_rgb = b_rgb.copy()
    f_rgb = np.clip(f_rgb.astype(np.int32) + rng.integers(-6, 7, f_rgb.shape), 0, 255).astype(np.uint8)
    b_depth = rng.integers(500, 1800, (NF, H, W)).astype(np.uint16)
    f_depth = b_depth.copy()
    for i in range(NF):
        y0, x0 = 8 + 5 * i, 10 + 7 * i
        f_rgb[i, y0:y0 + 20, x0:x0 + 24] = rng.integers(0, 256, (20, 24, 3), dtype=np.uint8)
        f_depth[i, y0:y0 + 20, x0:x0 + 24] = rng.integers(300, 1200, (20, 24)).astype(np.uint16)
    for d in (b_depth, f_depth):
        d[rng.random(d.shape) < 0.05] = 0
    return f_rgb, b_rgb, f_depth, b_depth


def _preds(rng):
    """seeded 2-class 'probabilities' with several blobs so do_cca has a choice to make"""
    p = np.zeros((NF, 2, H, W), dtype=np.float32)
    for i in range(NF):
        fg = rng.random((H, W)).astype(np.float32) * 0.35
        fg[5:20, 6:30] += 0.6
        fg[28:44, 34:60] += 0.55 + 0.03 * i
        fg[40:46, 2:9] += 0.7
        fg = np.clip(fg, 0, 1)
        p[i, 1], p[i, 0] = fg, 1 - fg
    return p


def main():
    ref_shim.install()
    import scipy.ndimage as ndi
    import background_subtraction.utils as ref

    ref.transforms.ToTensor = _ToTensor
    ref.transforms.Normalize = _Normalize

    def connected_components(mask, connectivity=8):
        assert connectivity == 8
        labels, n = ndi.label(mask != 0, structure=np.ones((3, 3), dtype=bool))
        return n + 1, labels.astype(np.int32)

    ref.cv2.connectedComponents = connected_components
    rng = np.random.default_rng(20260103)
    f_rgb, b_rgb, f_depth, b_depth = _frames(rng)
    preds = _preds(rng)
    r2e = np.eye(4)
    r2e[:3, 3] = [400.0, -120.0, 650.0]
    he = np.eye(4)
    he[:3, 3] = [10.0, 20.0, 30.0]
    reference_point = np.array([450.0, -100.0, -240.0])
    out = {"f_rgb": f_rgb, "b_rgb": b_rgb, "f_depth": f_depth, "b_depth": b_depth, "preds": preds,
           "robot2endEff_tf": r2e.reshape(-1), "hand_eye_calibration": he.reshape(-1), "reference_point": reference_point}
    with tempfile.TemporaryDirectory() as root:
        obj = os.path.join(root, "data_generation", "data", "thing")
        for name, rgb, depth in (("background", b_rgb, b_depth), ("foreground", f_rgb, f_depth)):
            d = os.path.join(obj, name)
            os.makedirs(d)
            for i in range(NF):
                Image.fromarray(rgb[i], "RGB").save(os.path.join(d, "%06d.color.png" % i))
                Image.fromarray(depth[i]).save(os.path.join(d, "%06d.depth.png" % i))
                with open(os.path.join(d, "%06d.meta.json" % i), "w") as f:
                    json.dump({"robot2endEff_tf": r2e.reshape(-1).tolist(), "hand_eye_calibration": he.reshape(-1).tolist()}, f)
        for tag, rp in (("gate", reference_point.view(_RefPoint)), ("nogate", _NoRefPoint())):
            rec = _Recorder(preds)
            ref.get_default_model = lambda root_: rec
            ref.get_mask_prediction("thing", root, reference_point=rp, use_cuda=False)
            out["x_" + tag] = np.stack(rec.xs)
            out["label_" + tag] = np.stack([np.array(Image.open(os.path.join(root, "label_generator", "data", "thing", "foreground",
                                                                              "%06d.pred.label.png" % i))) for i in range(NF)])
    # do_cca alone on a case with exact ties (two identical blobs: the first in raster order must win) and an empty frame
    tie = np.zeros((2, 2, H, W), dtype=np.float32)
    tie[:, 0] = 1.0
    tie[0, 1, 4:10, 4:10] = 3.0
    tie[0, 1, 30:36, 40:46] = 3.0
    out["cca_in"] = tie
    out["cca_out"] = ref.do_cca(torch.from_numpy(tie), cuda=False)

    # Pillow HSV: exhaustive check of the restatement, sampled fixture for the CPU tests
    from oracle import bgsub_oracle as O
    v = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    hsv = np.array(Image.fromarray(rgb, "RGB").convert("HSV"))
    mism = int((O.pil_rgb_to_hsv(rgb) != hsv).sum())
    print("pil_rgb_to_hsv vs Pillow %s over 2^24 colours: %d mismatches" % (Image.__version__, mism))
    assert mism == 0
    pick = rng.choice(1 << 24, 4096, replace=False)
    out["hsv_rgb"] = rgb.reshape(-1, 3)[pick]
    out["hsv_ref"] = hsv.reshape(-1, 3)[pick]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bgsub.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
