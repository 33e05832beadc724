"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the background-subtraction labelling step of the reference,
background_subtraction/utils.py: `get_mask_prediction`'s per-frame block (:721-828) and `do_cca` (:199-222).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(autoposeestimation_amd/background_subtraction) never does.

Pinning:
  * `pil_rgb_to_hsv` restates Pillow's 8-bit RGB -> HSV conversion (third-party, the reference pins Pillow 8.1.2, README.md:46;
    call sites :727-731).  tools/gen_golden_bgsub.py compares it with the installed Pillow over ALL 2^24 colours (0 mismatches)
    and commits a 4096-colour sample as tests/golden/bgsub.npz['hsv_rgb' / 'hsv_ref'].
  * `subtraction_features` is pinned by running the reference's own `get_mask_prediction` (imported from /root/reference by the
    generator) on a synthetic directory with a recording model in place of the smp network: the tensor it hands to
    `model.predict` is the golden ('x_gate', 'x_nogate').
  * `do_cca`: the reference calls cv2.connectedComponents (OpenCV 4.5.1, not installed) -> connected-component labelling is
    restated with scipy.ndimage.label (8-connectivity, raster-order label ids like OpenCV's) -- PARITY UNPINNED for that one
    call, as for pipeline/utils.py:450; the golden masks come from the reference's do_cca running on that
    restated labelling.
"""
import numpy as np

DEFAULT_MEAN = [0.040278014, 0.04060352, 0.038310923, 0.0381776, 0.03656849, 0.03636289, 0.03556486]      # :670-673
DEFAULT_STD = [0.059689723, 0.05965291, 0.056203008, 0.05619316, 0.054657422, 0.054514673, 0.05377024]


def pil_rgb_to_hsv(rgb):
    """rgb[...,3] u8 -> hsv[...,3] u8 exactly as PIL's Image.convert('HSV') (libImaging Convert.c rgb2hsv_row, which follows
    colorsys.rgb_to_hsv with float intermediates, the hue wrap `fmod(h/6 + 1, 1)` in double, truncation to 0..255)."""
    rgb = np.asarray(rgb)
    r = rgb[..., 0].astype(np.int32)
    g = rgb[..., 1].astype(np.int32)
    b = rgb[..., 2].astype(np.int32)
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    grey = maxc == minc
    cr = (maxc - minc).astype(np.float32)
    cr[grey] = 1
    mx = maxc.astype(np.float32)
    mx[maxc == 0] = 1
    s = cr / mx
    rc = (maxc - r).astype(np.float32) / cr
    gc = (maxc - g).astype(np.float32) / cr
    bc = (maxc - b).astype(np.float32) / cr
    h = np.where(r == maxc, (bc - gc).astype(np.float32),
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc).astype(np.float32))).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    uh[grey] = 0
    us[grey] = 0
    return np.stack([uh, us, maxc], -1).astype(np.uint8)


def depth_gate(measure_dist):
    """:747-752 -- (min, max) of the accepted sensor range; measure_dist None/0 -> (0, 1500)."""
    if not measure_dist:
        return 0.0, float(int(1500))
    return measure_dist - 150, measure_dist + 150


def measure_distance(meta, reference_point):
    """:735-743 -- distance of the camera (robot2endEff_tf . hand_eye_calibration) from the reference point, in mm"""
    r2e = np.array(meta.get("robot2endEff_tf")).reshape(4, 4)
    he = np.array(meta.get("hand_eye_calibration")).reshape(4, 4)
    pos = np.dot(r2e, he)[:3, 3]
    return np.linalg.norm(np.asarray(reference_point) - pos)


def subtraction_features(f_rgb, b_rgb, f_depth, b_depth, gate, mean=None, std=None):
    """f_rgb/b_rgb[H,W,3] u8, f_depth/b_depth[H,W] u16, gate=(min,max) -> (x_u8[H,W,7], x[7,H,W] f32) (:721-819)"""
    mean = DEFAULT_MEAN if mean is None else mean
    std = DEFAULT_STD if std is None else std
    fr = np.asarray(f_rgb, dtype=np.float64)
    br = np.asarray(b_rgb, dtype=np.float64)
    fh = pil_rgb_to_hsv(f_rgb).astype(np.float64)
    bh = pil_rgb_to_hsv(b_rgb).astype(np.float64)
    fd = np.array(f_depth, dtype=np.float64)
    bd = np.array(b_depth, dtype=np.float64)
    dmin, dmax = gate
    fd[fd > dmax] = 0           # :756-759
    bd[bd > dmax] = 0
    fd[fd < dmin] = 0
    bd[bd < dmin] = 0
    fd[bd == 0] = 0             # :762-763
    bd[fd == 0] = 0
    x = np.concatenate((np.abs(fr - br), np.abs(fh - bh), np.abs(fd - bd)[..., None]), axis=2)     # :766-808
    x_u8 = (x.astype(np.int64) & 255).astype(np.uint8)      # np.array(x, dtype=np.uint8) (:811): truncation, wraps modulo 256
    t = x_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255)         # torchvision ToTensor
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return x_u8, ((t - m) / s).astype(np.float32)                           # Normalize: sub_(mean).div_(std)


def softmax(x, axis):
    x = np.asarray(x, dtype=np.float32)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return (e / e.sum(axis=axis, keepdims=True)).astype(np.float32)


def label8(mask):
    """cv2.connectedComponents(mask, connectivity=8): labels 1.. in raster order of each component's first pixel."""
    from scipy import ndimage
    labels, n = ndimage.label(mask != 0, structure=np.ones((3, 3), dtype=bool))
    return n + 1, labels.astype(np.int32)


def do_cca(predicted):
    """predicted[B,C,H,W] (the model's output, softmax already applied by smp's activation) -> [B,H,W] f64 in {0,1}:
    softmax again (:200), argmax / max over channels, 8-connected components of argmax != 0, keep the component with the
    largest SUMMED max-probability (first one on ties; label 1 when there is none) (:204-219)."""
    predicted = softmax(predicted, 1)
    out = []
    for pred in predicted:
        pred = pred.transpose(1, 2, 0)
        mask = np.array(np.argmax(pred, axis=2), dtype=np.uint8)
        mask2 = np.array(np.max(pred, axis=2))
        _, labels = label8(mask)
        biggest, biggest_score = 1, 0
        for u in np.unique(labels)[1:]:
            score = np.sum(mask2[labels == u])
            if score > biggest_score:
                biggest_score, biggest = score, u
        o = np.zeros(mask.shape)
        o[labels == biggest] = 1
        out.append(o[None])
    return np.concatenate(out, axis=0)
