import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from autoposeestimation_amd import engine as E, synthetic as S
from autoposeestimation_amd.segmentation.utils import get_model
import bench
dev = torch.device("cuda:0")
seg = get_model("PsPNet", {"encoder_name": "resnet18", "encoder_weights": None, "activation": "softmax", "in_channels": 3, "classes": 13})
seg_sd = S.pspnet_state_dict("resnet18", seed=5, stem_gain=1.0); seg.load_state_dict(seg_sd); seg = seg.to(dev).eval()
fit_frames = [S.synthetic_frame(900 + c, cls=c, box=(40 + 20 * c, 20 + 36 * c), size=(150, 150)) for c in range(1, 13)]
test_frames = bench.make_frames(12, 0)
rects = torch.zeros(1, 3, dtype=torch.int32, device=dev)
def feats_of(rgb):
    x4 = E.preprocess_u8(torch.from_numpy(rgb[None]).to(dev), rects, 480, 640, True)
    return seg.plan().features(x4)[0].reshape(-1, 64)
F_fit = [feats_of(f[0]) for f in fit_frames]
print("feature abs mean", F_fit[0].abs().mean().item(), "max", F_fit[0].abs().max().item())
for mode in ("balanced", "bg4", "all"):
    feats, labels = [], []
    for f, (rgb, _, label) in zip(F_fit, fit_frames):
        flat = label.reshape(-1)
        fg = np.nonzero(flat)[0]
        bgall = np.nonzero(flat == 0)[0]
        if mode == "balanced": bg = np.random.default_rng(0).choice(bgall, size=len(fg), replace=False)
        elif mode == "bg4": bg = np.random.default_rng(0).choice(bgall, size=4*len(fg), replace=False)
        else: bg = bgall
        sel = torch.from_numpy(np.concatenate([fg, bg]))
        feats.append(f[sel.to(dev)]); labels.append(torch.from_numpy(flat.astype(np.int64))[sel])
    for ridge in (1e-2, 1e-4):
        w, b = S.fit_final_layer(torch.cat(feats), torch.cat(labels), 13, ridge=ridge)
        w, b = w.to(dev), b.to(dev)
        errs = []
        for rgb, _, label in test_frames:
            lg = feats_of(rgb) @ w.t() + b
            am = lg.argmax(1).cpu().numpy().reshape(480, 640)
            cls = int(label.max())
            hist = np.bincount(am.reshape(-1), minlength=13)
            wrong_classes = [(c, int(hist[c])) for c in range(1, 13) if c != cls and hist[c] > 0]
            miss = int(((am != cls) & (label == cls)).sum())
            errs.append((cls, int(hist[cls]), miss, wrong_classes))
        print(mode, ridge, errs[:6])
