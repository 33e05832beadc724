import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from autoposeestimation_amd import engine as E, synthetic as S
from autoposeestimation_amd.segmentation.utils import get_model
import bench
dev = torch.device("cuda:0")
seg = get_model("PsPNet", {"encoder_name": "resnet18", "encoder_weights": None, "activation": "softmax", "in_channels": 3, "classes": 13})
seg_sd = S.pspnet_state_dict("resnet18", seed=5, stem_gain=1.0); seg.load_state_dict(seg_sd); seg = seg.to(dev).eval()
def fit_frames(nk):
    return [S.synthetic_frame(900 + 7 * c + k, cls=c, box=(30 + 45 * c + 20 * (k % 2) + 10 * (k // 2), 20 + 60 * c + 90 * (k % 2) - 15 * (k // 2)), size=(126, 126)) for c in range(1, NC + 1) for k in range(nk)]
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 3
test_frames = [S.synthetic_frame(5000 + i, cls=1 + (i % NC), box=(int(np.random.default_rng(i).integers(20, 330)), int(np.random.default_rng(i + 99).integers(20, 490))), size=(126, 126)) for i in range(64)]
rects = torch.zeros(1, 3, dtype=torch.int32, device=dev)
def feats_of(rgb):
    x4 = E.preprocess_u8(torch.from_numpy(rgb[None]).to(dev), rects, 480, 640, True)
    return seg.plan().features(x4)[0].reshape(-1, 64)
T = [(feats_of(f[0]), f[2]) for f in test_frames]
for nk in (2,):
    ff = fit_frames(nk)
    F_fit = [feats_of(f[0]) for f in ff]
    for bgmul, fgw in ((6, 1), (18, 1), (18, 4), (18, 10)):
        feats, labels, wts = [], [], []
        for f, (rgb, _, label) in zip(F_fit, ff):
            flat = label.reshape(-1); fg = np.nonzero(flat)[0]; bgall = np.nonzero(flat == 0)[0]
            bg = bgall if bgmul >= 18 else np.random.default_rng(0).choice(bgall, size=bgmul * len(fg), replace=False)
            sel = np.concatenate([fg, bg]); rep = np.concatenate([np.full(len(fg), fgw), np.ones(len(bg), int)])
            sel = np.repeat(sel, rep)
            feats.append(f[torch.from_numpy(sel).to(dev)]); labels.append(torch.from_numpy(flat.astype(np.int64))[sel])
        w, b = S.fit_final_layer(torch.cat(feats), torch.cat(labels), 13, ridge=1e-2)
        w, b = w.to(dev), b.to(dev)
        spurious = 0; miss = []; small = 0
        for f, label in T:
            am = (f @ w.t() + b).argmax(1).cpu().numpy().reshape(480, 640)
            cls = int(label.max()); hist = np.bincount(am.reshape(-1), minlength=13)
            spurious += sum(1 for c in range(1, 13) if c != cls and hist[c] > 100)
            ys, xs = np.nonzero(am == cls)
            miss.append(int(((am != cls) & (label == cls)).sum()))
            if len(ys) and (ys.max() - ys.min() >= 160 or xs.max() - xs.min() >= 160): small += 1
        print("fit frames/class %d bgmul %d fgw %d: spurious classes over 64 frames = %d, mean missed px %.0f, tight bbox > 160: %d" % (nk, bgmul, fgw, spurious, np.mean(miss), small))
