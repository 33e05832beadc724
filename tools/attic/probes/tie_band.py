"""which class the device arg-max returns inside the two-softmax tie band (development probe)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from autoposeestimation_amd import engine as E
C = 5
x = torch.tensor(0.1)
for k in range(1, 15):
    x = torch.nextafter(x, torch.tensor(1.0))
    logits = torch.full((1, 1, 16, 8), -4.0)
    logits[..., 1] = 0.1
    logits[..., 3] = float(x)
    want = int(F.softmax(F.softmax(logits[..., :C], -1), -1).argmax(-1)[0, 0, 0])
    lab, _ = E.seg_argmax(logits.cuda(), C, double_softmax=True)
    feat = torch.zeros(1, 1, 16, 64); feat[..., 0] = 1.0
    w = torch.zeros(C, 64); w[:, 0] = torch.tensor([-4.0, 0.1, -4.0, float(x), -4.0])
    l2, _ = E.seg_head(feat.cuda(), w.cuda().contiguous(), torch.zeros(C).cuda(), True)
    print(k, "torch", want, "seg_argmax", int(lab[0, 0, 0]), "head", int(l2[0, 0, 0]))
