"""Which stage of the pose path gives different results when several crop-size buckets run side by side on their own streams?  Every bucket's
stages are computed alone first (reference), then all buckets concurrently for some rounds; differing stages are listed."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "40")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from autoposeestimation_amd import engine as E, synthetic as S
from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet
est = PoseNet(1000, 12); est.load_state_dict(S.posenet_state_dict(12, 0)); est = est.cuda().eval(); est.set_precision("bf16x3")
pl = est.plan()
refn = PoseRefineNet(1000, 12); refn.load_state_dict(S.refiner_state_dict(12, 0)); refn = refn.cuda().eval(); refn.set_precision("bf16x3")
g = torch.Generator().manual_seed(0)
rgb = torch.randint(0, 256, (8, 480, 640, 3), generator=g, dtype=torch.uint8).cuda()
buckets = []
for k, (hc, wc, nobj) in enumerate([(80, 80, 9), (120, 160, 30), (160, 160, 12), (240, 240, 20), (320, 400, 7), (120, 160, 5), (240, 240, 3), (80, 80, 20)]):
    rects = torch.stack([torch.randint(0, 8, (nobj,), generator=g), torch.randint(0, 480 - hc, (nobj,), generator=g), torch.randint(0, 640 - wc, (nobj,), generator=g)], 1).int().cuda()
    choose = torch.stack([torch.randperm(hc * wc, generator=g)[:1000].sort().values for _ in range(nobj)]).cuda()
    pts = torch.randn(nobj, 1000, 4, generator=g).cuda(); pts[..., 3] = 0
    obj = torch.randint(0, 12, (nobj,), generator=g).cuda()
    buckets.append((hc, wc, rects, choose, pts, obj))

def stages(bk):
    hc, wc, rects, choose, pts, obj = bk
    b, n = pts.shape[0], 1000
    img4 = E.U8Frames(rgb, rects, hc, wc, div255=False)
    out = {}
    p2 = pl.cnn.features(img4, stop_before_up3=True)
    out["A features"] = (p2.t if isinstance(p2, E.S32) else p2).clone()
    if getattr(pl, "up3_matrix", None) is None:
        pl.up3_matrix = E.conv3x3_as_matrix(pl.cnn.up3)
    pg = E.ups_patch_gather(p2, choose)
    out["B patches"] = pg.clone()
    gg = pl.up3_matrix(pg).view(b, n, 64)
    out["C up3"] = gg.clone()
    emb = E.log_softmax_rows(pl.cnn.final(gg.view(b, n, 1, 64)).view(b, n, 32))
    out["D emb"] = emb.clone()
    heads, emb2 = est.forward_batch(img4, pts, choose, obj)
    out["E heads"] = heads.clone()
    pose, _, newp = E.pose_select(heads, pts)
    out["F pose0"], out["F newp"] = pose.clone(), newp.clone()
    for it in range(2):
        o = refn.forward_batch(newp, emb2, obj)
        out["G refiner %d" % it] = o.clone()
    E.pose_compose(pose, o[:, 0:4], o[:, 4:7])
    out["H pose"] = pose.clone()
    return out

ref = [stages(bk) for bk in buckets]
ref = [stages(bk) for bk in buckets]          # (second pass: every lazy operand exists)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in buckets]
bad = {}
rounds = int(os.environ.get("ROUNDS", "12"))
for r in range(rounds):
    outs = []
    start = torch.cuda.Event(); start.record()
    for bk, st in zip(buckets, streams):
        st.wait_event(start)
        with torch.cuda.stream(st):
            outs.append(stages(bk))
    torch.cuda.synchronize()
    for k, (o, w) in enumerate(zip(outs, ref)):
        for name in w:
            if not torch.equal(o[name], w[name]):
                bad.setdefault((name, k, buckets[k][0], buckets[k][1]), []).append(r)
print("rounds", rounds)
if not bad:
    print("no stage differs")
for key in sorted(bad):
    print(key, bad[key])
