"""pose_select (one small workgroup per crop) beside ONE other stage of the pose path on a second stream: which neighbour corrupts it?"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "40")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import engine as E, synthetic as S
from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet
est = PoseNet(1000, 12); est.load_state_dict(S.posenet_state_dict(12, 0)); est = est.cuda().eval(); est.set_precision("bf16x3")
refn = PoseRefineNet(1000, 12); refn.load_state_dict(S.refiner_state_dict(12, 0)); refn = refn.cuda().eval(); refn.set_precision("bf16x3")
pl = est.plan()
g = torch.Generator().manual_seed(0)
nobj = 30
heads = torch.randn(nobj, 1000, 8, generator=g).cuda()
pts = torch.randn(nobj, 1000, 4, generator=g).cuda(); pts[..., 3] = 0
want = E.pose_select(heads, pts)[2].clone()
rgb = torch.randint(0, 256, (8, 480, 640, 3), generator=g, dtype=torch.uint8).cuda()
hc, wc = 120, 160
rects = torch.stack([torch.randint(0, 8, (nobj,), generator=g), torch.randint(0, 480 - hc, (nobj,), generator=g), torch.randint(0, 640 - wc, (nobj,), generator=g)], 1).int().cuda()
choose = torch.stack([torch.randperm(hc * wc, generator=g)[:1000].sort().values for _ in range(nobj)]).cuda()
obj = torch.randint(0, 12, (nobj,), generator=g).cuda()
img4 = E.U8Frames(rgb, rects, hc, wc, div255=False)
p2 = pl.cnn.features(img4, stop_before_up3=True)
pl.up3_matrix = E.conv3x3_as_matrix(pl.cnn.up3)
emb = torch.randn(nobj, 1000, 32, generator=g).cuda()
x6 = torch.randn(nobj, 1000, 1024, device="cuda")
neigh = {
    "cnn features": lambda: pl.cnn.features(img4, stop_before_up3=True),
    "patch gather + up3 matrix": lambda: pl.up3_matrix(E.ups_patch_gather(p2, choose)),
    "point features (gemm_s32 route)": lambda: pl.feat(pts, emb),
    "mean_rows": lambda: E.mean_rows(x6),
    "refiner": lambda: refn.forward_batch(pts, emb, obj),
    "estimator whole": lambda: est.forward_batch(img4, pts, choose, obj),
    "pose_select itself": lambda: E.pose_select(heads, pts),
    "nothing": lambda: None,
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, f in neigh.items():
    f(); torch.cuda.synchronize()
    bad = 0
    for r in range(int(os.environ.get("ROUNDS", "30"))):
        outs = []
        with torch.cuda.stream(s2):
            for _ in range(3): f()
        with torch.cuda.stream(s1):
            for _ in range(20): outs.append(E.pose_select(heads, pts)[2])
        with torch.cuda.stream(s2):
            for _ in range(3): f()
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, want)) for o in outs)
    print("%-34s wrong pose_select outputs: %d" % (name, bad))

# ---- what does a wrong output look like?
f = neigh["cnn features"]
shown = 0
for r in range(40):
    outs = []
    with torch.cuda.stream(s2):
        for _ in range(3): f()
    with torch.cuda.stream(s1):
        for _ in range(20): outs.append(E.pose_select(heads, pts)[2])
    torch.cuda.synchronize()
    for o in outs:
        if not torch.equal(o, want) and shown < 6:
            d = (o != want).reshape(nobj, -1)
            crops = d.any(1).nonzero().flatten().tolist()
            idx = (o != want).reshape(-1).nonzero().flatten()
            print("crops", crops, "elements", int(idx.numel()), "first", int(idx[0]), "last", int(idx[-1]),
                  "got", o.reshape(-1)[idx[:4]].tolist(), "want", want.reshape(-1)[idx[:4]].tolist())
            shown += 1
print("inputs intact:", torch.equal(E.pose_select(heads, pts)[2], want))
