"""cProfile of the estimator-phase training step (development aid): python tools/prof_train.py [precision]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from autoposeestimation_amd import synthetic as S
from autoposeestimation_amd.autograd import Adam
from autoposeestimation_amd.DenseFusion.lib.loss import Loss
from autoposeestimation_amd.DenseFusion.lib.loss_refiner import Loss_refine
from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet
from autoposeestimation_amd.DenseFusion.tools.train import train_step
dev = "cuda:0"
N, M, NOBJ, HC, WC = 1000, 500, 12, 160, 160
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
est, ref = PoseNet(N, NOBJ), PoseRefineNet(N, NOBJ)
est.load_state_dict(S.posenet_state_dict(NOBJ, seed=1)); ref.load_state_dict(S.refiner_state_dict(NOBJ, seed=2))
est.to(dev); ref.to(dev); est.set_precision(prec); ref.set_precision(prec)
g = torch.Generator().manual_seed(0)
img = torch.randn(1, 3, HC, WC, generator=g); pts = torch.randn(1, N, 3, generator=g) * 0.1
choose = torch.randperm(HC * WC, generator=g)[:N].sort()[0].view(1, 1, N)
model = torch.randn(1, M, 3, generator=g) * 0.05; target = model + 0.01
data = (pts, choose, img, target, model, torch.tensor([[3]]))
crit, crit_r = Loss(M, [3]), Loss_refine(M, [3])
opt = SimpleNamespace(w=0.015, refine_start=False, iteration=2)
est.train()
optim = Adam(est.parameters(), lr=1e-4)
def step():
    optim.zero_grad(); train_step(est, ref, crit, crit_r, data, opt, dev); optim.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("estimator phase %s: %.2f ms per step" % (prec, (time.perf_counter() - t) / 10 * 1e3))
# phases
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return r, (t1 - t0) * 1e3, (t2 - t0) * 1e3
if os.environ.get("APE_CPROFILE", "1") == "1":
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
