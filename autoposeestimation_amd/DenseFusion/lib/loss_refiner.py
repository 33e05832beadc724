"""Drop-in for DenseFusion/lib/loss_refiner.py (`Loss_refine`, reference :67-76 over loss_calculation :12-64).  When the
refiner's outputs require grad (training, train.py:219-222) `dis` sits on the tape of autoposeestimation_amd/autograd.py."""
import torch

from autoposeestimation_amd import engine as E


def loss_calculation(pred_r, pred_t, target, model_points, idx, points, num_point_mesh, sym_list):
    if not pred_r.is_cuda:
        raise RuntimeError("Loss_refine.forward needs device tensors: the MI355X path has no CPU fallback")
    r = pred_r.detach().float().reshape(1, 4).contiguous()
    t = pred_t.detach().float().reshape(1, 3).contiguous()
    tgt = target.detach().float().reshape(num_point_mesh, 3).contiguous()
    mdl = model_points.detach().float().reshape(num_point_mesh, 3).contiguous()
    n_in = points.shape[1]
    pts = points.detach().float().reshape(n_in, 3).contiguous()
    symmetric = int(idx.reshape(-1)[0].item()) in sym_list                              # loss_refiner.py:41
    if torch.is_grad_enabled() and (pred_r.requires_grad or pred_t.requires_grad):
        from autoposeestimation_amd.autograd import RefineDisFn
        dis, pred = RefineDisFn.apply(pred_r.float().reshape(1, 4), pred_t.float().reshape(1, 3), mdl, tgt, symmetric)
        dis = dis.reshape(1)
    else:
        dis, _, pred = E.adds_dis(r, t, None, mdl, tgt, symmetric, want_pred=True, want_std=False)
    qt = torch.cat([r.view(4), t.view(3)]).contiguous()
    new_points = E.recentre_qt(pts, qt).view(1, n_in, 3)
    new_target = E.recentre_qt(tgt, qt).view(1, num_point_mesh, 3)
    return dis, new_points, new_target, pred


class Loss_refine(torch.nn.Module):
    def __init__(self, num_points_mesh, sym_list):
        super().__init__()
        self.num_pt_mesh = num_points_mesh
        self.sym_list = sym_list

    def forward(self, pred_r, pred_t, target, model_points, idx, points):
        return loss_calculation(pred_r, pred_t, target, model_points, idx, points, self.num_pt_mesh, self.sym_list)
