"""Drop-in for DenseFusion/lib/loss.py (`Loss`, reference :76-85 over loss_calculation :12-73): the DenseFusion loss value,
the ADD / ADD-S distance of the most confident point and the re-centred clouds for refinement, computed by
ape_adds_dis_f32 / ape_adds_select_f32 / ape_recentre_qt_f32 (no N x M x 3 repeats, no k-NN distance matrix).
When a prediction requires grad (training, train.py:218-225) the returned `loss` sits on the tape of
autoposeestimation_amd/autograd.py and `loss.backward()` runs ape_adds_grad_f32; dis / new_points / new_target are detached
as in the reference (:73)."""
import torch

from autoposeestimation_amd import engine as E


def _f(t, shape):
    return t.detach().float().reshape(shape).contiguous()


def loss_calculation(pred_r, pred_t, pred_c, target, model_points, idx, points, w, refine, num_point_mesh, sym_list):
    bs, num_p, _ = pred_c.size()
    if bs != 1:
        raise ValueError("reference semantics are batch-1 (loss.py:55-59 index batch element 0)")
    if not pred_r.is_cuda:
        raise RuntimeError("Loss.forward needs device tensors: the MI355X path has no CPU fallback")
    pts, tgt, mdl = _f(points, (num_p, 3)), _f(target, (num_point_mesh, 3)), _f(model_points, (num_point_mesh, 3))
    symmetric = (not refine) and (int(idx.reshape(-1)[0].item()) in sym_list)           # loss.py:40-41
    if torch.is_grad_enabled() and (pred_r.requires_grad or pred_t.requires_grad or pred_c.requires_grad):
        from autoposeestimation_amd.autograd import PoseLossFn
        loss, out9, pred = PoseLossFn.apply(pred_r.float().reshape(num_p, 4), pred_t.float().reshape(num_p, 3),
                                            pred_c.float().reshape(num_p), pts, mdl, tgt, symmetric, float(w))
        qt = out9[2:9].contiguous()
        return (loss, out9[1], E.recentre_qt(pts, qt).view(1, num_p, 3), E.recentre_qt(tgt, qt).view(1, num_point_mesh, 3), pred)
    r, t, c = _f(pred_r, (num_p, 4)), _f(pred_t, (num_p, 3)), _f(pred_c, (num_p,))
    dis, std, pred = E.adds_dis(r, t, pts, mdl, tgt, symmetric, want_pred=True)
    out9, _ = E.adds_select(dis, std, c, r, t, pts, w)
    qt = out9[2:9].contiguous()
    new_points = E.recentre_qt(pts, qt).view(1, num_p, 3)
    new_target = E.recentre_qt(tgt, qt).view(1, num_point_mesh, 3)
    return out9[0], out9[1], new_points, new_target, pred


class Loss(torch.nn.Module):
    def __init__(self, num_points_mesh, sym_list):
        super().__init__()
        self.num_pt_mesh = num_points_mesh
        self.sym_list = sym_list

    def forward(self, pred_r, pred_t, pred_c, target, model_points, idx, points, w, refine):
        return loss_calculation(pred_r, pred_t, pred_c, target, model_points, idx, points, w, refine, self.num_pt_mesh,
                                self.sym_list)
