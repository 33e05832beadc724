"""Label-path parity against fixtures made by RUNNING THE REFERENCE (tools/gen_golden.py, tools/gen_golden_label.py):
`points2pixel` / `pixels2points` / `get_my_source_center` (pc_reconstruction/open3d_utils.py:215-243,273-292), the `get_surface`
back-projection loop (:171-192) and the relabel trust checks of `create_pose_data` (label_generator/create_labels.py:96-214).
CPU half: the host functions and the oracle restatements; GPU half: the kernels."""
import numpy as np
import pytest
import torch

from conftest import golden
from autoposeestimation_amd import synthetic as S

H, W = 480, 640


class _PC:
    def __init__(self, p):
        self.points = p

    def get_center(self):
        return np.mean(self.points, axis=0)


def test_pc_utils_host_functions_match_reference_golden():
    from autoposeestimation_amd.pc_reconstruction import open3d_utils as U
    g = golden("pc_utils")
    intr = dict(S.REALSENSE_META["intr"])
    assert np.array_equal(np.array(U.points2pixel(g["points"], intr)), g["pixels"])               # int() truncation, (row, col) order
    got = np.array(U.pixels2points(g["pix"], g["depth"].astype(np.float64), intr))
    assert got.shape == g["pix_points"].shape and np.array_equal(got, g["pix_points"])          # zero-depth pixels dropped, float64 op order
    assert np.array_equal(U.get_my_source_center(_PC(g["points"])), g["centre"])
    # the drawing helper stamps exactly the pixels points2pixel names
    img = np.zeros((480, 640, 3))
    out = U.pointcloud2image(img, g["points"][:20], 3, intr, color=[10, 20, 30])
    for r, c in g["pixels"][:20]:
        if 1 <= r < 479 and 1 <= c < 639:
            assert out[r, c, 0] > 0


def _surface_case(g, ci):
    h, w = (int(v) for v in g["gs%d_shape" % ci])
    label = np.unpackbits(g["gs%d_label" % ci])[:h * w].reshape(h, w).astype(np.uint8) * 255
    fx, fy, ppx, ppy = (float(v) for v in g["gs%d_intr" % ci])
    return label, g["gs%d_depth" % ci], {"fx": fx, "fy": fy, "ppx": ppx, "ppy": ppy}, g["gs%d_robot2cam" % ci], g["gs%d_points" % ci]


def test_oracle_surface_points_match_reference_get_surface_loop():
    """pins oracle/pointcloud_oracle.surface_points to the points the reference's own get_surface loop builds"""
    from oracle import pointcloud_oracle as PO
    g = golden("label_path")
    for ci in range(int(g["gs_n"])):
        label, depth, intr, r2c, want = _surface_case(g, ci)
        got = PO.surface_points(label, depth.astype(np.float64), intr, r2c)
        assert got.shape == want.shape
        # same pixels in the same (row-major) order; the reference multiplies a 4x4 by hand per pixel (np.dot), the oracle spells the
        # same float64 sum -- equal to the last bit or, at worst, an ulp of the mm-scale coordinates
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


def _relabel_inputs(g):
    n = int(g["rl_n"])
    p1 = np.stack([g["rl%d_p1" % i].astype(np.float32) for i in range(n)])
    bs = np.stack([np.unpackbits(g["rl%d_bs" % i])[:H * W].reshape(H, W) * 255 for i in range(n)]).astype(np.uint8)
    depth = np.stack([g["rl%d_depth" % i] for i in range(n)])
    want = [np.unpackbits(g["rl%d_label" % i])[:H * W].reshape(H, W).astype(np.uint8) * 255 if g["rl_saved"][i] else None for i in range(n)]
    r2c = np.tile(np.dot(g["rl_robot2end"], g["rl_handeye"]), (n, 1, 1))
    return n, p1, bs, depth, want, r2c


def test_trust_check_restatement_matches_reference_run():
    """the numpy restatement of create_labels.py:127-196 that the GPU tests use as their checker reproduces the reference's own
    decisions (saved / bs copied / dropped) and label images on the scripted frames"""
    from oracle import densefusion_oracle as O
    g = golden("label_path")
    n, p1, bs, depth, want, r2c = _relabel_inputs(g)
    dist = np.linalg.norm(g["rl_reference_point"] - r2c[0][:3, 3])
    stats = {"bs": 0, "nodepth": 0, "centre": 0}
    for i in range(n):
        pred = torch.softmax(torch.from_numpy(np.stack([1 - p1[i], p1[i]])), 0)            # F.softmax of the (already soft-maxed) predict
        lab = O.seg_postprocess(pred, min_pixels=-1).get(1, np.zeros((H, W), np.uint8))
        d = depth[i].astype(np.float64)
        d[d > dist + 150] = 0
        d[d < dist - 150] = 0
        if len(np.unique(lab[bs[i] != 0])) <= 1:
            saved, lab = True, bs[i]
            stats["bs"] += 1
        elif len(np.unique(lab[d != 0])) <= 1:
            saved = False
            stats["nodepth"] += 1
        else:
            saved = len(np.unique(lab[30:H - 30, 50:W - 50])) > 1
            stats["centre"] += 0 if saved else 1
        assert saved == bool(g["rl_saved"][i]), i
        if saved:
            assert np.array_equal(lab, want[i]), i
    assert (stats["bs"], stats["nodepth"], stats["centre"]) == (int(g["rl_log_bs_copied"]), int(g["rl_log_no_depth"]), int(g["rl_log_not_centre"]))


@pytest.mark.gpu
def test_gpu_surface_points_match_reference_get_surface_loop():
    from autoposeestimation_amd.pc_reconstruction import pointcloud as pc
    g = golden("label_path")
    for ci in range(int(g["gs_n"])):
        label, depth, intr, r2c, want = _surface_case(g, ci)
        got = np.array(pc.surface_points(label, depth, intr, r2c).points)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.gpu
def test_gpu_relabel_frames_match_reference_run():
    """relabel_frames (segmentor output -> best component -> trust checks) on the frames the reference's create_pose_data was run on:
    same frames saved, same label images, same counters."""
    from autoposeestimation_amd.label_generator.create_labels import relabel_frames
    g = golden("label_path")
    n, p1, bs, depth, want, r2c = _relabel_inputs(g)

    class Scripted:                       # logits whose softmax is the reference run's predict() output
        classes = 2

        def logits_nhwc(self, x4):
            p = torch.from_numpy(np.stack([1 - p1, p1], -1))
            return torch.log(p).contiguous().cuda()

    labels, save, stats = relabel_frames(Scripted(), torch.zeros(n, H, W, 3, dtype=torch.uint8).cuda(), torch.from_numpy(depth).cuda(), r2c,
                                         g["rl_reference_point"], 0, torch.from_numpy(bs).cuda())
    labels = labels.cpu().numpy()
    assert [bool(v) for v in save] == [bool(v) for v in g["rl_saved"]]
    for i in range(n):
        if save[i]:
            assert np.array_equal(labels[i], want[i]), i
    assert (stats["bs_copied"], stats["no_depth_overlap"], stats["not_in_center"]) == \
        (int(g["rl_log_bs_copied"]), int(g["rl_log_no_depth"]), int(g["rl_log_not_centre"]))


def test_capture_path_fixture_is_a_camera_orbit():
    """tests/golden/viewpoints_path2.npz (the reference's capture path through its hand-eye calibration, tools/gen_golden_viewpoints.py):
    164 rigid camera poses whose optical axes meet near one point -- what bench.py --workload label renders its views from."""
    import numpy as np
    from autoposeestimation_amd import synthetic as S
    poses, focus = S.capture_path()
    assert poses.shape == (164, 4, 4) and focus.shape == (3,)
    R = poses[:, :3, :3]
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-5) and np.allclose(np.linalg.det(R), 1.0, atol=1e-5)   # (the stored calibration is orthonormal to 3e-7)
    assert np.allclose(poses[:, 3], [0, 0, 0, 1])
    to_focus = focus - poses[:, :3, 3]
    dist = np.linalg.norm(to_focus, axis=1)
    assert 300 < dist.min() and dist.max() < 1200
    cosang = np.einsum("ij,ij->i", to_focus / dist[:, None], poses[:, :3, 2])
    assert np.median(cosang) > 0.99 and cosang.min() > 0.9          # every camera looks at the turntable
    views = S.label_views(2, cloud=S.bumpy_sphere(60000, 21, centre=focus), poses=[poses[0], poses[100]])
    assert all((d != 0).sum() > 2000 for _, d, _ in views)
