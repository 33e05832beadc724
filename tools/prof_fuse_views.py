"""cProfile of fuse_views on synthetic views (where the label path's host time goes).  python tools/prof_fuse_views.py [n_views]"""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autoposeestimation_amd import synthetic as S  # noqa: E402
from autoposeestimation_amd.pc_reconstruction import open3d_utils as U  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
views = S.label_views(n, seed=0)
U.fuse_views(views[:2], S.LABEL_INTR, voxel_size=2, threshold=10)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
U.fuse_views(views, S.LABEL_INTR, voxel_size=2, threshold=10, icp_point2point=True, icp_point2plane=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
