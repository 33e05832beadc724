"""Timing of ape_grid_knn_mean_dist_f64 on one view's surface for several cell sizes, with the share of points whose k-th neighbour
lies outside one cell (those take the serial fallback).  python tools/mb_knn_grid.py"""
import sys
import time

import numpy as np
import torch
from scipy.spatial import cKDTree

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autoposeestimation_amd import _lib, synthetic as S  # noqa: E402
from autoposeestimation_amd.pc_reconstruction import pointcloud as PC  # noqa: E402

label, depth, cam = S.label_views(1, seed=0)[0]
pc = PC.surface_points(label, depth, S.LABEL_INTR, cam).voxel_down_sample(2.0)
pc, _ = pc.remove_radius_outlier(20, 5.0)
pts = np.asarray(pc.points)
n, k = len(pts), 20
dk = cKDTree(pts).query(pts, k)[0][:, -1]
print("points", n, "k-th NN distance: median %.2f max %.2f" % (np.median(dk), dk.max()))
mean = torch.empty(n, dtype=torch.float64, device="cuda")
for cell in (5.0, 7.57, 10.0, 15.0):
    g = pc._grid(cell)
    ijk = np.floor((pts - pts.min(0) + cell) / cell).astype(np.int64)
    occ = len(np.unique(ijk, axis=0))
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(_lib.lib().ape_grid_knn_mean_dist_f64(*PC.PointCloud._gargs(g), k, _lib.dptr(mean), None), "knn")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("cell %5.2f: %7.1f us   outside-one-cell share %.4f   points per occupied cell %.1f" % (cell, dt * 1e6, (dk >= cell).mean(), n / occ))
