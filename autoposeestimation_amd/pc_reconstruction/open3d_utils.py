"""Drop-in surface of pc_reconstruction/open3d_utils.py.  Round 1 provides the numpy-only helpers that the live path
touches (`points2pixel` :233-243, `pixels2points` :215-231, `pointcloud2image` :246-270, `get_my_source_center` :273-292)
vectorised on the host -- they are visualisation / bookkeeping, not the hot path.  The open3d-backed functions
(get_surface, preprocess_point_cloud, icp_regression, align_point_clouds) are the ICP row of SURVEY.md 8a (a15-a17) and
land with the HIP ICP kernels."""
import numpy as np


def pixels2points(pixels, depth, intr):
    pixels = np.asarray(pixels).reshape(-1, 2)
    p2 = depth[pixels[:, 0], pixels[:, 1]]
    keep = p2 != 0
    py, px, p2 = pixels[keep, 0], pixels[keep, 1], p2[keep]
    p0 = (px - intr.get("ppx")) * p2 / intr.get("fx")
    p1 = (py - intr.get("ppy")) * p2 / intr.get("fy")
    return [[a, b, c] for a, b, c in zip(p0, p1, p2)]


def points2pixel(points, intr):
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    p1 = (x / (z / intr.get("fx")) + intr.get("ppx")).astype(np.int64)     # int() truncates toward zero
    p0 = (y / (z / intr.get("fy")) + intr.get("ppy")).astype(np.int64)
    return [[int(a), int(b)] for a, b in zip(p0, p1)]


def pointcloud2image(image, point_cloud, point_size, intr, color=None):
    step = int((point_size - 1) / 2)
    mark = np.zeros((point_size, point_size, 3))
    if not color:
        mark[:, :, 0] = 255
    else:
        mark[:, :, :] = np.asarray(color)[:3]
    points = point_cloud if isinstance(point_cloud, np.ndarray) else np.array(point_cloud.points)
    for r, c in points2pixel(points, intr):
        patch = image[r - step:r + step + 1, c - step:c + step + 1, :]
        if r - step < 0 or c - step < 0 or patch.shape != mark.shape:
            continue                     # the reference swallows the broadcast error for stamps leaving the image
        image[r - step:r + step + 1, c - step:c + step + 1, :] = mark * 0.3 + patch * 0.7
    return image


def get_my_source_center(source):
    pts = np.array(source.points)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    return lo + (hi - lo) / 2
