"""Camera poses of the reference's own capture path, as data (build container only):

    python tools/gen_golden_viewpoints.py        ->  tests/golden/viewpoints_path2.npz

`robot_controller/robot_path/viewpointsPath2.json` holds the 169 end-effector poses the robot visits per (object, direction) run, 164 of
them capture points (`via_points == 0`, data_generation/getData.py:175); the end-effector pose is x, y, z (mm) + an axis-angle vector
(a, b, c) (getData.py:189-197), and the camera pose is `robot2endEff_tf . hand_eye_calibration` (create_labels.py:104-106) with the
calibration of `hand_eye_calibration/data/handEye_tf.json`.  Stored: robot2cam[164, 4, 4] (float64) and the point closest to all optical
axes (where the turntable object sits).  BASELINE configs[4] ("200 synthetic views ... on the viewpointsPath2.json pattern") renders
its views from these poses (bench.py --workload label, synthetic.label_views(poses=...))."""
import json
import os

import numpy as np

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def axangle(r):
    r = np.asarray(r, dtype=np.float64)
    a = np.linalg.norm(r)
    if a < 1e-12:
        return np.eye(3)
    k = r / a
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


path = json.load(open(os.path.join(REF, "robot_controller/robot_path/viewpointsPath2.json")))
hand_eye = np.array(json.load(open(os.path.join(REF, "hand_eye_calibration/data/handEye_tf.json")))["tf"], dtype=np.float64).reshape(4, 4)
cams = []
for p, via in zip(path["cart_pose"], path["via_points"]):
    if int(via) != 0:
        continue
    T = np.eye(4)
    T[:3, :3] = axangle([p["a"], p["b"], p["c"]])
    T[:3, 3] = [p["x"], p["y"], p["z"]]
    cams.append(T @ hand_eye)
cams = np.array(cams)
A, b = np.zeros((3, 3)), np.zeros(3)
for o, z in zip(cams[:, :3, 3], cams[:, :3, 2]):
    P = np.eye(3) - np.outer(z, z)
    A += P
    b += P @ o
focus = np.linalg.solve(A, b)
out = os.path.join(REPO, "tests", "golden", "viewpoints_path2.npz")
np.savez_compressed(out, robot2cam=cams, focus=focus)
d = np.linalg.norm(cams[:, :3, 3] - focus, axis=1)
print("%d capture poses, focus %s, camera distance %.0f .. %.0f mm (median %.0f); wrote %s (%.1f KB)"
      % (len(cams), focus.round(1), d.min(), d.max(), np.median(d), out, os.path.getsize(out) / 1024))
