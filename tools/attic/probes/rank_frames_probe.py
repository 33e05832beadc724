"""Do the synthetic frames of every rank (bench.make_frames(64, rank)) give one detection per frame like rank 0's?  (weak-scaling sanity:
extra detections on some rank would mean extra pose-stage work there)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from autoposeestimation_amd import synthetic as S  # noqa: E402
from autoposeestimation_amd.pipeline.utils import FramePipeline  # noqa: E402

dev = torch.device("cuda", 0)
fit = [S.synthetic_frame(900 + 7 * c + k, cls=c, box=(30 + 45 * c + 20 * k, 20 + 60 * c + 90 * k), size=(126, 126)) for c in range(1, 4) for k in range(2)]
seg, est, ref, *_ = bench.build_models(dev, fit)
for m in (seg, est, ref):
    m.set_precision("bf16x3")
pipe = FramePipeline(seg, est, ref, bench.CLASSES, num_points=1000, pose_stream=False)
for rank in range(8):
    frames = bench.make_frames(64, rank)
    rgb = torch.from_numpy(np.stack([f[0] for f in frames])).to(dev)
    depth = torch.from_numpy(np.stack([f[1] for f in frames])).to(dev)
    out = pipe.run(rgb, depth, S.REALSENSE_META, seed=0)
    sizes = {}
    for o in out["objects"]:
        k = "%dx%d" % (o[3] - o[2], o[5] - o[4])
        sizes[k] = sizes.get(k, 0) + 1
    per_frame = np.bincount([o[0] for o in out["objects"]], minlength=64)
    print("rank %d: %d objects, frames with != 1 detection: %d, crop buckets %s" % (rank, len(out["objects"]), int((per_frame != 1).sum()), sizes))
