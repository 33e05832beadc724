import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from autoposeestimation_amd import synthetic as S
from autoposeestimation_amd.pipeline.utils import FramePipeline
dev = torch.device("cuda", 0)
fit = [S.synthetic_frame(900 + 7 * c + k, cls=c, box=(30 + 45 * c + 20 * k, 20 + 60 * c + 90 * k), size=(126, 126)) for c in range(1, 4) for k in range(2)]
seg, est, ref, *_ = bench.build_models(dev, fit)
for m in (seg, est, ref): m.set_precision("bf16x3")
pipe = FramePipeline(seg, est, ref, bench.CLASSES, num_points=1000)
f0 = bench.select_frames(64, 0, pipe, dev); g0 = bench.make_frames(64, 0)
print("rank 0 selection == make_frames:", all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(f0, g0)))
for r in (1, 6, 7):
    fr = bench.select_frames(128, r, pipe, dev)
    rgb = torch.from_numpy(np.stack([f[0] for f in fr[:64]])).to(dev); dep = torch.from_numpy(np.stack([f[1] for f in fr[:64]])).to(dev)
    o = pipe.run(rgb, dep, S.REALSENSE_META, seed=3)["objects"]
    print("rank", r, len(fr), "objects in first 64:", len(o), set((x[3]-x[2], x[5]-x[4]) for x in o))
