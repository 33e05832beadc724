"""One batch-1 frame's launch list from a rocprofv3 kernel trace of tools/mb_latency.py (the LAST frame of the batch-1 loop is not identifiable
there, so this script runs its own loop: 12 frames of batch 1, the trace's last frame is listed).
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/lat_step_list.py
    python tools/prof_step_list.py <dir>/*/*_kernel_trace.csv"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from autoposeestimation_amd import synthetic as S  # noqa: E402
from autoposeestimation_amd.pipeline.utils import FramePipeline  # noqa: E402
dev = torch.device("cuda", 0)
frames = bench.make_frames(1, 0)
fit = [S.synthetic_frame(900 + 7 * c + k, cls=c, box=(30 + 45 * c + 20 * k, 20 + 60 * c + 90 * k), size=(126, 126)) for c in range(1, 4) for k in range(2)]
seg, est, ref, *_ = bench.build_models(dev, fit)
for m in (seg, est, ref):
    m.set_precision("bf16x3")
pipe = FramePipeline(seg, est, ref, bench.CLASSES, num_points=1000, refine_mode="live_compat", pose_stream=False, low_latency=os.environ.get("APE_LOW_LATENCY", "1") != "0")
rgb = torch.from_numpy(frames[0][0][None]).to(dev)
depth = torch.from_numpy(frames[0][1][None]).to(dev)
for i in range(12):
    out = pipe.run(rgb, depth, S.REALSENSE_META, seed=i)
    out["pose"].cpu()
torch.cuda.synchronize()
