#!/usr/bin/env python3
"""bench.py -- RGB-D frames/s through seg + DenseFusion + 2 refine iterations on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the whole live path (FramePipeline.run: u8 RGB + u16 depth resident in HBM -> PSPNet-resnet18
segmentation at 480x640 -> masks / CCL / bbox -> 160x160 crop -> PoseNet(N=1000) -> 2x PoseRefineNet -> float64 pose)
over one batch of synthetic frames (BASELINE config 3: batch=64 640x480 frames per GPU).  Frames shard across ranks
(weak scaling, no data-path collective); each step ends with the single RCCL all_gather of the poses (config 4).
By default the loop is software-pipelined (pose stage of step i on a second HIP stream beside the segmentation of step i+1; every
step still does all of its work between the fences): about +4 % frames/s over `--no-overlap`, whose per-kernel event timings are
the kernels' own (with the overlap they include what the co-running small pose launches cost them).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The `sweep` leg replays one captured graph per
# crop-size bucket on its own stream (FramePipeline(pose_graphs=True)): with four queues two chains run side by side, with one queue
# per stream all of them do (pose stage of the ragged batch 32 -> 17 ms).  Must be set before the first HIP call; no effect on `value`
# (same-box A/B of 2 / 4 / 8 / 16 queues: 31.85 ms per step each).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "40")

import numpy as np  # noqa: E402
import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from autoposeestimation_amd import engine as E  # noqa: E402
from autoposeestimation_amd import synthetic as S  # noqa: E402

CLASSES = ["obj%02d" % i for i in range(12)]
H, W, N_POINTS = 480, 640, 1000
# algorithmic work per frame: PSPNet-r18 segmentation 275.15 GF + crop encoder 22.95 + PointNet/heads 7.96 + 2 x 1.48
GFLOP_PER_FRAME = 309.0
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA, dense"
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md "HBM3E peak BW" (spec; 6.29 TB/s measured by a float4 copy)


def pmc_traffic(kernel, shape=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json, made by
    tools/pmc_summary.py with the guide's FETCH_SIZE x2 gfx950 correction); None when no PMC run covers this kernel.
    `shape`: the launches of that one layer shape only (the passes record every dispatch of the last step in launch order, and
    tools/pmc_summary.py --shapes labels them from bench.py's own launch list of the same step), so a kernel that serves layers of very
    different sizes is compared like with like.  PMC collection cannot run inside bench.py itself (it needs rocprofv3 around the process)."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            k = json.load(open(path))["kernels"].get(kernel.replace(" ", ""))
        except (OSError, ValueError, KeyError):
            continue
        if k:
            if shape is None:
                return k["hbm_bytes_per_launch"]
            row = (k.get("shapes") or {}).get(shape)
            return None if row is None else row["hbm_bytes_per_launch"]
    return None


def ranks_seen(rank, device, dist):
    """[(rank, device index, device uuid, device name)] of every rank, all-gathered over the job's process group: evidence in the line
    that an N-GPU run really had N distinct GPUs behind its ranks (one process per GPU)."""
    try:
        props = torch.cuda.get_device_properties(device)
        mine = [int(rank), int(device.index), str(getattr(props, "uuid", "")), props.name]
    except Exception as e:      # noqa: BLE001 -- evidence only, never a reason to fail the bench
        mine = [int(rank), int(device.index), "unknown (%s)" % type(e).__name__, "unknown"]
    if dist is None or not dist.is_initialized():
        return [mine]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, mine)
    return parts


def make_frame(rank, i):
    fid = rank * 100003 + i
    rng = np.random.default_rng(fid)
    box = (int(rng.integers(20, 480 - 150)), int(rng.integers(20, 640 - 150)))
    return S.synthetic_frame(fid, cls=1 + (i % 3), box=box, size=(126, 126))


def make_frames(batch, rank):
    return [make_frame(rank, i) for i in range(batch)]


def select_frames(batch, rank, pipe, device):
    """`batch` frames of rank `rank`, each with exactly ONE detection whose crop is 160x160 -- the workload BASELINE configs[2] describes
    ("one painted object per frame => one 160x160 crop").  The frozen random segmentor leaves a stray 40x40 blob or clips a border row
    (crop 120x160) in about 1 of 100 of the random frames; such frames are skipped during set-up (untimed) so that every rank carries
    the same work (weak scaling).  Rank 0's first 64 candidates all qualify, so its batch is make_frames(64, 0)."""
    frames, nxt, skipped = [], 0, 0
    while len(frames) < batch:
        cand = [make_frame(rank, nxt + k) for k in range(64)]
        nxt += 64
        rgb = torch.from_numpy(np.stack([f[0] for f in cand])).to(device)
        depth = torch.from_numpy(np.stack([f[1] for f in cand])).to(device)
        objects = pipe.run(rgb, depth, S.REALSENSE_META, seed=0)["objects"]
        per_frame = {}
        for o in objects:
            per_frame.setdefault(o[0], []).append((o[3] - o[2], o[5] - o[4]))
        for k, f in enumerate(cand):
            if len(frames) < batch:
                if per_frame.get(k) == [(160, 160)]:
                    frames.append(f)
                else:
                    skipped += 1
        if nxt > 64 * 50:
            raise RuntimeError("synthetic frame selection does not converge")
    return frames, skipped


def build_models(device, frames_for_fit):
    from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet
    from autoposeestimation_amd.segmentation.utils import get_model
    seg = get_model("PsPNet", {"encoder_name": "resnet18", "encoder_weights": None, "activation": "softmax",
                               "in_channels": 3, "classes": 13})
    seg_sd = S.pspnet_state_dict("resnet18", seed=5, stem_gain=1.0)
    seg.load_state_dict(seg_sd)
    seg = seg.to(device).eval()
    # "train" the final 1x1 conv by least squares on frozen random features so the painted objects are segmented
    feats, labels = [], []
    rects = torch.zeros(1, 3, dtype=torch.int32, device=device)
    for rgb, _, label in frames_for_fit:
        x4 = E.preprocess_u8(torch.from_numpy(rgb[None]).to(device), rects, H, W, True)
        f = seg.plan().features(x4)[0].reshape(-1, 64)
        flat = label.reshape(-1)
        fg = np.nonzero(flat)[0]
        bg = np.random.default_rng(0).choice(np.nonzero(flat == 0)[0], size=6 * len(fg), replace=False)
        sel = torch.from_numpy(np.concatenate([fg, bg]))
        feats.append(f[sel.to(device)])
        labels.append(torch.from_numpy(flat.astype(np.int64))[sel])
    w, b = S.fit_final_layer(torch.cat(feats), torch.cat(labels), 13)
    fw, fb = seg_sd["final.0.weight"].clone(), seg_sd["final.0.bias"].clone()
    fw[:13, :, 0, 0], fb[:13] = w, b
    seg_sd["final.0.weight"], seg_sd["final.0.bias"] = fw, fb
    seg.load_state_dict(seg_sd)
    seg = seg.to(device).eval()
    est = PoseNet(N_POINTS, 12)
    est_sd = S.posenet_state_dict(12, 0)
    est.load_state_dict(est_sd)
    ref = PoseRefineNet(N_POINTS, 12)
    ref_sd = S.refiner_state_dict(12, 0)
    ref.load_state_dict(ref_sd)
    return seg, est.to(device).eval(), ref.to(device).eval(), seg_sd, est_sd, ref_sd


def _cpu_info():
    model, phys = "unknown", None
    try:
        cores = set()
        phys_id = core_id = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys_id = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core_id = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys_id is not None and core_id is not None:
                    cores.add((phys_id, core_id))
                phys_id = core_id = None
        phys = len(cores) or None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))       # the CPUs this process may run on (the GPU box gives a share of the host)
    except (AttributeError, OSError):
        usable = os.cpu_count()
    return model, phys, usable


def _cpu_quota():
    """CPUs the cgroup lets this process use at once (cpu.max quota / period), None = unlimited / unknown"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except (OSError, ValueError):
        return None


def cpu_baseline(frames, seg_sd, est_sd, ref_sd, gpu_choose, n_frames=16, n_frames_all=64):
    """The oracle (CPU restatement of pipeline/utils.py:410-641, pinned by the reference goldens) timed on the host cores on a bounded
    sample of the SAME frames.  The reference path is batch-1 Python + torch CPU and does not scale with intra-op threads (one frame
    on 128 threads is SLOWER than on one), so the fair multi-core figure runs independent frames side by side: W single-threaded
    workers (W = the CPUs of a one-GPU box share, at most 16), `n_frames` frames.  Also reported: one frame on one thread.
    SURVEY.md 8d asks for "all physical cores" beside the single-thread figure: a second leg runs the REMAINING frames of the sample
    (up to `n_frames_all` in total) with one single-threaded worker per physical core the process may use (capped by the frames left)
    -> `all_cores_value`.  Reported baseline only -- never part of `value`.  `gpu_choose(frame, class, nz, n)` injects the GPU run's
    point selection (the reference draws it from an unseeded shuffle), so the oracle results of BOTH legs double as the checker of
    the `parity` block."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import densefusion_oracle as O
    model, phys, usable = _cpu_info()
    workers = max(1, min(16, usable or 1, phys or 16))
    n_frames = min(n_frames, len(frames))

    def run(i):
        rgb, depth, _ = frames[i]
        return O.full_prediction(rgb, depth, S.REALSENSE_META, seg_sd, est_sd, ref_sd, CLASSES,
                                 choose_fn=lambda name, nz, n: gpu_choose(i, name, nz, n))

    old_threads = torch.get_num_threads()
    torch.set_num_threads(1)                                 # intra-op: one thread per frame; the frames run in parallel
    run(0)                                                   # warm (oneDNN primitives)
    t1 = time.time()
    run(0)
    dt1 = time.time() - t1
    t = time.time()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        results = dict(zip(range(n_frames), ex.map(run, range(n_frames))))
    dt = time.time() - t
    # second leg: the rest of the sample on every physical core this process may use
    n_all = max(0, min(n_frames_all, len(frames)) - n_frames)
    workers_all = max(1, min(phys or usable or 1, usable or 1, n_all)) if n_all else 0
    all_cores = None
    if n_all and workers_all > workers:
        t = time.time()
        with ThreadPoolExecutor(max_workers=workers_all) as ex:
            results.update(zip(range(n_frames, n_frames + n_all), ex.map(run, range(n_frames, n_frames + n_all))))
        dt_all = time.time() - t
        quota = _cpu_quota()
        # `cores` = the CPUs those workers can actually occupy: a cgroup quota below the worker count makes them time-slice
        all_cores = {"value": round(n_all / dt_all, 4), "cores": int(min(workers_all, quota)) if quota else workers_all, "workers": workers_all,
                     "frames": n_all}
    torch.set_num_threads(old_threads)
    found = sum(len(r) for r in results.values())
    return {"value": round(n_frames / dt, 4), "unit": "frames/s", "cores": workers, "kind": "port",
            "single_thread_value": round(1.0 / dt1, 4), "all_cores_value": None if all_cores is None else all_cores["value"],
            "all_cores": all_cores, "cpu_model": model, "physical_cores": phys, "usable_cpus": usable, "cgroup_cpu_quota": _cpu_quota(),
            "sample": "%d of the benchmark's 640x480 frames through oracle/densefusion_oracle.full_prediction (torch CPU fp32), %d "
                      "single-threaded workers side by side; single_thread_value = 1 frame on 1 thread; all_cores_value = %s "
                      "(%d objects found over both legs)"
                      % (n_frames, workers, "not run" if all_cores is None else "%d further frames of the batch on %d single-threaded "
                         "workers (one per physical core the process may use; `all_cores.cores` = min(workers, cgroup CPU quota), what they can occupy)"
                         % (n_all, workers_all), found)}, results


def parity_block(out, oracle_results, frames, seg_sd):
    """GPU (the timed path's last step) vs the oracle on the frames cpu_baseline ran: max |dq| (sign-aligned quaternion), max |dt| (m),
    differing mask pixels, and how many of those are NOT arg-max near-ties in the oracle's own probabilities (must be 0)."""
    from oracle import densefusion_oracle as O
    import torch.nn.functional as F
    objmap = out["objmap"].cpu().numpy()
    pose = out["pose"].cpu().numpy()
    max_dq = max_dt = adds_delta = adds_gpu_sum = adds_cpu_sum = 0.0
    diff_px = diff_outside = missing = n_adds = 0
    dev = out["pose"].device
    for fi, want in oracle_results.items():
        mine = {CLASSES[o[1] - 1]: k for k, o in enumerate(out["objects"]) if o[0] == fi}
        missing += len(set(want) ^ set(mine))
        for name, w in want.items():
            if name not in mine:
                continue
            k = mine[name]
            cls = out["objects"][k][1]
            differs = (objmap[fi] == cls) != (w["mask"] == 255)
            if differs.any():
                with torch.no_grad():
                    pr = F.softmax(O.segmentor_predict(seg_sd, O.seg_input(frames[fi][0]), len(CLASSES) + 1), dim=1)[0]
                ys, xs = np.nonzero(differs)
                top = torch.topk(pr[:, ys, xs], 2, dim=0).values
                diff_px += len(ys)
                diff_outside += int(((top[0] - top[1]) >= 1e-4).sum())
            q = pose[k, :4] if np.dot(pose[k, :4], w["rotation"]) >= 0 else -pose[k, :4]
            max_dq = max(max_dq, float(np.abs(q - w["rotation"]).max()))
            max_dt = max(max_dt, float(np.abs(pose[k, 4:] - w["position"]).max()))
            # ADD-S of both poses against ONE seeded ground-truth placement of the class's 1000-point model cloud (0.1 m cube,
            # SURVEY.md 8d; DenseFusion/tools/eval_linemod.py:118-130): the build's through its own HIP k-NN / ADD-S kernel, the
            # restatement's through the oracle's CPU arithmetic.  Ground truth = the oracle pose turned by 3 degrees about a seeded
            # axis and moved by 3 mm, so both distances are millimetres, not zero.
            rng = np.random.default_rng([77, int(fi), int(cls)])
            axis = rng.standard_normal(3)
            axis /= np.linalg.norm(axis)
            half = np.deg2rad(3.0) / 2
            dq = np.concatenate([[np.cos(half)], np.sin(half) * axis])
            R_gt = O.quaternion_matrix(np.asarray(w["rotation"], np.float64))[:3, :3] @ O.quaternion_matrix(dq)[:3, :3]
            step = rng.standard_normal(3)
            t_gt = np.asarray(w["position"], np.float64) + 0.003 * step / np.linalg.norm(step)
            model = S.model_cloud(cls)
            target = (model.astype(np.float64) @ R_gt.T + t_gt).astype(np.float32)
            g_dis, _, _ = E.adds_dis(torch.from_numpy(pose[k:k + 1, :4].astype(np.float32)).to(dev), torch.from_numpy(pose[k:k + 1, 4:].astype(np.float32)).to(dev),
                                     None, torch.from_numpy(model).to(dev), torch.from_numpy(target).to(dev), True, want_std=False)
            pred = (model @ O.quaternion_matrix(np.asarray(w["rotation"], np.float64))[:3, :3].T + np.asarray(w["position"], np.float64)).astype(np.float32)
            tt, pp = torch.from_numpy(target).t().contiguous(), torch.from_numpy(pred).t().contiguous()
            inds = O.knn1(tt.unsqueeze(0), pp.unsqueeze(0)).view(-1) - 1
            c_dis = float(torch.mean(torch.norm(pp.t() - tt.t()[inds], dim=1)))
            g = float(g_dis[0])
            adds_delta, adds_gpu_sum, adds_cpu_sum, n_adds = max(adds_delta, abs(g - c_dis)), adds_gpu_sum + g, adds_cpu_sum + c_dis, n_adds + 1
    return {"frames": len(oracle_results), "max_dq": max_dq, "max_dt": max_dt, "tolerance": 1e-4,
            "adds_delta_m": adds_delta, "adds_mean_gpu_m": adds_gpu_sum / max(n_adds, 1), "adds_mean_oracle_m": adds_cpu_sum / max(n_adds, 1),
            "adds_objects": n_adds, "mask_diff_px": diff_px,
            "mask_diff_px_outside_tie_band": diff_outside, "objects_not_matched": missing,
            "note": "mask pixels may differ only where the oracle's own top-2 class probabilities are closer than 1e-4 (arg-max near-ties); "
                    "adds_delta_m = max |ADD-S(GPU pose, HIP k-NN/ADD-S kernel) - ADD-S(oracle pose, CPU restatement)| against one seeded "
                    "ground-truth placement of the 1000-point 0.1 m-cube model cloud (eval_linemod.py:118-130), bar 1e-4 m"}


def mixed_sweep(args, rank, world, device, dist, seg, est, ref, seg_sd, est_sd, ref_sd, fence, steps, with_parity):
    """`bench.py --mixed`: --batch frames per rank of synthetic.mixed_frame (1-3 objects, five painted sizes) through FramePipeline --
    segmentation, components, then ONE pose-stage pass per distinct crop size of the batch (FramePipeline.poses' buckets) -- and one
    all_gather of a [frames, len(CLASSES), 8] result block per step.  Returns the `sweep` object (rank 0; None elsewhere)."""
    from autoposeestimation_amd.pipeline.utils import FramePipeline
    from autoposeestimation_amd.sharding import gather_results
    n = args.batch
    mframes = [S.mixed_frame(rank * 100003 + i) for i in range(n)]
    rgb = torch.from_numpy(np.stack([f[0] for f in mframes])).to(device)
    depth = torch.from_numpy(np.stack([f[1] for f in mframes])).to(device)
    # With the buckets' graphs fanned over their own streams the step runs as segmentation, THEN all pose chains side by side (45 ms); beside
    # the next batch's segmentation -- persistent workgroups that own every CU -- the chains only advance at its kernel boundaries (48 ms).
    # The eager launches (--no-pose-graphs) keep the software-pipelined loop: there the host is the bound.
    overlap = bool(args.overlap and args.no_pose_graphs)
    pipe = FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=overlap,
                         pose_graphs=not args.no_pose_graphs)

    def tail(out):
        with torch.cuda.stream(out.get("stream") or torch.cuda.current_stream()):
            block = torch.zeros(n, len(CLASSES), 8, dtype=torch.float32, device=device)    # (one detection per class at most: FramePipeline keeps the best component of each)
            if out["objects"]:
                o = np.asarray(out["objects"], dtype=np.int64)
                first = np.searchsorted(o[:, 0], o[:, 0], side="left")          # objects are frame-major: slot = position within the frame
                slot = np.arange(len(o)) - first
                t = torch.from_numpy(np.stack([o[:, 0], slot, o[:, 1]])).pin_memory().to(device, non_blocking=True)
                block[t[0], t[1], 0] = t[2].float()
                block[t[0], t[1], 1:] = out["pose"].float()
            out["gathered"] = gather_results(block, dist)
        return out

    def run(count, seed0):
        out = None
        n_obj = 0
        if not overlap:
            for i in range(count):
                out = tail(pipe.run(rgb, depth, S.REALSENSE_META, seed=seed0 + i))
                n_obj += len(out["objects"])
            return out, n_obj
        h = pipe.begin(rgb)
        for i in range(count):
            h_next = pipe.begin(rgb) if i + 1 < count else None
            out = tail(pipe.finish(h, rgb, depth, S.REALSENSE_META, seed=seed0 + i))
            n_obj += len(out["objects"])
            h = h_next
        return out, n_obj

    run(3, 0)       # (a crop-size bucket runs eagerly on its first occurrence, is captured on its second and replayed from the third on)
    fence()
    pipe.host_poses_s = 0.0
    t0 = time.perf_counter()
    out, n_obj = run(steps, 3)
    fence()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt, float(n_obj)], dtype=torch.float64, device=device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, n_obj = float(tmax[0]), int(round(float(t[1])))
    if rank != 0:
        return None
    buckets, painted = {}, {}
    for o in out["objects"]:
        k = "%dx%d" % (o[3] - o[2], o[5] - o[4])
        buckets[k] = buckets.get(k, 0) + 1
    for f in mframes:
        for _, _, (rh, rw) in f[3]:
            painted["%dx%d" % (rh, rw)] = painted.get("%dx%d" % (rh, rw), 0) + 1
    sweep = {"value": round(n * world * steps / dt, 2), "unit": "frames/s", "objects_per_s": round(n_obj / dt, 2),
             "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "frames_per_gpu_per_step": n,
             "objects_per_step_rank0": len(out["objects"]), "objects_painted_rank0": sum(len(f[3]) for f in mframes),
             "crop_buckets_last_step": dict(sorted(buckets.items())), "painted_sizes_rank0": dict(sorted(painted.items())),
             "overlap": overlap, "pose_graphs": not args.no_pose_graphs, "bucket_streams": len(pipe._bucket_streams),
             "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
             "host_pose_enqueue_ms_per_step": round(pipe.host_poses_s / steps * 1e3, 3),
             "note": "synthetic.mixed_frame: 1-3 painted objects of distinct classes per 640x480 frame, sizes drawn from {70x70, 110x150, 150x150, "
                     "230x230, 310x390} (crops 80x80 .. 320x400, SURVEY.md 8d); one pose-stage pass per distinct crop size of the batch, each "
                     "replayed as ONE captured HIP graph (FramePipeline(pose_graphs=True): the ~90 launches of a bucket are captured on its second "
                     "occurrence; --no-pose-graphs enqueues them one by one) on a stream of its own, all buckets side by side behind the batch's "
                     "segmentation; run after the timed region, never part of `value`"}
    if with_parity and not args.no_cpu_baseline and world == 1:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import densefusion_oracle as O
        if out.get("stream") is not None:
            out["stream"].synchronize()
        choose_h = out["choose"].cpu().numpy()
        by_frame = {(o[0], CLASSES[o[1] - 1]): k for k, o in enumerate(out["objects"])}

        def gpu_choose(fi, name, nz, cnt):
            k = by_frame.get((fi, name))
            if k is None:
                return nz[(np.arange(cnt) * len(nz)) // cnt] if len(nz) > cnt else np.pad(nz, (0, cnt - len(nz)), "wrap")
            return choose_h[k]

        sample = list(range(min(8, n)))
        old_threads = torch.get_num_threads()
        torch.set_num_threads(1)
        with ThreadPoolExecutor(max_workers=min(8, len(sample))) as ex:
            res = dict(zip(sample, ex.map(lambda i: O.full_prediction(mframes[i][0], mframes[i][1], S.REALSENSE_META, seg_sd, est_sd, ref_sd, CLASSES,
                                                                       choose_fn=lambda name, nz, cnt: gpu_choose(i, name, nz, cnt)), sample)))
        torch.set_num_threads(old_threads)
        sweep["parity"] = parity_block(out, res, mframes, seg_sd)
        sweep["parity"]["objects_checked"] = sum(len(r) for r in res.values())
    return sweep


def latency_leg(args, device, seg, est, ref, frame, runs):
    """`bench.py --latency`: the batch-1 live loop (main.py:517-553 -> pipeline/utils.py full_prediction): ONE frame already in HBM, one painted
    object; per run the host clock from FramePipeline.run(...) to the pose in host memory (segmentation, components, the one D2H of the
    detections, crop, PoseNet, 2 x refiner, pose D2H: ~140 dependent launches on one stream).  p50 / p99 / min over `runs` runs
    behind 10 warm-up runs.  Rank 0 only (a latency, not a throughput: it does not aggregate over ranks)."""
    from autoposeestimation_amd.pipeline.utils import FramePipeline
    rgb = torch.from_numpy(frame[0][None]).to(device)
    depth = torch.from_numpy(frame[1][None]).to(device)
    pick = lambda ts, q: float(ts[min(len(ts) - 1, int(np.ceil(q * len(ts))) - 1)])  # noqa: E731

    def measure(pipe):
        for i in range(10):
            out = pipe.run(rgb, depth, S.REALSENSE_META, seed=i)
            out["pose"].cpu()
        torch.cuda.synchronize()
        ts = []
        for i in range(runs):
            t0 = time.perf_counter()
            out = pipe.run(rgb, depth, S.REALSENSE_META, seed=i)
            pose = out["pose"].cpu()
            ts.append(time.perf_counter() - t0)
        return np.sort(np.asarray(ts)) * 1e3, out, pose

    # the pose stage's ~60 launches can only be enqueued once the detections are on the host; with pose_graphs the crop-size bucket is ONE
    # replayed HIP graph (the same kernels in the same order: bit-identical poses, checked on the last run).  Measured round 6: p50 3.97 ms
    # against 3.95 eager -- the host cost is not what the frame waits for; both are reported
    # low_latency=True: what full_prediction (the reference's live API, one frame per call) runs -- the crop's small-M layers split K
    te, out_e, pose_e = measure(FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=False, low_latency=True))
    ts, out, pose_g = measure(FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=False, pose_graphs=True,
                                            low_latency=True))
    tn, out_n, pose_n = measure(FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=False))
    return {"p50_ms": round(pick(te, 0.50), 3), "p99_ms": round(pick(te, 0.99), 3), "min_ms": round(float(te[0]), 3), "mean_ms": round(float(te.mean()), 3),
            "pose_graph_p50_ms": round(pick(ts, 0.50), 3), "pose_graph_min_ms": round(float(ts[0]), 3), "graph_equals_eager_bitwise": bool(torch.equal(pose_e, pose_g)),
            "unsplit_p50_ms": round(pick(tn, 0.50), 3), "split_vs_unsplit_max_abs_pose_diff": float((pose_e - pose_n).abs().max()),
            "runs": int(len(te)), "objects": len(out_e["objects"]), "frames": 1,
            "note": "one resident 640x480 frame, one object, batch 1, one stream, eager launches, FramePipeline(low_latency=True) as full_prediction uses it "
                    "(the crop's small-M layers split K; unsplit_p50_ms: the default batched pipeline's form at batch 1); host wall clock from "
                    "FramePipeline.run() to the pose on the host (includes the detections' D2H sync in the middle and the pose D2H at the end); pose_graph_*: the same with the pose "
                    "stage as ONE replayed HIP graph (FramePipeline(pose_graphs=True)) -- no faster: the frame is its ~140 dependent small kernels, "
                    "not their host cost; after the timed region, never part of `value`"}


def kernel_peak(label):
    """(bound, peak, unit) for a profiled kernel label"""
    if "upconv_gather" in label:
        return "hbm", PEAK_HBM_GBS, "GB/s"
    if "conv_f32" in label:
        return "mfma", PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
    # split-bf16: three bf16 MFMAs per algorithmic product (upconv_fused.hip exists in that precision only)
    split = "_s32_kernel" in label or "<3," in label or "upconv_fused_kernel" in label
    return "mfma", PEAK_BF16_MFMA_TFLOPS / (3.0 if split else 1.0), "TFLOP/s"


# ---- --workload label: BASELINE configs[4] -------------------------------------------------------------------------------------
LABEL_KW = dict(voxel_size=2, threshold=10, min_friends=20, min_dist=5, nb_neighbors=20, icp_point2point=True, icp_point2plane=True)
# algorithmic HBM bytes per (source point, ICP evaluation), float64 xyz: transform 24 r + 24 w; correspondence search 24 r (query) +
# 24 r (the nearest target, the grid walk's other candidates are not counted) + 12 w (index, distance); sums 24 r (source) + 12 r
# (index, distance) + 24 r (target point) [+ 24 r target normal for point-to-plane]
ICP_BYTES_PER_PAIR = {0: 168, 1: 192}


def label_cpu_baseline(views_np, n_views=6):
    """oracle/pointcloud_oracle (numpy + scipy cKDTree restatement of get_surface + the sequential p2p / p2plane fusion,
    pc_reconstruction/open3d_utils.py:63-213, create_pointcloud.py:288-312) on the first `n_views` views of chain 0, one thread."""
    from oracle import pointcloud_oracle as PO
    model, phys, usable = _cpu_info()
    t0 = time.time()
    acc = None
    for label, depth, cam in views_np[:n_views]:
        p = PO.voxel_down_sample(PO.surface_points(label, depth, S.LABEL_INTR, cam), 2.0)
        p = p[PO.radius_outlier_mask(p, 20, 5.0)]
        mask, _ = PO.statistical_outlier_mask(p, 20, float(np.std(PO.mahalanobis(p))))
        p = p[mask]
        if acc is None:
            acc = p
            continue
        tg, sr = PO.voxel_down_sample(acc, 2.0), PO.voxel_down_sample(p, 2.0)
        nrm = PO.estimate_normals(tg, 4.0, 30)
        T, _, _ = PO.registration_icp(sr, tg, 10.0, np.eye(4), False, None, 1e-2, 1e-2, 100)
        T, _, _ = PO.registration_icp(sr, tg, 10.0, T, True, nrm, 1e-2, 1e-2, 100)
        acc = PO.voxel_down_sample(np.concatenate([p @ T[:3, :3].T + T[:3, 3], acc]), 2.0)
    dt = time.time() - t0
    return {"value": round(n_views / dt, 3), "unit": "views/s", "cores": 1, "kind": "port", "cpu_model": model, "physical_cores": phys,
            "usable_cpus": usable, "sample": "the first %d views of chain 0 through oracle/pointcloud_oracle (numpy + scipy cKDTree), one thread; "
                                             "the chain is sequential, so it has no multi-core form short of running chains side by side" % n_views}, acc


def label_main(args, rank, world, device, dist, embedded=False):
    """configs[4]: pose-label generation.  A step = `--views` synthetic 640x480 views in chains of 25 (one chain = one (object,
    direction) sequence of create_pointcloud.py:276-312): get_surface (back-projection, voxel / radius / statistical filters) for every
    view, spread over the ranks; one padded all-gather; then the sequential p2p + point-to-plane ICP fusion of each chain on its
    owner rank (chain i -> rank i % world).  Total work is fixed as ranks are added (strong scaling).  Views are resident in HBM."""
    from autoposeestimation_amd.pc_reconstruction import open3d_utils as U
    from autoposeestimation_amd.pc_reconstruction import pointcloud as PC
    per_chain = 25
    n_chains = max(1, args.views // per_chain)
    # the views: rendered from the reference's own capture path (164 camera poses around the turntable, 322..1168 mm from the object;
    # tests/golden/viewpoints_path2.npz) -- chain c takes 25 consecutive poses and sees the object turned by c x 45 degrees about z,
    # like the rotation directories of one object (create_pointcloud.py:227-312); without the fixture, cameras on a cap around it
    path = S.capture_path()
    if path is not None:
        poses, focus = path
        base = S.bumpy_sphere(300000, 21, centre=np.zeros(3))
        chains_np = []
        for c in range(n_chains):
            a = c * np.pi / 4
            Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
            chains_np.append(S.label_views(per_chain, cloud=base @ Rz.T + focus, poses=[poses[(c * per_chain + i) % len(poses)] for i in range(per_chain)]))
        view_source = "the reference capture path (viewpointsPath2.json x hand-eye calibration: 164 poses, 322-1168 mm from the object)"
    else:
        cloud = S.bumpy_sphere(300000, 21)
        chains_np = [S.label_views(per_chain, seed=c, cloud=cloud) for c in range(n_chains)]
        view_source = "cameras on a cap 500 mm around the object"
    chains = [[(torch.from_numpy(l).to(device), torch.from_numpy(d).to(device), cam) for (l, d, cam) in ch] for ch in chains_np]
    n_views = n_chains * per_chain

    def step():
        return U.fuse_chains(chains, S.LABEL_INTR, dist=dist, **LABEL_KW)

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    fence()
    PC.ICP_STATS = {"registrations": 0, "evaluations": 0, "pairs": 0, "events": []}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    stats, PC.ICP_STATS = PC.ICP_STATS, None
    icp_ms = sum(a.elapsed_time(b) for a, b in stats["events"])
    counts = torch.tensor([stats["registrations"], stats["evaluations"], stats["pairs"], stats.get("kind0", 0), stats.get("kind1", 0),
                           sum(len(c) for c, _ in res.values())], dtype=torch.float64, device=device)
    tmax = torch.tensor([dt, icp_ms], dtype=torch.float64, device=device)
    if dist:
        dist.all_reduce(counts)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt, icp_ms_max = float(tmax[0]), float(tmax[1])
    if rank == 0:
        regs, evals, pairs, p0, p1, fused_pts = [float(x) for x in counts]
        icp_bytes = p0 * ICP_BYTES_PER_PAIR[0] + p1 * ICP_BYTES_PER_PAIR[1]
        ach = icp_bytes / world / (icp_ms_max * 1e-3) / 1e9 if icp_ms_max else 0.0
        line = {"metric": "pose-label views/sec (get_surface + sequential p2p + point-to-plane ICP fusion), 640x480 views",
                "value": round(n_views * args.steps / dt, 2), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": "configs[4]: pose-label generation, %d synthetic 640x480 views of a 300k-point surface in %d chains of %d "
                                       "(voxel 2 mm, radius + statistical outlier filters, p2p then point-to-plane ICP at 10 mm per view), views "
                                       "rendered from %s, resident in HBM" % (n_views, n_chains, per_chain, view_source),
                           "parallelism": "per-view get_surface over %d ranks, 1 padded all_gather/step, chain i fused on rank i %% %d" % (world, world),
                           "fused_points_last_step": int(fused_pts)},
                "ranks_seen": args.ranks_seen, "distinct_gpus": len({tuple(r[1:3]) for r in args.ranks_seen}),
                "icp": {"registrations_per_s": round(regs / dt, 1), "evaluations_per_registration": round(evals / max(regs, 1), 2),
                        "point_pairs_per_s": round(pairs / dt, 0)},
                "roofline": {"kernel": "ICP registration (icp_step + icp_transform + nn1 + p2p/p2plane sums per evaluation)", "bound": "hbm",
                             "achieved": round(ach, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 5),
                             "traffic": None, "share_of_step_time": round(icp_ms_max * 1e-3 / dt, 3),
                             "note": "algorithmic bytes = point pairs x %d B (p2p) / %d B (p2plane), see ICP_BYTES_PER_PAIR; time = HIP events around "
                                     "every registration on its stream (max over ranks).  The clouds hold ~10^4 points (240 KB): every kernel "
                                     "of the chain is a few microseconds of work behind a dependent launch, so this stage is latency-bound by "
                                     "construction and sits far below the HBM roofline; chains run side by side to use the GPU(s)"
                                     % (ICP_BYTES_PER_PAIR[0], ICP_BYTES_PER_PAIR[1])}}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"], acc = label_cpu_baseline(chains_np[0])
            # parity on the sample: the GPU fusion of the same first views against the oracle's
            g, _ = U.fuse_views(chains[0][:6], S.LABEL_INTR, **LABEL_KW)
            gp = np.asarray(g.points)
            from scipy.spatial import cKDTree
            d = cKDTree(acc).query(gp)[0]
            line["parity"] = {"checker": "oracle/pointcloud_oracle on the cpu_baseline sample (6 views of chain 0)", "gpu_points": len(gp),
                              "oracle_points": len(acc), "max_nn_distance_mm": round(float(d.max()), 6)}
        if embedded:
            return line
        print(json.dumps(line))
    if embedded:
        return None
    if dist:
        dist.barrier()
        dist.destroy_process_group()


PEAK_F32_LANE_OPS = 256 * 4 * 32 * 2.4e9 / 1e12     # T lane-operations/s: 256 CUs x 4 SIMD-32 at 2.4 GHz, one fp32 VALU operation per lane and cycle
KNN_LANE_OPS_PER_PAIR = 11                           # 3 sub, 3 mul, 2 add, 1 compare, 2 selects (csrc/knn.hip; the reference's arithmetic, no FMA)


def pose_main(args, rank, world, device, dist, embedded=False):
    """BASELINE configs[1]: PoseNet + 2 x PoseRefineNet on --crops 160x160 crops per GPU and step (N = 1000 points), then ADD-S of every
    refined pose against its ground-truth cloud through the hand-written k-NN / ADD-S kernel (1000 x 1000 pair evaluations per crop,
    eval_linemod.py:118-130).  A step = FramePipeline.poses (choose / back-projection / crop normalisation / PoseNet / pose selection /
    two refiner passes / float64 composition) + ape_adds_dis_batched_f32 + the single all_gather of (class, q, t, ADD-S) per crop.
    Inputs (u8 frames, u16 depth, the object map and its detections) are resident in HBM; crops shard over the ranks (weak scaling).
    Secondary object `knn_training_size`: ape_knn_f32 at the training loss's size, 10^6 queries x 1000 refs (loss.py:38-47)."""
    from autoposeestimation_amd.pipeline.utils import FramePipeline
    from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet
    from autoposeestimation_amd.DenseFusion.lib.knn import KNearestNeighbor
    from autoposeestimation_amd.sharding import gather_results
    n = args.crops
    est_sd, ref_sd = S.posenet_state_dict(12, 0), S.refiner_state_dict(12, 0)
    est, ref = PoseNet(N_POINTS, 12), PoseRefineNet(N_POINTS, 12)
    est.load_state_dict(est_sd)
    ref.load_state_dict(ref_sd)
    est, ref = est.to(device).eval().set_precision(args.pose_precision), ref.to(device).eval().set_precision(args.pose_precision)
    frames = make_frames(n, rank)
    rgb = torch.from_numpy(np.stack([f[0] for f in frames])).to(device)
    depth = torch.from_numpy(np.stack([f[1] for f in frames])).to(device)
    label = torch.from_numpy(np.stack([f[2] for f in frames]).astype(np.uint8)).to(device)
    pipe = FramePipeline(None, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat")
    # detections of the painted objects from the label maps themselves (the product's own component / bbox kernels; no segmentor here)
    objmap, det = E.seg_components(label, torch.ones(label.shape, dtype=torch.float32, device=device), len(CLASSES) + 1, 100)
    det_h = det.cpu().numpy()
    fb, fc = np.nonzero(det_h[:, 1:, 0])
    objects = [(int(b), int(c) + 1, *map(int, det_h[b, c + 1, 1:5])) for b, c in zip(fb, fc)]
    assert len(objects) == n and all((o[3] - o[2], o[5] - o[4]) == (160, 160) for o in objects), "one 160x160 crop per frame expected"
    cls_t = torch.tensor([o[1] for o in objects], dtype=torch.float32, device=device)
    model = torch.from_numpy(np.stack([S.model_cloud(o[1]) for o in objects])).to(device)           # [n, 1000, 3] (0.1 m cubes, per class)

    def step(seed, target, timed=None, gather=True):
        pose, n_cand, choose = pipe.poses(rgb, depth, objmap, objects, S.REALSENSE_META, seed=seed)
        q, t = pose[:, :4].float().contiguous(), pose[:, 4:].float().contiguous()
        if timed is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dis = E.adds_dis_batched(q, t, model, target, True)
        if timed is not None:
            e1.record()
            timed.append((e0, e1))
        block = torch.cat([cls_t[:, None], pose.float(), dis[:, None]], 1)[:, None, :]            # [n, 1, 9]
        return pose, choose, dis, gather_results(block, dist if gather else None)      # (gather=False: rank 0's profiled extra step)

    # ground truth of every crop: the first run's pose turned by 3 degrees about a seeded axis and moved by 3 mm (ADD-S in millimetres)
    pose0, _, _, _ = step(0, model)
    p0 = pose0.cpu().numpy()
    targets = []
    for k, o in enumerate(objects):
        rng = np.random.default_rng([77, rank, k])
        axis = rng.standard_normal(3)
        axis /= np.linalg.norm(axis)
        half = np.deg2rad(3.0) / 2
        R_gt = _quat_matrix(p0[k, :4]) @ _quat_matrix(np.concatenate([[np.cos(half)], np.sin(half) * axis]))
        stp = rng.standard_normal(3)
        t_gt = p0[k, 4:] + 0.003 * stp / np.linalg.norm(stp)
        targets.append((S.model_cloud(o[1]).astype(np.float64) @ R_gt.T + t_gt).astype(np.float32))
    target = torch.from_numpy(np.stack(targets)).to(device)

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, target)
    fence()
    timed = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        pose, choose, dis, gathered = step(args.warmup + i, target, timed)
    fence()
    dt = time.perf_counter() - t0
    adds_ms = sum(a.elapsed_time(b) for a, b in timed) / max(len(timed), 1)
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0])
    if rank == 0:
        m = model.shape[1]
        pairs = n * m * m
        ach = pairs * KNN_LANE_OPS_PER_PAIR / (adds_ms * 1e-3) / 1e12
        # after the timed region: every conv launch of one step timed (the MFMA side of the workload), and the k-NN kernel at training size
        prof = E.LaunchProfile()
        E.PROFILE = prof
        step(args.warmup + args.steps, target, gather=False)
        torch.cuda.synchronize()
        E.PROFILE = None
        top = sorted(prof.summary().items(), key=lambda kv: -kv[1]["ms"])[:3]
        mfma = []
        for lab, d in top:
            bound, peak, unit = kernel_peak(lab)
            a_ = (d["bytes"] / (d["ms"] * 1e-3) / 1e9) if bound == "hbm" else (d["flop"] / (d["ms"] * 1e-3) / 1e12)
            mfma.append({"kernel": lab, "bound": bound, "launches": d["launches"], "ms_per_step": round(d["ms"], 3), "achieved": round(a_, 2),
                         "peak": round(peak, 1), "unit": unit, "frac": round(a_ / peak, 4)})
        knn = KNearestNeighbor(1)
        kref, kqry = torch.randn(1, 3, 1000, device=device), torch.randn(1, 3, 1_000_000, device=device)
        knn(kref, kqry)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            knn(kref, kqry)
        e1.record()
        torch.cuda.synchronize()
        knn_ms = e0.elapsed_time(e1) / 10
        knn_ach = 1e9 * KNN_LANE_OPS_PER_PAIR / (knn_ms * 1e-3) / 1e12
        line = {"metric": "crops/sec (PoseNet + 2 x PoseRefineNet + ADD-S via the HIP k-NN / ADD-S kernel), 160x160 crops, N=1000",
                "value": round(n * world * args.steps / dt, 2), "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.pose_precision, "data": "synthetic",
                "config": {"workload": "configs[1]: PoseNet + PoseRefineNet (2 iterations) on %d crops of 160x160 per GPU and step, N = 1000 points, "
                                       "ADD-S of every refined pose against a 1000-point ground-truth cloud" % n,
                           "inputs": "u8 frames, u16 depth, object map and detections resident in HBM, the same crops every step",
                           "parallelism": "dp%d (crops shard over the ranks, one all_gather of [crops, 1, 9] per step)" % world},
                "ranks_seen": args.ranks_seen, "distinct_gpus": len({tuple(r[1:3]) for r in args.ranks_seen}),
                "roofline": {"kernel": "adds_points_kernel + adds_mean_kernel (symmetric: the k-NN kernel's pair evaluation, M x M per object)", "bound": "valu",
                             "achieved": round(ach, 3), "peak": round(PEAK_F32_LANE_OPS, 1), "unit": "T lane-op/s", "frac": round(ach / PEAK_F32_LANE_OPS, 4),
                             "traffic": None, "avg_launch_us": round(adds_ms * 1e3, 1), "pairs_per_launch": pairs,
                             "share_of_step_time": round(adds_ms * 1e-3 / (dt / args.steps), 4),
                             "note": "algorithmic work = objects x M x M pair evaluations x %d fp32 lane-operations (3 sub, 3 mul, 2 add, compare, 2 "
                                     "selects: the reference's arithmetic, which the bit-exact index contract fixes); peak = 256 CUs x 4 SIMD-32 x 2.4 GHz; "
                                     "time = HIP events around the two launches, every step of the timed region, on their stream.  %d objects x "
                                     "10^6 pairs are ~10 us of vector work for the whole chip: the pair of launches is latency-bound -- "
                                     "`knn_training_size` is the same arithmetic at a size that fills the chip"
                                     % (KNN_LANE_OPS_PER_PAIR, n),
                             "kernels": mfma},
                "knn_training_size": {"kernel": "knn1_d3_q<2, 8>", "refs": 1000, "queries": 1000000, "ms": round(knn_ms, 4),
                                      "pairs_per_s": round(1e9 / (knn_ms * 1e-3), 0), "bound": "valu", "achieved": round(knn_ach, 3),
                                      "peak": round(PEAK_F32_LANE_OPS, 1), "unit": "T lane-op/s", "frac": round(knn_ach / PEAK_F32_LANE_OPS, 4)},
                "adds_mean_m": round(float(dis.mean()), 6)}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"], line["parity"] = pose_cpu_baseline(frames, objects, est_sd, ref_sd, choose.cpu().numpy(), pose.cpu().numpy(),
                                                                     dis.cpu().numpy(), targets, device)
        if embedded:
            return line
        print(json.dumps(line))
    if embedded:
        return None
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def _quat_matrix(q):
    """3x3 rotation of a wxyz quaternion (transformations.py:1254-1278), float64"""
    from autoposeestimation_amd.DenseFusion.lib.transformations import quaternion_matrix
    return quaternion_matrix(np.asarray(q, np.float64))[:3, :3]


def pose_cpu_baseline(frames, objects, est_sd, ref_sd, choose, pose_gpu, dis_gpu, targets, device, n_crops=4):
    """The oracle on the first `n_crops` crops of the same workload, one thread: PoseNet + 2 refiner forwards + composition
    (oracle/densefusion_oracle.py, pinned by the reference goldens) + ADD-S with the nearest neighbours from oracle/liboracle_knn.so (the
    plain-C restatement of knn_cpu.cpp, pinned by the reference object).  Doubles as the checker: pose within 1e-4, the HIP k-NN's
    indices bit for bit the oracle's, ADD-S within 1e-4 m."""
    import ctypes
    from oracle import densefusion_oracle as O
    from autoposeestimation_amd.DenseFusion.lib.knn import KNearestNeighbor
    model_name, phys, usable = _cpu_info()
    olib = ctypes.CDLL(os.path.join(REPO, "oracle", "liboracle_knn.so"))
    olib.oracle_knn.restype = ctypes.c_int
    olib.oracle_knn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long] * 5
    old = torch.get_num_threads()
    torch.set_num_threads(1)
    n_crops = min(n_crops, len(objects))
    max_dq = max_dt = adds_delta = 0.0
    knn_equal = True
    t0 = time.time()
    for k in range(n_crops):
        fi, cls, rmin, rmax, cmin, cmax = objects[k]
        rgb, depth, _ = frames[fi]
        ch = choose[k].astype(np.int64)
        pts = torch.from_numpy(O.backproject(depth, ch, rmin, rmax, cmin, cmax, S.REALSENSE_META)).unsqueeze(0)
        img = O.crop_image(rgb, rmin, rmax, cmin, cmax)
        cht = torch.from_numpy(ch).view(1, 1, -1)
        idx = torch.tensor([[cls - 1]])
        with torch.no_grad():
            pr, pt, pc, emb = O.posenet_forward(est_sd, img, pts, cht, idx, 12)
            newp = O.get_new_points(pr, pt, pc, pts)
            _, my_r, my_t = O.estimator_prediction(pr, pt, pc, N_POINTS, 1, pts)
            for _ in range(2):
                rr, rt = O.refiner_forward(ref_sd, newp, emb, idx, 12)
            _, want_r, want_t = O.refined_prediction(rr, rt, my_r, my_t)
        want_r, want_t = np.asarray(want_r, np.float64), np.asarray(want_t, np.float64)
        # ADD-S of the oracle's pose, neighbours from the C restatement of the reference's k-NN
        mc = S.model_cloud(cls)
        pred = (mc.astype(np.float64) @ O.quaternion_matrix(want_r)[:3, :3].T + want_t).astype(np.float32)
        tgt = targets[k]
        ref_a, qry_a = np.ascontiguousarray(tgt.T[None]), np.ascontiguousarray(pred.T[None])
        inds = np.zeros((1, 1, pred.shape[0]), np.int64)
        assert olib.oracle_knn(ref_a.ctypes.data, qry_a.ctypes.data, inds.ctypes.data, 1, 3, tgt.shape[0], pred.shape[0], 1) == 1
        c_dis = float(np.mean(np.linalg.norm(pred - tgt[inds[0, 0] - 1], axis=1)))
        if k == 0:
            dt_first = time.time() - t0
        # the checker half (not part of the CPU timing's meaning, but cheap): HIP k-NN on the same clouds, pose and ADD-S deltas
        got = KNearestNeighbor(1)(torch.from_numpy(ref_a).to(device), torch.from_numpy(qry_a).to(device)).cpu().numpy()
        knn_equal = knn_equal and bool(np.array_equal(got, inds))
        q = pose_gpu[k, :4] if np.dot(pose_gpu[k, :4], want_r) >= 0 else -pose_gpu[k, :4]
        max_dq = max(max_dq, float(np.abs(q - want_r).max()))
        max_dt = max(max_dt, float(np.abs(pose_gpu[k, 4:] - want_t).max()))
        adds_delta = max(adds_delta, abs(float(dis_gpu[k]) - c_dis))
    dt = time.time() - t0
    torch.set_num_threads(old)
    base = {"value": round(n_crops / dt, 4), "unit": "crops/s", "cores": 1, "kind": "port", "cpu_model": model_name, "physical_cores": phys,
            "sample": "%d of the benchmark's crops through oracle/densefusion_oracle (PoseNet + 2 refiner forwards + composition, torch CPU fp32, one "
                      "thread) + ADD-S with oracle/liboracle_knn.so's neighbours (includes the checker's HIP k-NN call per crop)" % n_crops}
    parity = {"crops": n_crops, "max_dq": max_dq, "max_dt": max_dt, "tolerance": 1e-4, "adds_delta_m": adds_delta,
              "knn_indices_bit_exact": knn_equal, "checker": "oracle/densefusion_oracle + oracle/liboracle_knn.so on the cpu_baseline sample"}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed steps (default 40: about 1.5 s of GPU work at 64 frames per step)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--frames", type=int, default=0,
                    help="BASELINE configs[3]: a step = this many frames in TOTAL, sharded over the ranks (strong scaling; each rank "
                         "walks its share in sub-batches of --batch) with one all_gather of all poses per step.  0 = configs[2], "
                         "--batch frames per GPU per step (weak scaling)")
    ap.add_argument("--workload", default="frames", choices=["frames", "label", "pose"],
                    help="frames: the live path (BASELINE metric).  label: configs[4], pose-label generation -- multi-view depth -> "
                         "point-cloud fusion + point-to-plane ICP over --views synthetic views, sharded across the ranks.  pose: configs[1], "
                         "PoseNet + 2 x PoseRefineNet + ADD-S (HIP k-NN / ADD-S kernel) on --crops 160x160 crops per GPU")
    ap.add_argument("--views", type=int, default=200, help="--workload label: views per step (one (object, direction) chain)")
    ap.add_argument("--crops", type=int, default=32, help="--workload pose (BASELINE configs[1]): 160x160 crops per GPU and step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfiltered-frames", action="store_true",
                    help="time the first --batch random frames of each rank as they come (about 1 in 100 carries a stray blob or a clipped "
                         "120x160 crop) instead of selecting frames with exactly one 160x160 detection (configs[2], the default)")
    ap.add_argument("--baseline-frames", type=int, default=64,
                    help="frames of the batch the CPU baseline / parity leg runs (all-cores leg; the 16-worker leg uses the first 16 of them)")
    ap.add_argument("--overlap", dest="overlap", action="store_true", default=True,
                    help="(default) pose stage of step i on a second HIP stream beside the segmentation of step i+1 (software-pipelined "
                         "loop, every step still does all of its work inside the fences): +4 %% frames/s; the per-kernel event timings "
                         "of the roofline leg then include whatever the co-running pose kernels cost them")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", help="one stream, segmentation and pose stage back to back")
    ap.add_argument("--dump-launches", default="", help="write [[kernel label, layer shape], ...] of ONE step's profiled launches in launch order "
                    "(for tools/pmc_summary.py --shapes: per-shape HBM traffic from the rocprofv3 PMC passes; use with --no-overlap)")
    ap.add_argument("--no-staged", action="store_true", help="skip the secondary `staged` leg (the same loop fed from pinned host memory through a copy stream)")
    ap.add_argument("--no-modes", action="store_true", help="skip the secondary `modes` leg (two steps of the exact-fp32 operand mode after the timed region)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the compact secondary `sweep` leg (4 steps of the --mixed frames, no parity block)")
    ap.add_argument("--no-latency", action="store_true", help="skip the compact secondary `latency` leg (50 runs of the batch-1 live loop)")
    ap.add_argument("--no-step-check", action="store_true", help="skip the re-run of every timed step on one stream after the timed region "
                    "(`parity.steps_bitwise_equal`; the profile passes of tools/make_profiles.sh skip it: their traces count steps)")
    ap.add_argument("--no-pose-leg", action="store_true", help="skip the compact secondary `pose` leg (BASELINE configs[1]: 10 steps of --workload pose)")
    ap.add_argument("--no-label-leg", action="store_true", help="skip the compact secondary `label` leg (BASELINE configs[4]: 1 step of --workload label, 200 views)")
    ap.add_argument("--mixed", action="store_true",
                    help="the full `sweep` leg (--mixed-steps steps + its own parity block; the default line carries a compact one: 4 steps, no "
                         "parity block).  `sweep`: --batch frames with 1-3 painted objects each, sizes drawn from SURVEY.md 8d's crop "
                         "sweep {80x80, 120x160, 160x160, 240x240, 320x400}, through the same path (several crop-size buckets per step); run after the "
                         "timed region, never part of `value`")
    ap.add_argument("--mixed-steps", type=int, default=5)
    ap.add_argument("--no-pose-graphs", action="store_true",
                    help="--mixed: enqueue every crop-size bucket's ~90 pose-stage launches one by one instead of replaying one captured HIP graph per bucket")
    ap.add_argument("--latency", action="store_true",
                    help="the full `latency` leg (--latency-runs runs; the default line carries a compact one of 50 runs).  `latency`: ONE resident 640x480 frame with one object through the whole path (what the reference's live "
                         "loop does per frame, main.py:517-553), host wall clock from the call to the pose on the host, p50 / p99 over --latency-runs "
                         "runs; after the timed region, never part of `value`")
    ap.add_argument("--latency-runs", type=int, default=200)
    ap.add_argument("--seg-precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--pose-precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"])
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # One rank per GPU.  Rehearsal mode (APE_DIST_BACKEND=gloo): several ranks may share the GPUs that exist -- a single-GPU box can then
    # run the driver's N = 2 launch line through every multi-rank code path (rank-dependent frames, collectives, max-over-ranks timing);
    # its number means nothing, and RCCL itself needs one GPU per rank.
    backend = os.environ.get("APE_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if os.environ.get("APE_MAIN_STREAM_PRIORITY", "default") == "high":
        # everything the bench enqueues (segmentation, collectives) on a stream of the most urgent HIP priority; the pose stream of the
        # software-pipelined loop keeps the default one, i.e. it is the LOWER-priority stream of the two
        torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=torch.cuda.Stream.priority_range()[1]))
    dist = None
    if world > 1 or "RANK" in os.environ:       # under torch.distributed.run the RCCL path is exercised even for one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    args.ranks_seen = ranks_seen(rank, device, dist)
    if args.workload == "label":
        return label_main(args, rank, world, device, dist)
    if args.workload == "pose":
        return pose_main(args, rank, world, device, dist)

    from autoposeestimation_amd.pipeline.utils import FramePipeline
    if args.frames:
        if args.frames % world:
            raise SystemExit("--frames %d does not divide over %d ranks" % (args.frames, world))
        per_rank = args.frames // world
        args.batch = min(args.batch, per_rank)
        if per_rank % args.batch:
            raise SystemExit("the rank's share of %d frames is not a multiple of --batch %d" % (per_rank, args.batch))
    else:
        per_rank = args.batch
    n_chunks = per_rank // args.batch
    # three primary colours (classes 1..3) that a LINEAR read-out of the frozen random features separates cleanly from the
    # grey-noise background AND from each other's blurred borders (six colours left ~10 spurious >100-px detections per 64
    # frames); channels 4..12 of the 13-way segmentor stay silent
    fit_frames = [S.synthetic_frame(900 + 7 * c + k, cls=c, box=(30 + 45 * c + 20 * k, 20 + 60 * c + 90 * k), size=(126, 126))
                  for c in range(1, 4) for k in range(2)]
    seg, est, ref, seg_sd, est_sd, ref_sd = build_models(device, fit_frames)
    seg.set_precision(args.seg_precision)
    est.set_precision(args.pose_precision)
    ref.set_precision(args.pose_precision)
    if args.unfiltered_frames:
        frames, skipped = make_frames(per_rank, rank), 0
    else:
        frames, skipped = select_frames(per_rank, rank, FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat"), device)
    rgb = torch.from_numpy(np.stack([f[0] for f in frames])).to(device).split(args.batch)        # inputs resident in HBM
    depth = torch.from_numpy(np.stack([f[1] for f in frames])).to(device).split(args.batch)
    # --overlap (the default): the pose stage of step i runs on a second HIP stream beside the segmentation of step i+1 (the timed
    # region ends with torch.cuda.synchronize(), which waits for both streams; every step still does all of its work).
    pipe = FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=args.overlap)
    from autoposeestimation_amd.sharding import gather_results

    step_poses = {}
    kept = {}           # step -> the step's gathered [world * per_rank, 1, 8] block, for the timed steps (`parity.steps_bitwise_equal`)
    keep_steps = set()

    def tail(out, item):
        # one result slot per frame: the largest detection (the painted object) wins the slot.  The rank's poses of one step collect in
        # one [per_rank, 1, 8] buffer; the step's single RCCL all_gather goes out after its last sub-batch.
        step, chunk = item
        with torch.cuda.stream(out.get("stream") or torch.cuda.current_stream()):     # the pose results live on the pose stream
            if chunk == 0:
                step_poses[step] = torch.zeros(per_rank, 1, 8, dtype=torch.float32, device=device)
            poses = step_poses[step][chunk * args.batch:(chunk + 1) * args.batch]
            if out["objects"]:
                o = np.asarray(out["objects"], dtype=np.int64)
                order = np.argsort(-(o[:, 3] - o[:, 2]) * (o[:, 5] - o[:, 4]), kind="stable")
                frames_u, first = np.unique(o[order, 0], return_index=True)
                pick = order[first]
                # one small H2D from pinned memory, non-blocking: a pageable copy would park the host behind the whole pose stage
                t = torch.from_numpy(np.stack([frames_u, pick, o[pick, 1]])).pin_memory().to(device, non_blocking=True)
                poses[t[0], 0, 0] = t[2].float()
                poses[t[0], 0, 1:] = out["pose"][t[1]].float()
            if chunk == n_chunks - 1:
                out["gathered"] = gather_results(step_poses.pop(step), dist)   # the single RCCL collective of the path: (cls, q, t) per frame
                if step in keep_steps:
                    kept[step] = out["gathered"]
        return out

    def run_steps(first, count):
        """`count` steps (batches `first` .. `first + count - 1`), every one a full pass seg -> CCL -> crops -> PoseNet -> 2x refine ->
        gather over the rank's frames (in `n_chunks` sub-batches).  With the pose stream the loop is software-pipelined: the
        segmentation of the next sub-batch is enqueued BEFORE the host waits for this one's detections, so the main stream never idles
        while the host enqueues the ~100 pose launches.  Exactly `count * n_chunks` segmentation stages and as many pose stages are
        enqueued, all inside the caller's fences."""
        out = None
        items = [(i, c) for i in range(first, first + count) for c in range(n_chunks)]
        if not args.overlap:
            for (i, c) in items:
                out = tail(pipe.run(rgb[c], depth[c], S.REALSENSE_META, seed=i), (i, c))
            return out
        h = pipe.begin(rgb[items[0][1]]) if items else None
        for k, (i, c) in enumerate(items):
            h_next = pipe.begin(rgb[items[k + 1][1]]) if k + 1 < len(items) else None
            out = tail(pipe.finish(h, rgb[c], depth[c], S.REALSENSE_META, seed=i), (i, c))
            h = h_next
        return out

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The roofline leg times the FIVE heaviest kernels live, with HIP events on the stream they are launched on, over the timed
    # region.  Which five is found in the last warm-up step (every profiled launch timed); in the timed steps only their launches
    # carry the two event packets (each costs ~4 us of dispatch gap, ~0.5 ms per step when all ~120 launches have them).
    top_only = None
    out = None
    if args.warmup > 1:
        out = run_steps(0, args.warmup - 1)
    if args.warmup:
        fence()
        E.PROFILE = E.LaunchProfile()
        if args.overlap:
            # the five kernels are ranked by what they cost ALONE (one single-stream step, untimed): beside the segmentation the pose
            # stage's small launches wait for free CUs most of their event time and would outrank the kernels that do the work
            pipe_sel = FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=False)
            for c in range(n_chunks):
                out = tail(pipe_sel.run(rgb[c], depth[c], S.REALSENSE_META, seed=args.warmup - 1), (-1, c))
        else:
            out = run_steps(args.warmup - 1, 1)
    n_found = len(out["objects"]) if args.warmup else -1
    fence()
    if args.warmup:
        wsum = E.PROFILE.summary()
        if wsum:
            top_only = set(sorted(wsum, key=lambda k: -wsum[k]["ms"])[:5])
        if args.dump_launches and rank == 0:
            with open(args.dump_launches, "w") as f:
                json.dump([[r[0], r[1]] for r in E.PROFILE.records], f)
    prof = E.PROFILE = E.LaunchProfile(only=top_only)
    keep_steps.update(range(args.warmup, args.warmup + args.steps))
    t0 = time.perf_counter()
    out = run_steps(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    E.PROFILE = None
    keep_steps.clear()
    n_found = len(out["objects"])
    crop_hist = {}
    for o in out["objects"]:
        k = "%dx%d" % (o[3] - o[2], o[5] - o[4])
        crop_hist[k] = crop_hist.get(k, 0) + 1
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    # Every TIMED step's result block against the same step (same frames, same sampling seed) run again on ONE stream after the timed
    # region: the overlapped loop must give the single-stream bits (round 5 found a co-stream fault in this arrangement; the parity block
    # below only sees the last step).  Rank 0 reports `parity.steps_bitwise_equal`; any rank that differs fails the whole run.
    timed_blocks = dict(kept)
    kept.clear()
    steps_equal = None
    if not args.no_step_check:
        pipe_chk = FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=False)
        steps_equal = 0
        for i in range(args.warmup, args.warmup + args.steps):
            for c in range(n_chunks):
                o = tail(pipe_chk.run(rgb[c], depth[c], S.REALSENSE_META, seed=i), (4 * 10 ** 6 + i, c))
            steps_equal += int(i in timed_blocks and torch.equal(o["gathered"], timed_blocks[i]))
        fence()
        del pipe_chk
        if dist:
            t = torch.tensor([steps_equal], dtype=torch.int64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            steps_equal = int(t[0])
    del timed_blocks

    # With the software-pipelined loop the pose stage of the previous batch runs BESIDE the timed kernels on a second stream, so their
    # HIP-event durations in the timed region include what the co-running launches cost them (throughput goes up, every kernel takes
    # longer).  A short single-stream pass AFTER the timed region (not part of `value`) times the same five kernels alone; both sets
    # are reported (`frac` / `avg_launch_us` = timed region, `isolated_*` = alone).
    iso_summ = iso_by_shape = None
    if args.overlap and top_only:
        pipe_iso = FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=False)
        E.PROFILE = E.LaunchProfile(only=top_only)
        for i in range(3):
            for c in range(n_chunks):
                tail(pipe_iso.run(rgb[c], depth[c], S.REALSENSE_META, seed=i), (10 ** 6 + i, c))
        fence()
        iso_summ = E.PROFILE.summary()
        iso_by_shape = E.PROFILE.summary(by_shape=True)
        E.PROFILE = None

    # Secondary, AFTER the timed region (not part of `value`): the same step with exact-fp32 operands on the matrix cores -- what the
    # default mode's split-bf16 operands (three products per term, ~2^-16 operand error, inside the 1e-4 parity bar) buy.  Two
    # single-stream steps behind one warm-up step; every rank runs them (the step's gather is a collective), rank 0 reports.
    modes = None
    if not args.no_modes and (args.seg_precision, args.pose_precision) == ("bf16x3", "bf16x3"):
        for m in (seg, est, ref):
            m.set_precision("f32")
        pipe_f = FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=False)
        for c in range(n_chunks):
            tail(pipe_f.run(rgb[c], depth[c], S.REALSENSE_META, seed=0), (2 * 10 ** 6, c))
        fence()
        tf = time.perf_counter()
        f32_steps = 2
        for i in range(f32_steps):
            for c in range(n_chunks):
                tail(pipe_f.run(rgb[c], depth[c], S.REALSENSE_META, seed=i), (2 * 10 ** 6 + 1 + i, c))
        fence()
        dtf = time.perf_counter() - tf
        if dist:
            t = torch.tensor([dtf], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtf = float(t[0])
        modes = {"f32": {"value": round(per_rank * world * f32_steps / dtf, 2), "unit": "frames/s", "ms_per_step": round(dtf / f32_steps * 1e3, 3),
                         "steps": f32_steps, "note": "exact fp32 operands on the matrix cores (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak), one stream, "
                                                      "run after the timed region; never part of `value`"}}
        seg.set_precision(args.seg_precision)
        est.set_precision(args.pose_precision)
        ref.set_precision(args.pose_precision)
        del pipe_f

    # Secondary, AFTER the timed region (not part of `value`): the live loop receives its frames on the HOST (main.py:517-553,
    # pipeline/utils.py:421-427,556-560).  `staged`: every step takes a batch from pinned host memory through a COPY stream into one of two
    # device buffer pairs while the previous batch is being segmented (1.5 MB per frame: 98 MB per 64-frame step), then runs the same
    # software-pipelined loop on it.  value is the resident-input rate; this leg shows what the staging costs beside it.
    staged = None
    if not args.no_staged and not args.frames:
        copy_stream = torch.cuda.Stream()
        host = [(torch.from_numpy(np.stack([f[0] for f in frames])).pin_memory(), torch.from_numpy(np.stack([f[1] for f in frames])).pin_memory())]
        host.append((host[0][0].flip(0).contiguous().pin_memory(), host[0][1].flip(0).contiguous().pin_memory()))        # a second, different batch
        dev = [(torch.empty_like(rgb[0]), torch.empty_like(depth[0])) for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]
        freed = [torch.cuda.Event() for _ in range(2)]
        pipe_s = FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=args.overlap)

        def stage(i):
            slot = i % 2
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[slot])                      # the batch that used this buffer pair two steps ago has been consumed
                dev[slot][0].copy_(host[i % 2][0], non_blocking=True)
                dev[slot][1].copy_(host[i % 2][1], non_blocking=True)
                ready[slot].record(copy_stream)

        def run_staged(count):
            for slot in range(2):
                freed[slot].record()
            stage(0)
            torch.cuda.current_stream().wait_event(ready[0])
            h = pipe_s.begin(dev[0][0])
            for i in range(count):
                slot = i % 2
                if i + 1 < count:
                    stage(i + 1)
                    torch.cuda.current_stream().wait_event(ready[(i + 1) % 2])
                    h_next = pipe_s.begin(dev[(i + 1) % 2][0])
                else:
                    h_next = None
                o = tail(pipe_s.finish(h, dev[slot][0], dev[slot][1], S.REALSENSE_META, seed=i), (3 * 10 ** 6 + i, 0))
                # the pose stage (on its own stream when the loop is software-pipelined) is the last reader of the batch's buffers
                with torch.cuda.stream(o.get("stream") or torch.cuda.current_stream()):
                    freed[slot].record()
                h = h_next
            return o

        run_staged(2)
        fence()
        ts = time.perf_counter()
        staged_steps = max(4, args.steps)          # (as many as the timed region: the pipelined loop's fill and drain weigh the same in both)
        o = run_staged(staged_steps)
        fence()
        dts = time.perf_counter() - ts
        if dist:
            t = torch.tensor([dts], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t[0])
        staged = {"value": round(per_rank * world * staged_steps / dts, 2), "unit": "frames/s", "ms_per_step": round(dts / staged_steps * 1e3, 3),
                  "steps": staged_steps, "host_bytes_per_step": int(host[0][0].numel() + 2 * host[0][1].numel()),
                  "objects_found_last_step": len(o["objects"]),
                  "note": "every step's frames come from pinned host memory over PCIe on a copy stream, double-buffered against the previous "
                          "batch's segmentation; same loop otherwise; after the timed region, never part of `value`"}
        del pipe_s

    # Secondary, AFTER the timed region (not part of `value`): what real frames look like (pipeline/utils.py:444-470, 522-561: every detected
    # class of a frame gets its own crop, crops of one size go through the pose stage together) -- 1-3 objects per frame, five crop sizes.
    sweep = None
    if (args.mixed or not args.no_sweep) and not args.frames:
        sweep = mixed_sweep(args, rank, world, device, dist, seg, est, ref, seg_sd, est_sd, ref_sd, fence,
                            steps=args.mixed_steps if args.mixed else 4, with_parity=args.mixed)

    latency = None
    want_latency = (args.latency or not args.no_latency) and not args.frames
    if want_latency and rank == 0:
        latency = latency_leg(args, device, seg, est, ref, frames[0], runs=args.latency_runs if args.latency else 50)
    if dist and want_latency:
        dist.barrier()

    # Secondary, AFTER the timed region (not part of `value`): BASELINE configs[1] and configs[4] through the same code as --workload pose /
    # --workload label (every rank takes part: both have their own collectives), compact: 10 steps of 32 crops, 1 step of 200 views
    pose_leg = label_leg = None
    if not args.frames:
        import copy
        torch.cuda.synchronize()
        if not args.no_pose_leg:
            a = copy.copy(args)
            a.steps, a.warmup, a.crops = 10, 2, 32
            pose_leg = pose_main(a, rank, world, device, dist, embedded=True)
        if not args.no_label_leg:
            a = copy.copy(args)
            a.steps, a.warmup, a.views = 1, 1, 200
            label_leg = label_main(a, rank, world, device, dist, embedded=True)

    if rank == 0:
        summ = prof.summary()
        by_shape = prof.summary(by_shape=True)

        def entry(label, d, shapes=None):
            bound, peak, unit = kernel_peak(label)
            sec = d["ms"] * 1e-3
            ach = (d["bytes"] / sec / 1e9) if bound == "hbm" else (d["flop"] / sec / 1e12)
            traffic = pmc_traffic(label)
            alg_b = d["bytes"] / d["launches"]
            if shapes and all("traffic" in r for r in shapes):
                # like with like: the PMC bytes of each shape weighted by THIS run's launches of it (a plain per-kernel PMC average mixes
                # the big segmentation launches with the pose stage's small ones in another proportion than the timed region does)
                traffic = round(sum(r["traffic"] * r["launches"] for r in shapes) / d["launches"])
            e = {"kernel": label, "bound": bound, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": unit, "frac": round(ach / peak, 4),
                 "launches": d["launches"], "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 1),
                 "avg_launch_gflop": round(d["flop"] / d["launches"] / 1e9, 3), "avg_launch_algorithmic_mb": round(alg_b / 1e6, 1),
                 "traffic": traffic, "traffic_over_algorithmic": None if traffic is None else round(traffic / alg_b, 2),
                 "share_of_step_time": round(sec / dt, 3)}
            if iso_summ and label in iso_summ:
                di = iso_summ[label]
                sec_i = di["ms"] * 1e-3
                ach_i = (di["bytes"] / sec_i / 1e9) if bound == "hbm" else (di["flop"] / sec_i / 1e12)
                e["isolated_avg_launch_us"] = round(di["ms"] / di["launches"] * 1e3, 1)
                e["isolated_achieved"] = round(ach_i, 2)
                e["isolated_frac"] = round(ach_i / peak, 4)
            if shapes:
                e["shapes"] = shapes
            return e

        kernels = []
        # order (and the dominant kernel): by the time the kernels take ALONE when the isolated pass exists, else by timed-region time
        order_key = (lambda k: -(iso_summ[k]["ms"] if k in iso_summ else 0.0)) if iso_summ else (lambda k: -summ[k]["ms"])
        for label in sorted(summ, key=order_key):
            shapes = []
            for (lab, shape), d in sorted(by_shape.items(), key=lambda kv: -kv[1]["ms"]):
                if lab != label:
                    continue
                bound, peak, unit = kernel_peak(label)
                sec = d["ms"] * 1e-3
                ach = (d["bytes"] / sec / 1e9) if bound == "hbm" else (d["flop"] / sec / 1e12)
                row = {"shape": shape, "launches": d["launches"], "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 1),
                       "gflop": round(d["flop"] / d["launches"] / 1e9, 3), "algorithmic_mb": round(d["bytes"] / d["launches"] / 1e6, 1),
                       "achieved": round(ach, 2), "frac": round(ach / peak, 4)}
                tr = pmc_traffic(label, shape)
                if tr is not None:          # HBM bytes of THIS shape's launches (PMC) against its algorithmic bytes
                    row["traffic"] = tr
                    row["traffic_over_algorithmic"] = round(tr / (d["bytes"] / d["launches"]), 2)
                # the shape's OWN roofline: whichever of the two limits (algorithmic flop at the kernel's matrix peak, algorithmic bytes at
                # the HBM peak) takes longer bounds it -- a 1x1 layer with K = 128 is an HBM-bound launch of an MFMA kernel
                if bound == "mfma":
                    t_flop, t_hbm = d["flop"] / (peak * 1e12), d["bytes"] / (PEAK_HBM_GBS * 1e9)
                    row["hbm_frac"] = round(t_hbm / sec, 4)
                    row["roofline_bound"] = "hbm" if t_hbm > t_flop else "mfma"
                    row["roofline_frac"] = round(max(t_flop, t_hbm) / sec, 4)
                    di = iso_by_shape.get((lab, shape)) if iso_by_shape else None
                    if di:
                        sec_i = di["ms"] * 1e-3
                        row["isolated_avg_launch_us"] = round(di["ms"] / di["launches"] * 1e3, 1)
                        row["isolated_frac"] = round(di["flop"] / sec_i / 1e12 / peak, 4)
                        row["isolated_roofline_frac"] = round(max(di["flop"] / (peak * 1e12), di["bytes"] / (PEAK_HBM_GBS * 1e9)) / sec_i, 4)
                else:       # a byte-moving kernel: its one limit is the HBM peak
                    row["hbm_frac"] = row["frac"]
                    row["roofline_bound"] = "hbm"
                    row["roofline_frac"] = row["frac"]
                    di = iso_by_shape.get((lab, shape)) if iso_by_shape else None
                    if di:
                        row["isolated_avg_launch_us"] = round(di["ms"] / di["launches"] * 1e3, 1)
                        row["isolated_frac"] = row["isolated_roofline_frac"] = round(di["bytes"] / (di["ms"] * 1e-3) / 1e9 / peak, 4)
                shapes.append(row)
            kernels.append(entry(label, summ[label], shapes))
        # the dominant kernel: largest summed time (alone, when the loop is software-pipelined; see `order_key`); its frac / avg_launch_us are timed-region figures
        roofline = None
        if kernels:
            roofline = {k: v for k, v in kernels[0].items() if k != "shapes"}
            roofline["kernels"] = kernels
            roofline["note"] = ("achieved = algorithmic flop (2*M*Cout*KH*KW*Cin; a split-bf16 kernel issues 3 MFMAs per product, so its peak is the bf16 "
                                "dense peak / 3) or algorithmic bytes per launch / HIP-event time on the launch stream, timed region only; traffic = "
                                "HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), averaged over the kernel's shapes; per shape: roofline_bound / "
                                "roofline_frac price the launch against the slower of its two limits (algorithmic flop at the matrix peak, algorithmic bytes "
                                "at 8 TB/s), hbm_frac = the byte limit alone"
                                + ("; the loop is software-pipelined (pose stage of the previous batch on a second stream beside these kernels), so the "
                                   "timed-region durations include the co-running launches; isolated_* = the same kernels in a single-stream pass "
                                   "of 3 steps after the timed region" if args.overlap else ""))
        total_frames = per_rank * world * args.steps
        line = {
            "metric": "RGB-D frames/sec (seg+DenseFusion+2-refine), 640x480 N=1000",
            "value": round(total_frames / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if args.frames else "weak", "vs_baseline": None,
            # operand type handed to the matrix cores; accumulation and stored activations are fp32.  "bf16x3" = split-bf16
            # (hi + lo, three MFMA products per term, ~2^-16 operand error): the fastest mode that passes the 1e-4 R/t and
            # bit-exact-mask parity tests (tests/test_gpu_pipeline.py); "f32" = exact fp32 MFMA; "bf16" = plain bf16.
            "dtype": args.seg_precision if args.seg_precision == args.pose_precision else
                     "seg:%s pose:%s" % (args.seg_precision, args.pose_precision),
            "data": "synthetic",
            "config": {"workload": ("configs[3]: %d synthetic 640x480 RGB-D frames per step sharded over the ranks (sub-batches of %d), "
                                    "one RCCL all_gather of all poses per step; per frame as configs[2]" % (args.frames, args.batch)) if args.frames else
                                   ("configs[2]: end-to-end PSPNet-resnet18 segmentation -> mask/CCL/bbox -> 160x160 crop -> "
                                    "PoseNet(N=1000) -> 2x PoseRefineNet, batch=%d 640x480 frames per GPU" % args.batch),
                       "inputs": "u8 RGB + u16 depth resident in HBM when the timed region starts, the same batch every step (`staged` = the "
                                 "same loop fed from pinned host memory)",
                       "frames_per_gpu_per_step": per_rank, "objects_found_last_step": n_found,
                       "crop_buckets_last_step": crop_hist,
                       "frame_selection": ("the rank's first %d random synthetic frames as they come" % per_rank) if args.unfiltered_frames else
                                          "one 160x160 detection per frame (configs[2]): random candidates whose segmentation gives a stray blob "
                                          "or a border-clipped crop are skipped during set-up, untimed (--unfiltered-frames times them all)",
                       "candidates_skipped": skipped,
                       "gflop_per_frame_algorithmic": GFLOP_PER_FRAME, "parallelism": "frames sharded x%d, 1 all_gather of poses/step" % world},
            "achieved_tflops_algorithmic": round(total_frames * GFLOP_PER_FRAME / dt / 1e3, 2),
            "overlap": bool(args.overlap),
            "ranks_seen": args.ranks_seen, "distinct_gpus": len({tuple(r[1:3]) for r in args.ranks_seen}),
            "roofline": roofline,
            "modes": modes,
            "staged": staged,
        }
        if sweep is not None:
            line["sweep"] = sweep
        if latency is not None:
            line["latency"] = latency
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks would idle at the barrier)
            if out.get("stream") is not None:
                out["stream"].synchronize()
            choose_h = out["choose"].cpu().numpy()
            by_frame = {(o[0], CLASSES[o[1] - 1]): k for k, o in enumerate(out["objects"])}

            def gpu_choose(fi, name, nz, n):
                k = by_frame.get((fi, name))
                if k is None:       # the GPU did not detect this object: any deterministic selection (counted in objects_not_matched)
                    return nz[(np.arange(n) * len(nz)) // n] if len(nz) > n else np.pad(nz, (0, n - len(nz)), "wrap")
                return choose_h[k]

            last = frames[(n_chunks - 1) * args.batch:]      # `out` is the last sub-batch of the last step
            line["cpu_baseline"], oracle_results = cpu_baseline(last, seg_sd, est_sd, ref_sd, gpu_choose, n_frames=min(16, args.batch),
                                                                n_frames_all=min(args.baseline_frames, args.batch))
            line["parity"] = parity_block(out, oracle_results, last, seg_sd)
        line.setdefault("parity", {})["steps_bitwise_equal"] = "skipped" if steps_equal is None else "%d/%d" % (steps_equal, args.steps)
        line["parity"]["steps_bitwise_equal_note"] = ("the [frames, 1, 8] result block of EVERY timed step (software-pipelined loop, pose stage on the "
                                                      "second stream) against the same step re-run on one stream after the timed region, torch.equal")
        if pose_leg is not None:
            line["pose"] = pose_leg
        if label_leg is not None:
            line["label"] = label_leg
        # the secondary legs in one short object at the END of the line (the driver's record keeps the last 2000 bytes of stdout verbatim
        # and only the key names of everything it does not know)
        sec = {"steps_bitwise_equal": line["parity"]["steps_bitwise_equal"]}
        if "max_dq" in line["parity"]:
            sec["parity"] = {k: line["parity"].get(k) for k in ("max_dq", "max_dt", "adds_delta_m", "mask_diff_px", "mask_diff_px_outside_tie_band")}
        if staged:
            sec["staged_frames_s"] = staged["value"]
        if modes:
            sec["f32_frames_s"] = modes["f32"]["value"]
        if sweep is not None:
            sec["sweep_frames_s"] = sweep.get("value")
        if latency is not None:
            sec["latency_p50_ms"] = latency.get("p50_ms")
        if pose_leg is not None:
            sec["pose"] = {"crops_s": pose_leg["value"], "ms_per_step": pose_leg["ms_per_step"],
                           "knn_1e6x1000_ms": pose_leg["knn_training_size"]["ms"], "knn_frac_of_fp32_lane_rate": pose_leg["knn_training_size"]["frac"],
                           "adds_launch_us": pose_leg["roofline"]["avg_launch_us"],
                           "knn_indices_bit_exact": pose_leg.get("parity", {}).get("knn_indices_bit_exact"),
                           "max_dq": pose_leg.get("parity", {}).get("max_dq"), "cpu_crops_s": pose_leg.get("cpu_baseline", {}).get("value")}
        if label_leg is not None:
            sec["label"] = {"views_s": label_leg["value"], "icp_pairs_s": label_leg["icp"]["point_pairs_per_s"],
                            "icp_gb_s": label_leg["roofline"]["achieved"], "icp_frac_of_8tb_s": label_leg["roofline"]["frac"],
                            "max_nn_distance_mm": label_leg.get("parity", {}).get("max_nn_distance_mm"),
                            "cpu_views_s": label_leg.get("cpu_baseline", {}).get("value")}
        line["secondary"] = sec
        print(json.dumps(line))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if steps_equal is not None and steps_equal != args.steps:
        raise SystemExit("bench.py: %d of %d timed steps of the overlapped loop differ bitwise from their single-stream re-run" % (args.steps - steps_equal, args.steps))


if __name__ == "__main__":
    main()
