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

import numpy as np
import torch

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


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json, made by
    tools/pmc_summary.py with the guide's FETCH_SIZE x2 gfx950 correction); None when no PMC run covers this kernel.
    PMC collection cannot run inside bench.py itself (it needs rocprofv3 around the process)."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            k = json.load(open(path))["kernels"].get(kernel.replace(" ", ""))
        except (OSError, ValueError, KeyError):
            continue
        if k:
            return k["hbm_bytes_per_launch"]
    return None


def make_frames(batch, rank):
    frames = []
    for i in range(batch):
        fid = rank * 100003 + i
        rng = np.random.default_rng(fid)
        box = (int(rng.integers(20, 480 - 150)), int(rng.integers(20, 640 - 150)))
        frames.append(S.synthetic_frame(fid, cls=1 + (i % 3), box=box, size=(126, 126)))
    return frames


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


def cpu_baseline(frames, seg_sd, est_sd, ref_sd, gpu_choose, n_frames=16):
    """The oracle (CPU restatement of pipeline/utils.py:410-641, pinned by the reference goldens) timed on the host cores on a bounded
    sample of the SAME frames.  The reference path is batch-1 Python + torch CPU and does not scale with intra-op threads (one frame
    on 128 threads is SLOWER than on one), so the fair multi-core figure runs independent frames side by side: W single-threaded
    workers (W = the CPUs of a one-GPU box share, at most 16), `n_frames` frames.  Also reported: one frame on one thread.
    Reported baseline only -- never part of `value`.  `gpu_choose(frame, class, nz, n)` injects the GPU run's point selection (the
    reference draws it from an unseeded shuffle), so the oracle results double as the checker of the `parity` block."""
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
    torch.set_num_threads(old_threads)
    found = sum(len(r) for r in results.values())
    return {"value": round(n_frames / dt, 4), "unit": "frames/s", "cores": workers, "kind": "port",
            "single_thread_value": round(1.0 / dt1, 4), "cpu_model": model, "physical_cores": phys, "usable_cpus": usable,
            "sample": "%d of the benchmark's 640x480 frames (%d objects found) through oracle/densefusion_oracle.full_prediction "
                      "(torch CPU fp32), %d single-threaded workers side by side; single_thread_value = 1 frame on 1 thread"
                      % (n_frames, found, workers)}, results


def parity_block(out, oracle_results, frames, seg_sd):
    """GPU (the timed path's last step) vs the oracle on the frames cpu_baseline ran: max |dq| (sign-aligned quaternion), max |dt| (m),
    differing mask pixels, and how many of those are NOT arg-max near-ties in the oracle's own probabilities (must be 0)."""
    from oracle import densefusion_oracle as O
    import torch.nn.functional as F
    objmap = out["objmap"].cpu().numpy()
    pose = out["pose"].cpu().numpy()
    max_dq = max_dt = 0.0
    diff_px = diff_outside = missing = 0
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
    return {"frames": len(oracle_results), "max_dq": max_dq, "max_dt": max_dt, "tolerance": 1e-4, "mask_diff_px": diff_px,
            "mask_diff_px_outside_tie_band": diff_outside, "objects_not_matched": missing,
            "note": "mask pixels may differ only where the oracle's own top-2 class probabilities are closer than 1e-4 (arg-max near-ties)"}


def kernel_peak(label):
    """(bound, peak, unit) for a profiled kernel label"""
    if "upconv_gather" in label:
        return "hbm", PEAK_HBM_GBS, "GB/s"
    if "conv_f32" in label:
        return "mfma", PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
    split = "_s32_kernel" in label or "<3," in label        # split-bf16: three bf16 MFMAs per algorithmic product
    return "mfma", PEAK_BF16_MFMA_TFLOPS / (3.0 if split else 1.0), "TFLOP/s"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", dest="overlap", action="store_true", default=True,
                    help="(default) pose stage of step i on a second HIP stream beside the segmentation of step i+1 (software-pipelined "
                         "loop, every step still does all of its work inside the fences): +4 %% frames/s; the per-kernel event timings "
                         "of the roofline leg then include whatever the co-running pose kernels cost them")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", help="one stream, segmentation and pose stage back to back")
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
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:       # under torch.distributed.run the RCCL path is exercised even for one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from autoposeestimation_amd.pipeline.utils import FramePipeline
    frames = make_frames(args.batch, rank)
    # three primary colours (classes 1..3) that a LINEAR read-out of the frozen random features separates cleanly from the
    # grey-noise background AND from each other's blurred borders (six colours left ~10 spurious >100-px detections per 64
    # frames); channels 4..12 of the 13-way segmentor stay silent
    fit_frames = [S.synthetic_frame(900 + 7 * c + k, cls=c, box=(30 + 45 * c + 20 * k, 20 + 60 * c + 90 * k), size=(126, 126))
                  for c in range(1, 4) for k in range(2)]
    seg, est, ref, seg_sd, est_sd, ref_sd = build_models(device, fit_frames)
    rgb = torch.from_numpy(np.stack([f[0] for f in frames])).to(device)        # inputs resident in HBM
    depth = torch.from_numpy(np.stack([f[1] for f in frames])).to(device)
    seg.set_precision(args.seg_precision)
    est.set_precision(args.pose_precision)
    ref.set_precision(args.pose_precision)
    # --overlap: the pose stage of step i runs on a second HIP stream beside the segmentation of step i+1 (the timed region ends
    # with torch.cuda.synchronize(), which waits for both streams; every step still does all of its work).  Off by default: the
    # live per-kernel timings of the roofline leg should be the kernels' own.
    pipe = FramePipeline(seg, est, ref, CLASSES, num_points=N_POINTS, refine_mode="live_compat", pose_stream=args.overlap)
    from autoposeestimation_amd.sharding import gather_results

    def tail(out):
        # one result slot per frame: the largest detection (the painted object) wins the slot
        with torch.cuda.stream(out.get("stream") or torch.cuda.current_stream()):     # the pose results live on the pose stream
            poses = torch.zeros(args.batch, 1, 8, dtype=torch.float32, device=device)
            if out["objects"]:
                o = np.asarray(out["objects"], dtype=np.int64)
                order = np.argsort(-(o[:, 3] - o[:, 2]) * (o[:, 5] - o[:, 4]), kind="stable")
                frames_u, first = np.unique(o[order, 0], return_index=True)
                pick = order[first]
                # one small H2D from pinned memory, non-blocking: a pageable copy would park the host behind the whole pose stage
                t = torch.from_numpy(np.stack([frames_u, pick, o[pick, 1]])).pin_memory().to(device, non_blocking=True)
                poses[t[0], 0, 0] = t[2].float()
                poses[t[0], 0, 1:] = out["pose"][t[1]].float()
            out["gathered"] = gather_results(poses, dist)   # the single RCCL collective of the path: (cls, q, t) per frame
        return out

    def run_steps(first, count):
        """`count` steps (batches `first` .. `first + count - 1`), every one a full pass seg -> CCL -> crops -> PoseNet -> 2x refine ->
        gather.  With the pose stream the loop is software-pipelined: the segmentation of step i+1 is enqueued BEFORE the host
        waits for step i's detections, so the main stream never idles while the host enqueues step i's ~100 pose launches.
        Exactly `count` segmentation stages and `count` pose stages are enqueued, all inside the caller's fences."""
        out = None
        if not args.overlap:
            for i in range(first, first + count):
                out = tail(pipe.run(rgb, depth, S.REALSENSE_META, seed=i))
            return out
        h = pipe.begin(rgb) if count else None
        for i in range(first, first + count):
            h_next = pipe.begin(rgb) if i + 1 < first + count else None
            out = tail(pipe.finish(h, rgb, depth, S.REALSENSE_META, seed=i))
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
        out = run_steps(args.warmup - 1, 1)
    n_found = len(out["objects"]) if args.warmup else -1
    fence()
    if args.warmup:
        wsum = E.PROFILE.summary()
        if wsum:
            top_only = set(sorted(wsum, key=lambda k: -wsum[k]["ms"])[:5])
    prof = E.PROFILE = E.LaunchProfile(only=top_only)
    t0 = time.perf_counter()
    out = run_steps(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    E.PROFILE = None
    n_found = len(out["objects"])
    crop_hist = {}
    for o in out["objects"]:
        k = "%dx%d" % (o[3] - o[2], o[5] - o[4])
        crop_hist[k] = crop_hist.get(k, 0) + 1
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    if rank == 0:
        summ = prof.summary()
        by_shape = prof.summary(by_shape=True)

        def entry(label, d, shapes=None):
            bound, peak, unit = kernel_peak(label)
            sec = d["ms"] * 1e-3
            ach = (d["bytes"] / sec / 1e9) if bound == "hbm" else (d["flop"] / sec / 1e12)
            traffic = pmc_traffic(label)
            alg_b = d["bytes"] / d["launches"]
            e = {"kernel": label, "bound": bound, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": unit, "frac": round(ach / peak, 4),
                 "launches": d["launches"], "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 1),
                 "avg_launch_gflop": round(d["flop"] / d["launches"] / 1e9, 3), "avg_launch_algorithmic_mb": round(alg_b / 1e6, 1),
                 "traffic": traffic, "traffic_over_algorithmic": None if traffic is None else round(traffic / alg_b, 2),
                 "share_of_step_time": round(sec / dt, 3)}
            if shapes:
                e["shapes"] = shapes
            return e

        kernels = []
        for label in sorted(summ, key=lambda k: -summ[k]["ms"]):
            shapes = []
            for (lab, shape), d in sorted(by_shape.items(), key=lambda kv: -kv[1]["ms"]):
                if lab != label:
                    continue
                bound, peak, unit = kernel_peak(label)
                sec = d["ms"] * 1e-3
                ach = (d["bytes"] / sec / 1e9) if bound == "hbm" else (d["flop"] / sec / 1e12)
                shapes.append({"shape": shape, "launches": d["launches"], "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 1),
                               "gflop": round(d["flop"] / d["launches"] / 1e9, 3), "algorithmic_mb": round(d["bytes"] / d["launches"] / 1e6, 1),
                               "achieved": round(ach, 2), "frac": round(ach / peak, 4)})
            kernels.append(entry(label, summ[label], shapes))
        # the dominant kernel: largest summed time over the timed region (deterministic given the timings)
        roofline = None
        if kernels:
            roofline = {k: v for k, v in kernels[0].items() if k != "shapes"}
            roofline["kernels"] = kernels
            roofline["note"] = ("achieved = algorithmic flop (2*M*Cout*KH*KW*Cin; a split-bf16 kernel issues 3 MFMAs per product, so its peak is the bf16 "
                                "dense peak / 3) or algorithmic bytes per launch / HIP-event time on the launch stream, timed region only; traffic = "
                                "HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), averaged over the kernel's shapes"
                                + ("; --overlap: the pose stage of the previous step runs beside these kernels on a second stream" if args.overlap else ""))
        total_frames = args.batch * world * args.steps
        line = {
            "metric": "RGB-D frames/sec (seg+DenseFusion+2-refine), 640x480 N=1000",
            "value": round(total_frames / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            # operand type handed to the matrix cores; accumulation and stored activations are fp32.  "bf16x3" = split-bf16
            # (hi + lo, three MFMA products per term, ~2^-16 operand error): the fastest mode that passes the 1e-4 R/t and
            # bit-exact-mask parity tests (tests/test_gpu_pipeline.py); "f32" = exact fp32 MFMA; "bf16" = plain bf16.
            "dtype": args.seg_precision if args.seg_precision == args.pose_precision else
                     "seg:%s pose:%s" % (args.seg_precision, args.pose_precision),
            "data": "synthetic",
            "config": {"workload": "configs[2]: end-to-end PSPNet-resnet18 segmentation -> mask/CCL/bbox -> 160x160 crop -> "
                                   "PoseNet(N=1000) -> 2x PoseRefineNet, batch=%d 640x480 frames per GPU" % args.batch,
                       "frames_per_gpu_per_step": args.batch, "objects_found_last_step": n_found,
                       "crop_buckets_last_step": crop_hist,
                       "gflop_per_frame_algorithmic": GFLOP_PER_FRAME, "parallelism": "frames sharded x%d, 1 all_gather of poses/step" % world},
            "achieved_tflops_algorithmic": round(total_frames * GFLOP_PER_FRAME / dt / 1e3, 2),
            "overlap": bool(args.overlap),
            "roofline": roofline,
        }
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

            line["cpu_baseline"], oracle_results = cpu_baseline(frames, seg_sd, est_sd, ref_sd, gpu_choose, n_frames=min(16, args.batch))
            line["parity"] = parity_block(out, oracle_results, frames, seg_sd)
        print(json.dumps(line))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
