"""Drop-in for the live-prediction surface of pipeline/utils.py: `full_prediction` (reference :410-641) and
`get_prediction_models` (:643-718), plus `FramePipeline`, the batched device-resident form the benchmark drives.

Per frame the reference crosses host<->device ~8 times and loops in Python over classes / components / 307 200-element
index maps.  Here a batch of B frames stays in HBM end to end:

    u8 RGB + u16 depth  -> ToTensor/Normalize -> PSPNet segmentor -> softmax^2/argmax -> CCL + best component + bbox
      --(ONE small D2H: det[B,C,5] int32, needed because crop shapes are data-dependent)-->
    per (Hc,Wc) bucket:  choose/compaction -> back-projection -> crop normalise -> PoseNet -> pose select
                         -> 2x PoseRefineNet -> float64 compose        ==> poses[n_obj,7] f64 on the device

`refine_mode='live_compat'` reproduces the reference live loop literally (pipeline/utils.py:569-571: both refiner
forwards see the SAME new_points and the residual is composed once); `'iterative'` is the upstream loop
(DenseFusion/tools/eval_ycb.py:205-229).  Both run two real refiner forwards.
"""
import os
import time

import numpy as np
import torch

from autoposeestimation_amd import engine as E
from autoposeestimation_amd.DenseFusion.lib.network import PoseNet, PoseRefineNet


def _pose_stream():
    want = os.environ.get("APE_POSE_STREAM_PRIORITY", "low")
    if want == "default":
        return torch.cuda.Stream()
    least, greatest = torch.cuda.Stream.priority_range()         # (numerically largest = least urgent, smallest = most urgent)
    return torch.cuda.Stream(priority=least if want == "low" else greatest if want == "high" else int(want))


class FramePipeline:
    def __init__(self, segmentor, estimator, refiner, class_names, num_points=1000, refine_mode="live_compat",
                 min_pixels=100, iterations=2, pose_stream=False, pose_graphs=False, low_latency=False):
        if refine_mode not in ("live_compat", "iterative"):
            raise ValueError(refine_mode)
        self.segmentor, self.estimator, self.refiner = segmentor, estimator, refiner
        self.class_names = list(class_names)
        self.n_cls = len(self.class_names) + 1          # + background (pipeline/utils.py:695)
        self.num_points, self.refine_mode, self.min_pixels, self.iterations = num_points, refine_mode, min_pixels, iterations
        self._full_rects = {}
        # pose_stream: the pose stage of a batch (many small launches that leave most CUs idle) is enqueued on a SECOND HIP
        # stream, so that the segmentation of the NEXT batch -- enqueued on the caller's stream as soon as run() returns -- fills
        # the chip beside it.  The returned dict then carries `stream`: whoever consumes pose / n_cand / choose enqueues on it
        # (or waits for it); torch.cuda.synchronize() covers both.
        # The pose stream gets the LOWEST HIP stream priority the device offers: its small launches then take the CUs the segmentation
        # kernels leave idle instead of competing with them for every slot (APE_POSE_STREAM_PRIORITY=default turns that off).
        self.side = _pose_stream() if pose_stream else None
        # pose_graphs: frames with several objects of different sizes give one pose-stage pass (~90 launches) per crop-size bucket, and
        # the host's launch rate, not the GPU, bounds the step (bench.py --mixed: 19 buckets, ~1700 launches, 78 ms).  With this switch
        # the launches of a bucket (crop size, object count) are captured ONCE in a HIP graph -- on the bucket's second occurrence; the
        # first runs eagerly and does the lazy set-up -- and later steps replay it: one graph launch per bucket.  The graph reads the
        # batch through fixed buffers (the caller's rgb / depth tensors by address: new tensors mean a new capture; the object map,
        # the bucket's object table and the sampling seed through the pipeline's own static copies) and leaves pose / n_cand / choose in
        # static outputs that the caller copies from.  Results are those of the eager launches, kernel for kernel.
        # Replays of different buckets are independent and each fills a small part of the chip (a bucket holds 1..10 crops): they go out on
        # side streams side by side (APE_BUCKET_STREAMS, default 32: one per bucket) and the caller's stream joins them before it scatters
        # the results.  HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4: two chains then run side by side,
        # not four); the process has to set that variable before its first HIP call (bench.py does: 40) for the streams to be concurrent.
        self.pose_graphs = bool(pose_graphs)
        self._bucket_streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("APE_BUCKET_STREAMS", "32")))] if pose_graphs else []
        self._graphs = {}
        self._seen = set()
        self._static_objmap = None
        # low_latency: the pose networks' small-M layers (ONE crop: 4..16 output tiles per layer on a 256-CU chip) take the split-K form of
        # ape_conv_gemm_bf16_splitk -- the batch-1 live loop (main.py:517-553, full_prediction): 3.9 -> 3.0 ms per frame.  Off by default:
        # the split changes the fp32 summation order (poses move by <= 1e-5, still 10x inside the 1e-4 bar), and a batched run is held to
        # 1e-6 against its frames run alone (tests/test_gpu_bench_parity.py); in a batch that fills the chip only the layers that are small at any
        # batch size would split (the PSP prior branches, the heads' per-crop bias)
        self.low_latency = bool(low_latency)
        self.host_poses_s = 0.0        # host time spent inside poses() (enqueue only: nothing in there waits for the GPU once the graphs exist)

    # -- stage 1: segmentation + components, all on device ---------------------------------------------------------
    def segment(self, rgb, inject_logits=None):
        """rgb[B,H,W,3] u8 cuda -> objmap[B,H,W] u8, det[B,C,5] i32.  `inject_logits[B,H,W,C]` bypasses the CNN
        (parity tests of the integer post-processing, SURVEY.md section 7 "Bit-exact masks")."""
        b, h, w, _ = rgb.shape
        if inject_logits is None:
            key = (b, str(rgb.device))
            rects = self._full_rects.get(key)
            if rects is None:
                rects = torch.zeros(b, 3, dtype=torch.int32)
                rects[:, 0] = torch.arange(b, dtype=torch.int32)
                rects = self._full_rects[key] = rects.to(rgb.device)
            x4 = E.U8Frames(rgb, rects, h, w, div255=True)       # (normalised inside the stem kernel's patch load: never materialised)
            label, score = self.segmentor.label_score_nhwc(x4, double_softmax=True)
        else:
            label, score = E.seg_argmax(inject_logits, self.n_cls, double_softmax=True)
        return E.seg_components(label, score, self.n_cls, self.min_pixels)

    # -- stage 2: pose for a list of detected objects --------------------------------------------------------------
    def poses(self, rgb, depth, objmap, objects, meta, choose_override=None, seed=0):
        """objects: list of (frame, cls, rmin, rmax, cmin, cmax).  Returns (pose[n,7] f64 cuda, n_cand[n] i32 cuda,
        choose[n,N] i64 cuda) in the order of `objects`."""
        t_host = time.perf_counter()
        prev, E.SPLITK_SMALL_M = E.SPLITK_SMALL_M, self.low_latency
        try:
            return self._poses(rgb, depth, objmap, objects, meta, choose_override, seed)
        finally:
            E.SPLITK_SMALL_M = prev
            self.host_poses_s += time.perf_counter() - t_host

    def _poses(self, rgb, depth, objmap, objects, meta, choose_override, seed):
        n = len(objects)
        dev = rgb.device
        obj_np = np.asarray(objects, dtype=np.int32).reshape(n, 6)
        sizes = np.stack([obj_np[:, 3] - obj_np[:, 2], obj_np[:, 5] - obj_np[:, 4]], 1)
        uniq, inv = np.unique(sizes, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        single = len(uniq) == 1                      # one crop shape (the common case): results are already in object order
        if not single:
            pose_all = torch.zeros(n, 7, dtype=torch.float64, device=dev)
            ncand_all = torch.zeros(n, dtype=torch.int32, device=dev)
            choose_all = torch.zeros(n, self.num_points, dtype=torch.int64, device=dev)
        use_graphs = self.pose_graphs and choose_override is None and E.PROFILE is None and not torch.cuda.is_current_stream_capturing()
        if use_graphs:
            if self._static_objmap is None or self._static_objmap.shape != objmap.shape:
                self._static_objmap = torch.empty_like(objmap)
                self._graphs.clear()
            self._static_objmap.copy_(objmap, non_blocking=True)
        cur = torch.cuda.current_stream()
        fan = use_graphs and not single and len(self._bucket_streams) > 1
        if fan:
            start = torch.cuda.Event()
            start.record(cur)
            joins = []
        # buckets in the order of falling work (crop area x objects): their replays run side by side on the bucket streams, so the longest
        # chain should start first; results are scattered by object index, the order changes nothing else
        work = [-(int(hc) * int(wc) * int((inv == k).sum())) for k, (hc, wc) in enumerate(uniq.tolist())]
        for pos, k in enumerate(np.argsort(work, kind="stable").tolist()):
            hc, wc = map(int, uniq[k])
            ids = np.nonzero(inv == k)[0]
            sub = obj_np[ids]
            # one small H2D, from pinned memory and non-blocking: a pageable copy would park the host until the stream (the pose stream
            # still busy with the previous batch) reaches it
            both_h = torch.from_numpy(np.ascontiguousarray(np.concatenate([sub, sub[:, [0, 2, 4]]], 1))).pin_memory()
            if use_graphs:
                side = self._bucket_streams[pos % len(self._bucket_streams)] if fan and self._has_graph(rgb, depth, both_h.shape[0], hc, wc, meta, cur) else None
                if side is not None:            # a replay: on one of the bucket streams, beside the other buckets' replays
                    side.wait_event(start)
                    with torch.cuda.stream(side):
                        pose, n_cand, choose = self._bucket_replay(rgb, depth, both_h, hc, wc, meta, seed, key_stream=cur)
                        for t in (pose, n_cand, choose):
                            t.record_stream(cur)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    joins.append(ev)
                    cur.wait_event(ev)
                else:
                    pose, n_cand, choose = self._bucket_replay(rgb, depth, both_h, hc, wc, meta, seed)
            else:
                pose, n_cand, choose = self._bucket(rgb, depth, objmap, both_h.to(dev, non_blocking=True), hc, wc, meta, seed,
                                                    None if choose_override is None else {j: choose_override.get(i) for j, i in enumerate(ids.tolist())})
            if single:
                return pose, n_cand, choose
            ids_t = torch.from_numpy(ids.astype(np.int64)).pin_memory().to(dev, non_blocking=True)
            pose_all.index_copy_(0, ids_t, pose)
            ncand_all.index_copy_(0, ids_t, n_cand)
            choose_all.index_copy_(0, ids_t, choose)
        return pose_all, ncand_all, choose_all

    def _bucket(self, rgb, depth, objmap, both, hc, wc, meta, seed, override=None):
        """the pose stage of ONE crop-size bucket: both[n,9] i32 on the device = (frame, cls, rmin, rmax, cmin, cmax | frame, rmin, cmin);
        -> pose[n,7] f64, n_cand[n] i32, choose[n,N] i64"""
        dev = rgb.device
        objs, rects = both[:, :6].contiguous(), both[:, 6:9].contiguous()
        choose, n_cand = E.choose_points(objmap, depth, objs, self.num_points, seed)
        if override is not None:
            for j, ch in override.items():
                if ch is not None:
                    choose[j] = torch.as_tensor(ch, dtype=torch.int64).to(dev)
        pts4 = E.backproject(depth, objs, choose, meta["intr"], meta["depth_scale"])
        img4 = E.U8Frames(rgb, rects, hc, wc, div255=False)
        obj_idx = (objs[:, 1].to(torch.int64) - 1).contiguous()         # class_names.index(cls) (pipeline/utils.py:561)
        heads, emb = self.estimator.forward_batch(img4, pts4, choose, obj_idx)
        if self.refine_mode == "live_compat":
            pose, _, newp = E.pose_select(heads, pts4)
            for _ in range(self.iterations):
                out = self.refiner.forward_batch(newp, emb, obj_idx)
            E.pose_compose(pose, out[:, 0:4], out[:, 4:7])
        else:
            pose, _, _ = E.pose_select(heads, pts4, want_new_points=False)
            for _ in range(self.iterations):
                out = self.refiner.forward_batch(E.pose_recentre(pts4, pose), emb, obj_idx)
                E.pose_compose(pose, out[:, 0:4], out[:, 4:7])
        return pose, n_cand, choose

    def _graph_key(self, rgb, depth, n, hc, wc, meta, stream):
        intr = meta["intr"]
        return (hc, wc, n, rgb.data_ptr(), depth.data_ptr(), float(intr["fx"]), float(intr["fy"]), float(intr["ppx"]), float(intr["ppy"]),
                float(meta["depth_scale"]), stream.cuda_stream)

    def _has_graph(self, rgb, depth, n, hc, wc, meta, stream):
        return self._graph_key(rgb, depth, n, hc, wc, meta, stream) in self._graphs

    def _bucket_replay(self, rgb, depth, both_h, hc, wc, meta, seed, key_stream=None):
        """the same through a HIP graph per (crop size, object count, input buffers): see `pose_graphs` in __init__"""
        n = both_h.shape[0]
        key = self._graph_key(rgb, depth, n, hc, wc, meta, key_stream or torch.cuda.current_stream())
        g = self._graphs.get(key)
        if g is None:
            if key not in self._seen:                   # first occurrence: eager (weight plans, function attributes, lazy operands)
                self._seen.add(key)
                return self._bucket(rgb, depth, self._static_objmap, both_h.to(rgb.device, non_blocking=True), hc, wc, meta, seed)
            if len(self._graphs) >= 256:
                self._graphs.clear()
            both = torch.empty(n, 9, dtype=torch.int32, device=rgb.device)
            seed_d = torch.zeros(1, dtype=torch.int32, device=rgb.device)
            both.copy_(both_h, non_blocking=True)
            torch.cuda.current_stream().synchronize()   # (capture starts from a quiet stream; once per bucket shape)
            graph = torch.cuda.CUDAGraph()
            E.CAPTURING = True                          # (no cached workspaces from inside a graph's private pool)
            try:
                with torch.cuda.graph(graph):
                    outs = self._bucket(rgb, depth, self._static_objmap, both, hc, wc, meta, seed_d)
            finally:
                E.CAPTURING = False
            g = self._graphs[key] = (graph, both, seed_d, outs)
        graph, both, seed_d, outs = g
        if os.environ.get("APE_FAN_EAGER") == "1":      # (diagnosis: the bucket's launches one by one on this stream instead of the replay)
            return self._bucket(rgb, depth, self._static_objmap, both_h.to(rgb.device, non_blocking=True), hc, wc, meta, seed)
        both.copy_(both_h, non_blocking=True)
        seed_d.fill_(int(seed) & 0x7FFFFFFF)
        graph.replay()
        return tuple(o.clone() for o in outs)           # (the static outputs are overwritten by the bucket's next replay)

    def begin(self, rgb, inject_logits=None, asynchronous=True):
        """Enqueue the segmentation stage of a batch and the (pinned, non-blocking) copy of its detections; returns a handle for
        finish().  Nothing here waits for the GPU, so the next batch can be begun before this one is finished."""
        objmap, det = self.segment(rgb, inject_logits)
        if not asynchronous:
            return {"objmap": objmap, "det": det, "det_h": None, "event": None}
        det_h = torch.empty(det.shape, dtype=det.dtype, pin_memory=True)
        det_h.copy_(det, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return {"objmap": objmap, "det": det, "det_h": det_h, "event": ev}

    def finish(self, handle, rgb, depth, meta, choose_override=None, seed=0):
        """Wait for the batch's detections (the one host sync of the batch -- an event, not the stream, so segmentation work
        queued behind it keeps running), then enqueue its pose stage (on the pose stream when the pipeline has one)."""
        if handle["event"] is not None:
            handle["event"].synchronize()
            det_h = handle["det_h"].numpy()
        else:
            det_h = handle["det"].cpu().numpy()         # plain blocking copy (run(): nothing is queued behind it)
        objmap = handle["objmap"]
        fb, fc = np.nonzero(det_h[:, 1:, 0])            # (frame, class - 1) of every detection, frame-major like the reference loop
        objects = [(int(b), int(c) + 1, *map(int, det_h[b, c + 1, 1:5])) for b, c in zip(fb, fc)]
        if objects:
            if choose_override is not None:             # keyed by (frame, cls) -> keyed by object position
                choose_override = {i: choose_override.get((o[0], o[1])) for i, o in enumerate(objects)}
            if self.side is not None:
                if handle["event"] is not None:
                    self.side.wait_event(handle["event"])
                else:
                    self.side.wait_stream(torch.cuda.current_stream())
                objmap.record_stream(self.side)         # allocated on the caller's stream, read by the pose stage on the side stream
                with torch.cuda.stream(self.side):
                    pose, n_cand, choose = self.poses(rgb, depth, objmap, objects, meta, choose_override, seed)
                return {"objects": objects, "pose": pose, "n_cand": n_cand, "choose": choose, "objmap": objmap, "stream": self.side}
            pose, n_cand, choose = self.poses(rgb, depth, objmap, objects, meta, choose_override, seed)
        else:
            pose = torch.zeros(0, 7, dtype=torch.float64, device=rgb.device)
            n_cand = torch.zeros(0, dtype=torch.int32, device=rgb.device)
            choose = torch.zeros(0, self.num_points, dtype=torch.int64, device=rgb.device)
        return {"objects": objects, "pose": pose, "n_cand": n_cand, "choose": choose, "objmap": objmap}

    def run(self, rgb, depth, meta, inject_logits=None, choose_override=None, seed=0):
        """Whole batch.  Returns dict(objects=[(frame,cls,rmin,rmax,cmin,cmax)], pose, n_cand, choose, objmap)."""
        return self.finish(self.begin(rgb, inject_logits, asynchronous=False), rgb, depth, meta, choose_override, seed)


def _as_u8_frame(image):
    arr = np.asarray(image)
    if arr.ndim != 3 or arr.shape[2] < 3:
        raise ValueError("image must be HxWx3 uint8")
    return np.ascontiguousarray(arr[:, :, :3].astype(np.uint8))


def full_prediction(image, depth, meta, segmentor, estimator, refiner, to_tensor, normalize, device, cuda, color_dict,
                    class_names=None, point_clouds=None, plot=False, color_prediction=False, bbox=False, put_text=False,
                    refine_mode="live_compat", choose_override=None):
    """Reference signature (pipeline/utils.py:410-411) and output dict:
        {'predictions': {cls_name: {'mask': u8[H,W] in {0,255}, 'position': f64[3] (m), 'rotation': f64[4] wxyz}},
         'elapsed_times': {'segmentation', 'pose_estimation', 'total'}}
    `to_tensor` / `normalize` are accepted for signature compatibility; the normalisation runs on the device.
    `color_prediction` adds 'segmented_prediction' / 'pose_prediction' images (plain alpha blends; cv2 text/boxes are
    skipped when OpenCV is absent).  Objects without a valid depth pixel are dropped like the reference (:530-531,623-635)."""
    if not cuda or not torch.cuda.is_available():
        raise RuntimeError("full_prediction needs the GPU: the MI355X path has no CPU fallback")
    start_time = time.time()
    rgb_np = _as_u8_frame(image)
    depth_np = np.ascontiguousarray(np.asarray(depth))
    if depth_np.dtype != np.uint16:
        if (depth_np < 0).any() or (depth_np > 65535).any() or (depth_np != np.floor(depth_np)).any():
            raise ValueError("depth must hold integer sensor units in 0..65535")
        depth_np = depth_np.astype(np.uint16)
    rgb = torch.from_numpy(rgb_np).to(device).unsqueeze(0)
    dep = torch.from_numpy(depth_np).to(device).unsqueeze(0)
    pipe = FramePipeline(segmentor, estimator, refiner, class_names, refine_mode=refine_mode, low_latency=True)       # (one frame per call: the live loop)
    objmap, det = pipe.segment(rgb)
    det_h = det.cpu().numpy()
    objmap_h = objmap[0].cpu().numpy()
    output_dict = {"predictions": {}, "elapsed_times": {}}
    objects = []
    for c, d in enumerate(det_h[0]):
        if c > 0 and d[0]:
            name = class_names[c - 1]
            output_dict["predictions"][name] = {"mask": np.where(objmap_h == c, 255, 0).astype(np.uint8)}
            objects.append((0, c, int(d[1]), int(d[2]), int(d[3]), int(d[4])))
    output_dict["elapsed_times"]["segmentation"] = time.time() - start_time
    start_time_pose = time.time()
    if objects:
        ov = None
        if choose_override is not None:
            ov = {i: choose_override.get(class_names[o[1] - 1]) for i, o in enumerate(objects)}
        pose, n_cand, _ = pipe.poses(rgb, dep, objmap, objects, meta, ov)
        pose_h, n_cand_h = pose.cpu().numpy(), n_cand.cpu().numpy()
        for i, o in enumerate(objects):
            name = class_names[o[1] - 1]
            if n_cand_h[i] == 0:
                print('Deleting cls "{}"'.format(name))
                del output_dict["predictions"][name]
                continue
            output_dict["predictions"][name]["position"] = pose_h[i, 4:7].copy()
            output_dict["predictions"][name]["rotation"] = pose_h[i, 0:4].copy()
    if color_prediction:
        from autoposeestimation_amd.pc_reconstruction import open3d_utils as pc_utils
        seg_img = rgb_np.astype(np.float64)
        pose_img = rgb_np.astype(np.float64)
        for name, pred in output_dict["predictions"].items():
            colour = np.asarray(color_dict[name]["value"], dtype=np.float64)
            m = pred["mask"] != 0
            seg_img[m] = seg_img[m] * 0.7 + colour * 0.3
            if point_clouds is not None:
                from autoposeestimation_amd.DenseFusion.lib.transformations import quaternion_matrix
                R = quaternion_matrix(pred["rotation"])[:3, :3]
                cloud = np.dot(point_clouds[class_names.index(name)], R.T) + pred["position"]
                pose_img = pc_utils.pointcloud2image(pose_img, cloud, 3, meta["intr"], color=list(colour))
        output_dict["segmented_prediction"] = np.clip(seg_img, 0, 255).astype(np.uint8)
        output_dict["pose_prediction"] = np.clip(pose_img, 0, 255).astype(np.uint8)
    output_dict["elapsed_times"]["pose_estimation"] = time.time() - start_time_pose
    output_dict["elapsed_times"]["total"] = time.time() - start_time
    return output_dict


def _axangle2mat(axis, angle):
    """transforms3d.axangles.axangle2mat (Rodrigues), as used at pipeline/utils.py:390 -- UNPINNED third-party maths"""
    x, y, z = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    c, s = np.cos(angle), np.sin(angle)
    C = 1 - c
    return np.array([[x * x * C + c, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, y * y * C + c, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, z * z * C + c]])


def get_robot2object(prediction, controller, end2cam):
    """reference pipeline/utils.py:381-408: move every predicted camera-frame pose (m, wxyz) into the robot frame with the
    controller's end-effector pose (axis-angle a,b,c + x,y,z in mm) and the hand-eye transform `end2cam`."""
    from autoposeestimation_amd.DenseFusion.lib.transformations import quaternion_from_matrix, quaternion_matrix
    if len(prediction["predictions"].keys()) > 0:
        pose = controller.get_pose(return_mm=True)
        r = np.array([pose["a"], pose["b"], pose["c"]], dtype=np.float64)
        angle = np.linalg.norm(r)
        robot2end = np.identity(4)
        robot2end[:3, :3] = _axangle2mat(r / angle, angle) if angle > 0 else np.identity(3)
        robot2end[:3, 3] = [pose["x"], pose["y"], pose["z"]]
        robot2cam = np.dot(robot2end, end2cam)
        for cls in prediction["predictions"]:
            cam2obj = quaternion_matrix(prediction["predictions"][cls]["rotation"])
            cam2obj[:3, 3] = np.asarray(prediction["predictions"][cls]["position"]) * 1000
            robot2obj = np.dot(robot2cam, cam2obj)
            prediction["predictions"][cls]["position"] = robot2obj[:3, 3] / 1000
            m = np.identity(4)
            m[:3, :3] = robot2obj[:3, :3]
            prediction["predictions"][cls]["rotation"] = quaternion_from_matrix(m)
    return prediction


class _ToTensor:
    """torchvision.transforms.ToTensor stand-in returned by get_prediction_models (HWC u8 -> CHW float /255)."""

    def __call__(self, pic):
        arr = np.asarray(pic)
        return torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).float().div(255)


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean.to(t.device)) / self.std.to(t.device)


def read_xyz_cloud(path, to_meter=True):
    """`<cls>.xyz` text parser (pipeline/utils.py:667-684): one bracketed, space-separated point per line, mm."""
    pts = []
    with open(path) as f:
        for line in f:
            vals = [float(v) for v in line.strip().strip("[]").split() if v]
            if len(vals) >= 3:
                pts.append(vals[:3])
    arr = np.array(pts, dtype=np.float64)
    return arr / 1000.0 if to_meter else arr


def get_prediction_models(root, data_set_name, segmentor_name="PsPNet", encoder_name="resnet34"):
    """Reference pipeline/utils.py:643-718: returns (segmentor, estimator, refiner, classes, to_tensor, normalize, cld,
    device, cuda).  Weights are read from the same files (`pose_model.pth`, `pose_refine_model.pth`, and the segmentor
    checkpoint `{name}_{encoder}.ckpt` with a 'state_dict' entry, label_generator/create_labels.py:29-35)."""
    from autoposeestimation_amd.label_generator.create_labels import get_default_model
    if not torch.cuda.is_available():
        raise RuntimeError("get_prediction_models needs the GPU: the MI355X path has no CPU fallback")
    device, cuda = torch.device("cuda:0"), True
    classes, cld = [], {}
    with open(os.path.join(root, "label_generator", "data_sets", "segmentation", data_set_name, "classes.txt")) as f:
        for line in f:
            name = line.strip()
            if not name:
                break
            cld[len(classes)] = read_xyz_cloud(os.path.join(root, "pc_reconstruction", "data", name, "{}.xyz".format(name)))
            classes.append(name)
    to_tensor = _ToTensor()
    normalize = _Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    segmentor = get_default_model(root, data_set_name, len(classes) + 1, name=segmentor_name, encoder_name=encoder_name)
    segmentor.to(device).eval()
    pose_path = os.path.join(root, "DenseFusion", "trained_models", data_set_name)
    estimator = PoseNet(num_points=1000, num_obj=len(classes))
    refiner = PoseRefineNet(num_points=1000, num_obj=len(classes))
    estimator.load_state_dict(torch.load(os.path.join(pose_path, "pose_model.pth"), map_location="cpu"))
    refiner.load_state_dict(torch.load(os.path.join(pose_path, "pose_refine_model.pth"), map_location="cpu"))
    estimator.to(device).eval()
    refiner.to(device).eval()
    return segmentor, estimator, refiner, classes, to_tensor, normalize, cld, device, cuda
