"""bench.py's contract, checked end to end: the three workloads run as the driver runs them (a child process, one JSON line on stdout)
and the line carries every field the contract names, with consistent values."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                     # ONE JSON line
    return json.loads(lines[0])


def _common(d, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_default_line_has_roofline_cpu_baseline_and_parity():
    d = _run("--steps", "2", "--warmup", "1")
    _common(d, 2, 1)
    assert d["unit"] == "frames/s" and d["scaling"] == "weak" and d["dtype"] == "bf16x3"
    assert abs(d["value"] - 64 * 2 / (d["ms_per_step"] * 2e-3)) < 0.02 * d["value"]
    # the line says what it measures (inputs resident in HBM) and carries the same loop fed from pinned host memory beside it
    assert "resident in HBM" in d["config"]["inputs"]
    st = d["staged"]
    assert st["unit"] == "frames/s" and st["host_bytes_per_step"] == 64 * 480 * 640 * 5 and st["objects_found_last_step"] == 64
    assert st["value"] >= 0.9 * d["value"]              # the 98 MB per step hide behind the segmentation (within 3 % on a quiet box; 10 % here: short runs)
    # compact secondary legs the plain command carries: four steps of the 1-3-objects / five-crop-sizes frames, 50 runs of the batch-1 live loop
    sw, lat = d["sweep"], d["latency"]
    assert sw["steps"] == 4 and sw["unit"] == "frames/s" and 0 < sw["value"] < d["value"] * 1.2 and len(sw["crop_buckets_last_step"]) >= 4
    assert "parity" not in sw and sw["pose_graphs"] is True
    assert lat["runs"] == 50 and lat["objects"] == 1 and 0 < lat["min_ms"] <= lat["p50_ms"] <= lat["p99_ms"] < 50
    # the leg runs the low-latency pipeline (split-K for the crop's small-M layers) and reports the unsplit form and the graph replay beside it
    assert lat["graph_equals_eager_bitwise"] is True and lat["unsplit_p50_ms"] > 0 and 0 < lat["split_vs_unsplit_max_abs_pose_diff"] <= 2e-5
    ks = d["roofline"]["kernels"]
    assert len(ks) == 5 and d["roofline"]["kernel"] == ks[0]["kernel"]
    for k in ks:
        assert k["launches"] > 0 and k["avg_launch_us"] > 0 and 0 < k["frac"] < 1 and k["shapes"] and "isolated_frac" in k
        assert sum(s["launches"] for s in k["shapes"]) == k["launches"]
        for s in k["shapes"]:                           # every shape priced against its OWN limit (flop at the matrix peak / bytes at 8 TB/s)
            assert s["roofline_bound"] in ("hbm", "mfma") and 0 < s["roofline_frac"] < 1 and s["roofline_frac"] >= max(s["frac"], s["hbm_frac"]) - 1e-3
            assert "isolated_roofline_frac" in s and s["isolated_avg_launch_us"] > 0
    # the five kernels are ranked by what they cost alone: the first one is a segmentation conv kernel, not a starved pose-stage launch
    iso = [k["isolated_avg_launch_us"] * k["launches"] for k in ks]
    assert iso == sorted(iso, reverse=True) and ks[0]["kernel"].startswith(("halo_s32_kernel", "gemm_s32_kernel", "conv3x3_halo_kernel"))
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "single_thread_value", "cpu_model"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["value"] > 0
    p = d["parity"]
    assert p["frames"] >= 8 and p["max_dq"] <= 1e-4 and p["max_dt"] <= 1e-4 and p["mask_diff_px_outside_tie_band"] == 0 and p["objects_not_matched"] == 0
    # BASELINE's metric is "frames/sec ...; ADD-S delta" (SURVEY.md 8d): |ADD-S(build) - ADD-S(CPU restatement)| <= 1e-4 m on the 1000-point model cloud
    assert p["adds_objects"] >= 8 and p["adds_delta_m"] <= 1e-4 and 1e-3 < p["adds_mean_oracle_m"] < 2e-2
    # the all-physical-cores leg ran when the box has them; its `cores` is what the workers can occupy (the cgroup quota caps it)
    assert p["frames"] == 64
    ac = c["all_cores"]
    assert ac is None or (ac["workers"] > c["cores"] and ac["cores"] == min(ac["workers"], int(c.get("cgroup_cpu_quota") or ac["workers"])))
    m = d["modes"]["f32"]
    assert m["unit"] == "frames/s" and 0 < m["value"] < d["value"] and m["steps"] >= 1
    assert d["ranks_seen"] == [[0, 0, d["ranks_seen"][0][2], d["ranks_seen"][0][3]]] and d["distinct_gpus"] == 1
    assert "one 160x160 detection per frame" in d["config"]["frame_selection"] and d["config"]["candidates_skipped"] >= 0
    # every TIMED step of the overlapped loop re-run on one stream after the timed region: the same bits (a mismatch also fails the run)
    assert p["steps_bitwise_equal"] == "2/2"
    # BASELINE configs[1] and configs[4] ride along as compact legs (the lines of --workload pose / --workload label) ...
    po, la = d["pose"], d["label"]
    assert po["unit"] == "crops/s" and po["steps"] == 10 and "configs[1]" in po["config"]["workload"] and po["value"] > 0
    assert po["parity"]["knn_indices_bit_exact"] is True and po["parity"]["max_dq"] <= 1e-4 and 0.3 < po["knn_training_size"]["frac"] < 1
    assert la["unit"] == "views/s" and la["steps"] == 1 and "configs[4]" in la["config"]["workload"] and "200 synthetic" in la["config"]["workload"]
    assert la["parity"]["max_nn_distance_mm"] < 1e-6 and la["icp"]["point_pairs_per_s"] > 0 and 0 < la["roofline"]["frac"] < 1
    # ... and the LAST object of the line sums the secondary legs up in under 1500 bytes (the driver's record keeps the tail of stdout)
    assert list(d)[-1] == "secondary" and len(json.dumps(d["secondary"])) < 1500
    sec = d["secondary"]
    assert sec["steps_bitwise_equal"] == "2/2" and sec["staged_frames_s"] == st["value"] and sec["f32_frames_s"] == m["value"]
    assert sec["sweep_frames_s"] == sw["value"] and sec["latency_p50_ms"] == lat["p50_ms"]
    assert sec["pose"]["crops_s"] == po["value"] and sec["pose"]["knn_indices_bit_exact"] is True and sec["pose"]["knn_frac_of_fp32_lane_rate"] == po["knn_training_size"]["frac"]
    assert sec["label"]["views_s"] == la["value"] and sec["label"]["max_nn_distance_mm"] == la["parity"]["max_nn_distance_mm"]


def test_frames_1024_line():
    """BASELINE configs[3] at its own size: 1024 frames per step (16 sub-batches of 64 on the one GPU), parity block kept (the oracle
    runs 16 frames of the last sub-batch)"""
    d = _run("--frames", "1024", "--batch", "64", "--steps", "1", "--warmup", "1", "--baseline-frames", "16")
    _common(d, 1, 1)
    assert d["scaling"] == "strong" and d["config"]["frames_per_gpu_per_step"] == 1024 and "configs[3]" in d["config"]["workload"]
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    p = d["parity"]
    assert p["frames"] == 16 and p["max_dq"] <= 1e-4 and p["max_dt"] <= 1e-4 and p["mask_diff_px_outside_tie_band"] == 0 and p["objects_not_matched"] == 0
    assert p["adds_delta_m"] <= 1e-4


def test_mixed_sweep_and_latency_line():
    """`--latency`: one resident frame through the whole path, p50 / p99 of the host clock.  `--mixed`: frames with 1-3 objects and five crop sizes through the bucketed pose stage, as a secondary `sweep` object with its own
    parity block; the primary fields stay what the default command prints"""
    d = _run("--mixed", "--mixed-steps", "2", "--latency", "--latency-runs", "50", "--steps", "2", "--warmup", "2", "--baseline-frames", "16", "--no-modes")
    lat = d["latency"]
    assert lat["runs"] == 50 and lat["objects"] == 1 and 0 < lat["min_ms"] <= lat["p50_ms"] <= lat["p99_ms"] < 50
    _common(d, 2, 2)
    assert "configs[2]" in d["config"]["workload"] and d["config"]["crop_buckets_last_step"] == {"160x160": 64}
    s = d["sweep"]
    assert s["unit"] == "frames/s" and 0 < s["value"] < d["value"] * 1.2 and s["steps"] == 2 and s["frames_per_gpu_per_step"] == 64
    assert s["objects_per_s"] >= s["value"]                           # at least one object per frame
    painted = s["objects_painted_rank0"]
    assert 64 <= painted <= 192 and abs(s["objects_per_step_rank0"] - painted) <= 0.1 * painted
    assert len(s["crop_buckets_last_step"]) >= 4                     # several crop sizes in one step
    assert sum(s["crop_buckets_last_step"].values()) == s["objects_per_step_rank0"]
    p = s["parity"]
    assert p["frames"] == 8 and p["objects_checked"] >= 8
    assert p["max_dq"] <= 1e-4 and p["max_dt"] <= 1e-4 and p["mask_diff_px_outside_tie_band"] == 0 and p["objects_not_matched"] == 0
    assert p["adds_delta_m"] <= 1e-4


def test_label_line():
    d = _run("--workload", "label", "--steps", "1", "--warmup", "1")
    _common(d, 1, 1)
    assert d["unit"] == "views/s" and d["scaling"] == "strong" and d["dtype"] == "f64" and "configs[4]" in d["config"]["workload"]
    assert d["icp"]["registrations_per_s"] > 0 and d["icp"]["point_pairs_per_s"] > 0
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["unit"] == "views/s"
    assert d["parity"]["gpu_points"] == d["parity"]["oracle_points"] and d["parity"]["max_nn_distance_mm"] < 1e-6


def test_pose_line():
    """BASELINE configs[1]: PoseNet + 2 refiner passes + ADD-S through the HIP k-NN / ADD-S kernel on 32 crops; the line prices the pair
    evaluations against the fp32 vector rate, carries the k-NN kernel at the training loss's size, a CPU baseline and the parity block"""
    d = _run("--workload", "pose", "--steps", "3", "--warmup", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity", "knn_training_size"):
        assert k in d, k
    assert d["unit"] == "crops/s" and d["scaling"] == "weak" and d["steps"] == 3 and "configs[1]" in d["config"]["workload"] and "inputs" in d["config"]
    assert abs(d["value"] - 32 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "valu" and r["pairs_per_launch"] == 32 * 1000 * 1000 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert r["kernels"] and all(0 < k["frac"] < 1 for k in r["kernels"])
    kn = d["knn_training_size"]
    assert kn["queries"] == 1000000 and kn["refs"] == 1000 and 0.3 < kn["frac"] < 1
    p = d["parity"]
    assert p["knn_indices_bit_exact"] is True and p["max_dq"] <= 1e-4 and p["max_dt"] <= 1e-4 and p["adds_delta_m"] <= 1e-4
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["unit"] == "crops/s" and d["cpu_baseline"]["value"] > 0
    assert 1e-3 < d["adds_mean_m"] < 2e-2


def test_bench_under_torch_distributed_run_exercises_rccl():
    """the launch line the driver uses for N > 1, here with one rank: init_process_group('nccl'), the per-step all_gather of the poses,
    the barrier + all_reduce(MAX) of the timing"""
    port = 29500 + os.getpid() % 400
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _common(d, 2, 1)
    assert "1 all_gather" in d["config"]["parallelism"]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port + 1), os.path.join(REPO, "bench.py"), "--gpus", "1", "--workload", "label", "--steps", "1",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert d["unit"] == "views/s" and d["n_gpus"] == 1


@pytest.mark.parametrize("flags", [("--steps", "2", "--warmup", "1"), ("--frames", "256", "--steps", "1", "--warmup", "1"),
                                   ("--workload", "label", "--steps", "1", "--warmup", "1"), ("--workload", "pose", "--steps", "2", "--warmup", "1")])
def test_two_rank_rehearsal_on_one_gpu(flags):
    """The driver's N = 2 launch line with both ranks on the one GPU of the box (APE_DIST_BACKEND=gloo: RCCL needs a GPU per rank):
    rank-dependent frames / view shards, the collectives of every step, max-over-ranks timing, rank 0's single line."""
    port = 29900 + os.getpid() % 90
    env = dict(os.environ, APE_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--no-cpu-baseline", *flags],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1                              # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    if "pose" in flags:         # (rank 0's profiled extra step must not call the collective: it hung N > 1 before round 6)
        assert d["unit"] == "crops/s" and d["scaling"] == "weak" and "dp2" in d["config"]["parallelism"]
    elif "--workload" in flags:
        assert d["unit"] == "views/s" and "2 ranks" in d["config"]["parallelism"]
    elif "--frames" in flags:
        assert d["scaling"] == "strong" and d["config"]["frames_per_gpu_per_step"] == 128
    else:
        assert d["scaling"] == "weak" and d["config"]["frames_per_gpu_per_step"] == 64 and d["config"]["objects_found_last_step"] == 64
        assert abs(d["value"] - 2 * 64 * 2 / (d["ms_per_step"] * 2e-3)) < 0.02 * d["value"]
        # every rank re-ran its timed steps on one stream and compared; the compact configs[1] / configs[4] legs ran on both ranks
        assert d["parity"]["steps_bitwise_equal"] == "2/2" and d["pose"]["n_gpus"] == 2 and d["label"]["n_gpus"] == 2 and list(d)[-1] == "secondary"
