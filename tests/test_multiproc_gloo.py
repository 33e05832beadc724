"""N>1 path on CPU: two gloo ranks shard 10 frames, build their result blocks and gather them once."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from autoposeestimation_amd.sharding import gather_results, pack_results, shard_range


def _fake_pose(frame):
    return torch.tensor([1.0, 0.0, 0.0, 0.0, 0.1 * frame, -0.2 * frame, 0.5 + frame], dtype=torch.float64)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per_rank = 5
    lo, hi = rank * per_rank, (rank + 1) * per_rank
    objects = [(f - lo, 1 + f % 3, 0, 0, 0, 0) for f in range(lo, hi) if f != 7]     # frame 7 has no detection
    pose = torch.stack([_fake_pose(lo + o[0]) for o in objects])
    local = pack_results(per_rank, objects, pose, max_obj=1)
    full = gather_results(local, dist)
    if rank == 0:
        q.put(full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert full.shape == (10, 1, 8)
    for f in range(10):
        if f == 7:
            assert not full[f].any()
        else:
            assert full[f, 0, 0] == 1 + f % 3
            assert torch.allclose(full[f, 0, 1:], _fake_pose(f).float())


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 1024, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


# ---- label generation (BASELINE configs[4]): chains owned by one rank, per-view work sharded, one padded all-gather ------------------
def _views(n):
    """stand-ins for the pre-processed surfaces of n views: ragged float64 point sets, one of them empty"""
    import numpy as np
    rng = np.random.default_rng(5)
    return [rng.standard_normal((0 if i == 3 else 40 + 17 * i, 3)) * 50 for i in range(n)]


def _order_dependent_fuse(sets):
    """a fusion whose result depends on the ORDER of the sets like the sequential ICP accumulation does (create_pointcloud.py:288-312)"""
    import numpy as np
    acc = np.zeros(3)
    for k, s in enumerate(sets):
        s = s.numpy() if hasattr(s, "numpy") else np.asarray(s)
        if len(s):
            acc = acc * 0.5 + s.sum(0) * (k + 1)
    return acc


def _label_worker(rank, world, port, q):
    import numpy as np
    from autoposeestimation_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    views = _views(7)
    made = []

    def make_set(v):
        made.append(len(v))
        return torch.from_numpy(v * 2.0)             # the "get_surface" of a view (any deterministic per-view function)

    res = {}
    for chain, owner in ((0, sharding.chain_owner(0, world)), (1, sharding.chain_owner(1, world))):
        res[chain] = sharding.sharded_chain(views, make_set, _order_dependent_fuse, owner, dist)
    sets = sharding.gather_point_sets([(i, views[i]) for i in range(7) if i % world == rank][:4], 7, dist) if world == 2 else None
    n_single = len(made)
    # three chains of different length at once: one all-gather, every rank fuses what it owns
    multi = sharding.sharded_chains([views[:3], views[3:], views[1:6]], make_set, _order_dependent_fuse, dist)
    res["multi"] = np.array([[k] + v.tolist() for k, v in sorted(multi.items())])
    res["multi_made"] = np.array([len(made) - n_single])
    q.put((rank, {k: (None if v is None else v.tolist()) for k, v in res.items()}, n_single, [len(s) for s in sets]))
    dist.barrier()
    dist.destroy_process_group()


def test_label_chain_sharding_two_ranks_equals_single_rank():
    import numpy as np
    from autoposeestimation_amd import sharding
    views = _views(7)
    want = _order_dependent_fuse([v * 2.0 for v in views])
    single = sharding.sharded_chain(views, lambda v: torch.from_numpy(v * 2.0), _order_dependent_fuse, 0, None)
    assert np.array_equal(single, want)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29850 + os.getpid() % 100
    procs = [ctx.Process(target=_label_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (res, made, lens)) for r, res, made, lens in (q.get(), q.get()))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # chain 0 belongs to rank 0, chain 1 to rank 1; the owner's result is bit-identical to the single-rank chain, the other rank has none
    assert got[0][0][0] == want.tolist() and got[0][0][1] is None
    assert got[1][0][1] == want.tolist() and got[1][0][0] is None
    # each rank pre-processed only its share of the views, per chain: 4 + 3 of 7
    assert got[0][1] == 2 * 4 and got[1][1] == 2 * 3
    # the gather returns every set, in global order, with its true (ragged) length on both ranks
    assert got[0][2] == got[1][2] == [len(v) for v in views]
    # sharded_chains: chains 0 and 2 on rank 0, chain 1 on rank 1, each equal to its single-rank fusion; 12 views split 6 + 6
    chains = [views[:3], views[3:], views[1:6]]
    want_multi = [_order_dependent_fuse([v * 2.0 for v in ch]).tolist() for ch in chains]
    assert got[0][0]["multi"] == [[0.0] + want_multi[0], [2.0] + want_multi[2]]
    assert got[1][0]["multi"] == [[1.0] + want_multi[1]]
    assert got[0][0]["multi_made"] == [6] and got[1][0]["multi_made"] == [6]
    solo = sharding.sharded_chains(chains, lambda v: torch.from_numpy(v * 2.0), _order_dependent_fuse, None)
    assert [solo[i].tolist() for i in range(3)] == want_multi


def test_prefetched_keeps_order_and_propagates_errors():
    import time
    import pytest
    from autoposeestimation_amd.sharding import prefetched, run_side_by_side

    def load(x):
        time.sleep(0.002 * (7 - x % 7))                  # later items finish first
        if x == 41:
            raise KeyError("bad item")
        return x * x

    assert list(prefetched(list(range(40)), load, workers=6, window=9)) == [x * x for x in range(40)]
    assert list(prefetched([3, 1, 2], None)) == [3, 1, 2]
    assert list(prefetched([], load)) == []
    with pytest.raises(KeyError):
        list(prefetched(list(range(36, 45)), load, workers=3, window=4))
    # without a GPU (this test), or for one job, the jobs simply run in order
    assert run_side_by_side([lambda i=i: i + 1 for i in range(5)]) == [1, 2, 3, 4, 5]
    assert run_side_by_side([]) == []


# ---- more ranks than two, uneven shards (CPU rehearsal of the 8-GPU node: gloo, tiny clouds) ------------------------------------------
def _spawn(worker, world, port_base, *args, timeout=120):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = port_base + os.getpid() % 150
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get() for _ in range(world)]
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0
    return dict(got)


def _uneven_worker(rank, world, port, q, n_frames, chain_lens):
    import numpy as np
    from autoposeestimation_amd import sharding
    from autoposeestimation_amd.experiments.eval import merge_results
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # frames: this rank's (uneven) share, padded to the largest share for the single equally-shaped all_gather
    lo, hi = shard_range(n_frames, rank, world)
    per_rank = -(-n_frames // world)
    objects = [(f - lo, 1 + f % 3, 0, 0, 0, 0) for f in range(lo, hi) if f % 5 != 4]
    pose = torch.stack([_fake_pose(lo + o[0]) for o in objects]) if objects else torch.zeros(0, 7, dtype=torch.float64)
    full = gather_results(pack_results(per_rank, objects, pose, max_obj=1), dist)
    # label chains of different lengths: per-view work over all ranks, each surface to its chain's owner only
    views = _views(sum(chain_lens))
    chains, pos = [], 0
    for n in chain_lens:
        chains.append(views[pos:pos + n])
        pos += n
    made = []

    def make_set(v):
        made.append(len(v))
        return torch.from_numpy(v * 2.0)

    multi = sharding.sharded_chains(chains, make_set, _order_dependent_fuse, dist)
    # owner-only exchange: set gi goes to rank (gi * 3) % world and to nobody else
    n_sets = len(views)
    slo, shi = shard_range(n_sets, rank, world)
    owners = [(gi * 3) % world for gi in range(n_sets)]
    got_sets = sharding.gather_point_sets([(gi, views[gi]) for gi in range(slo, shi)], n_sets, dist, owners=owners)
    sets_ok = all((s is None) == (owners[gi] != rank) and (s is None or np.array_equal(s.numpy(), views[gi])) for gi, s in enumerate(got_sets))
    # ADD-S evaluation: every rank's per-class buckets -> the all-reduced (sum dis, n < 2 cm, n) table
    classes = ["a", "b", "c"]
    local = {c: {"<2": 0, ">=2": 0, "dis": []} for c in classes}
    elo, ehi = shard_range(29, rank, world)
    for j in range(elo, ehi):
        d = 0.001 * (1 + (j * 7) % 40)
        local[classes[j % 3]]["dis"].append(d)
        local[classes[j % 3]]["<2" if d < 0.02 else ">=2"] += 1
    merged = merge_results(local, classes, dist)
    q.put((rank, {"full": full, "multi": {k: v.tolist() for k, v in multi.items()}, "made": len(made), "sets_ok": sets_ok,
                  "merged": {c: (merged[c]["<2"], merged[c][">=2"], float(merged[c]["dis"]), float(merged[c]["p"]), merged[c]["dis_all"]) for c in classes}}))
    dist.barrier()
    dist.destroy_process_group()


def _check_uneven(world, n_frames, chain_lens, port_base):
    import numpy as np
    from autoposeestimation_amd import sharding
    from autoposeestimation_amd.experiments.eval import merge_results
    got = _spawn(_uneven_worker, world, port_base, n_frames, chain_lens)
    per_rank = -(-n_frames // world)
    for r in range(world):
        full = got[r]["full"]
        assert full.shape == (world * per_rank, 1, 8)
        for rr in range(world):
            lo, hi = shard_range(n_frames, rr, world)
            for f in range(lo, hi):
                row = full[rr * per_rank + f - lo, 0]
                if f % 5 == 4:
                    assert not row.any()
                else:
                    assert row[0] == 1 + f % 3 and torch.allclose(row[1:], _fake_pose(f).float())
            assert not full[rr * per_rank + hi - lo:(rr + 1) * per_rank].any()        # the padding of a short share stays empty
        assert got[r]["sets_ok"]
    views = _views(sum(chain_lens))
    chains, pos = [], 0
    for n in chain_lens:
        chains.append(views[pos:pos + n])
        pos += n
    want = [_order_dependent_fuse([v * 2.0 for v in ch]).tolist() for ch in chains]
    owned = {}
    for r in range(world):
        for ci, v in got[r]["multi"].items():
            assert sharding.chain_owner(ci, world) == r and ci not in owned
            owned[ci] = v
    assert [owned[ci] for ci in range(len(chains))] == want                          # every chain fused exactly once, bit-identical
    assert sum(got[r]["made"] for r in range(world)) == len(views)                   # every view pre-processed exactly once
    assert max(got[r]["made"] for r in range(world)) - min(got[r]["made"] for r in range(world)) <= 1
    classes = ["a", "b", "c"]
    local = {c: {"<2": 0, ">=2": 0, "dis": []} for c in classes}
    for j in range(29):
        d = 0.001 * (1 + (j * 7) % 40)
        local[classes[j % 3]]["dis"].append(d)
        local[classes[j % 3]]["<2" if d < 0.02 else ">=2"] += 1
    single = merge_results(local, classes, None)
    for r in range(world):
        for c in classes:
            less, more, dis, p, dis_all = got[r]["merged"][c]
            assert (less, more) == (single[c]["<2"], single[c][">=2"]) and p == float(single[c]["p"])
            assert abs(dis - float(single[c]["dis"])) <= 1e-5 and sorted(dis_all) == sorted(single[c]["dis_all"])
            assert np.isclose(np.mean(dis_all), np.mean(single[c]["dis_all"]))


def test_three_ranks_uneven_frames_views_and_chains():
    _check_uneven(3, 10, [3, 4, 2, 5], 30100)          # 10 frames over 3 ranks (4 + 3 + 3), 14 views (5 + 5 + 4), 4 chains (2 + 1 + 1)


def test_eight_ranks_uneven_frames_views_and_chains():
    _check_uneven(8, 13, [2, 1, 3, 2, 1, 2, 3, 1, 2, 2], 30300)      # fewer frames than 2 per rank, 19 views, 10 chains over 8 ranks


def _failing_worker(rank, world, port, q):
    from autoposeestimation_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    views = _views(6)

    def make_set(v):
        if rank == 1:
            raise KeyError("bad view on rank 1")
        return torch.from_numpy(v)

    try:
        sharding.sharded_chains([views[:3], views[3:]], make_set, _order_dependent_fuse, dist)
        outcome = "returned"
    except KeyError:
        outcome = "own error"
    except RuntimeError as e:
        outcome = "told: " + str(e)
    q.put((rank, outcome))
    dist.barrier()                      # every rank is still alive and in step: nobody is stuck in the gather
    dist.destroy_process_group()


def test_a_failure_on_one_rank_is_raised_on_every_rank():
    got = _spawn(_failing_worker, 3, 30500)
    assert got[1] == "own error"
    for r in (0, 2):
        assert got[r].startswith("told:") and "rank(s) [1]" in got[r]


def _failing_eval_worker(rank, world, port, q):
    from autoposeestimation_amd.experiments import eval as EV
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class OneBadSample:                  # stands in for PoseDataset("test", ...): one sample, unreadable
        def __init__(self, *a, **k):
            pass

        def __len__(self):
            return 1

        def __getitem__(self, j):
            raise KeyError("corrupt sample %d" % j)

        def get_num_points_mesh(self):
            return 500

        def get_sym_list(self):
            return []

    EV.PoseDataset = OneBadSample
    try:
        EV.eval(500, False, "set", False, "auto", 0.0, 0.0, None, 0.015, None, 0, 0, ["a"], dist=dist)
        outcome = "returned"
    except KeyError:
        outcome = "own error"
    except RuntimeError as e:
        outcome = "told: " + str(e)
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_eval_shard_is_raised_on_every_rank():
    """experiments/eval.py: the rank holding the unreadable sample raises its own error, the rank with an empty shard is told instead of
    waiting in merge_results' all_reduce"""
    got = _spawn(_failing_eval_worker, 2, 30700)
    owner = 0 if shard_range(1, 0, 2) == (0, 1) else 1
    assert got[owner] == "own error"
    assert got[1 - owner].startswith("told:") and "rank(s) [%d]" % owner in got[1 - owner]
