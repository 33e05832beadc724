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
