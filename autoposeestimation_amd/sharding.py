"""Frame sharding across the GPUs of one node: frames are independent, so each rank (one process per GPU)
owns a contiguous block and the only collective of the live path is ONE all_gather of the per-frame results
`[frames_per_rank, max_obj, 8] f32 = (cls, qw,qx,qy,qz, tx,ty,tz)` -- a few KB, latency-bound on any xGMI link.
Works with any torch.distributed backend (RCCL on the GPU box, gloo in the CPU tests).

Label generation (BASELINE configs[4]): inside one (object, rotation-directory) CHAIN the fusion is sequential and
order-dependent (create_pointcloud.py:288-312), so a chain is owned by ONE rank (`chain_owner`); what shards inside a chain is the
per-view work (decode, `get_surface` incl. its filters): every rank pre-processes its `shard_range` of the chain's views and ONE
grouped exchange (`gather_point_sets(owners=...)`: a small all-gather of the counts, then packed point-to-point messages) hands the
variable-length surfaces to the rank that owns the chain -- and to no other -- which registers them in view order: bit-identical to
the single-rank chain because the surfaces are the same arrays in the same order.  Every per-rank stage ends in `all_ranks_ok`, so
an exception on one rank is raised on all of them instead of leaving the others blocked in the next collective."""
import os

import numpy as np
import torch


def shard_range(n_items, rank, world):
    """Contiguous block of `n_items` for `rank`; the first `n_items % world` ranks take one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def pack_results(n_frames, objects, pose, max_obj=1, device=None):
    """objects: [(frame_local, cls, ...)], pose[n,7] -> [n_frames, max_obj, 8] f32, cls = 0 marks an empty slot.
    Objects beyond `max_obj` per frame are dropped (highest slots first come first served in detection order)."""
    device = pose.device if device is None else device
    out = torch.zeros(n_frames, max_obj, 8, dtype=torch.float32, device=device)
    used = [0] * n_frames
    p32 = pose.to(torch.float32)
    for i, o in enumerate(objects):
        f = o[0]
        if used[f] < max_obj:
            out[f, used[f], 0] = float(o[1])
            out[f, used[f], 1:] = p32[i]
            used[f] += 1
    return out


def gather_results(local, dist=None):
    """all_gather of equally-shaped per-rank result blocks -> [world * frames_per_rank, max_obj, 8] in rank order."""
    if dist is None or not dist.is_initialized():
        return local
    if dist.get_backend() != "nccl" and local.is_cuda:            # gloo gathers host tensors (CPU tests, single-GPU rehearsals)
        host = local.contiguous().cpu()
        parts = [torch.empty_like(host) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, host)
        return torch.cat(parts, 0).to(local.device)
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local.contiguous())
    return torch.cat(parts, 0)


def chain_owner(chain_index, world):
    """rank that owns chain number `chain_index` of an (object x direction) enumeration: round robin"""
    return chain_index % max(1, world)


def _dist_device(dist):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def all_ranks_ok(dist, error=None, what="a collective stage"):
    """Make a per-rank failure COLLECTIVE: every rank passes `error` (the exception it caught, or None), one tiny all-reduce (MAX) of
    the failure flags, and if any rank failed EVERY rank raises -- the failing ranks their own exception, the others a RuntimeError
    naming the ranks.  Without it a rank that raises before a collective leaves the others blocked in that collective for ever."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if error is not None:
            raise error
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    flags = torch.zeros(world, dtype=torch.int32, device=_dist_device(dist))
    if error is not None:
        flags[rank] = 1
    dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    failed = [r for r in range(world) if int(flags[r])]
    if error is not None:
        raise error
    if failed:
        raise RuntimeError("%s failed on rank(s) %s; rank %d stops with them" % (what, failed, rank))


def guarded(dist, fn, what="a collective stage"):
    """fn() on this rank, then `all_ranks_ok`: either every rank returns its result or every rank raises."""
    try:
        out, err = fn(), None
    except Exception as e:      # noqa: BLE001 -- re-raised on this rank by all_ranks_ok, announced to the others
        out, err = None, e
    all_ranks_ok(dist, err, what)
    return out


def gather_point_sets(local_sets, n_total, dist=None, owners=None):
    """local_sets: [(global_index, points[n_i, 3] float64 tensor or ndarray)] this rank produced; n_total: number of sets over all
    ranks (each rank holds at most ceil(n_total / world) of them, cf. shard_range).  Returns the list of the n_total point arrays
    (float64 tensors on the collective's device) in global-index order.

    `owners` = None: every rank receives every set (one small all_gather of the (index, count) table and ONE all_gather of the
    surfaces padded to [sets_per_rank, Pmax, 3]).
    `owners[gi]` = the rank that consumes set gi ("to the rank owning the chain"): after the table all_gather each
    producer sends, per destination rank, ONE packed `[sum of counts, 3]` float64 message with the sets that rank owns (grouped
    point-to-point: `batch_isend_irecv`, at most world - 1 sends and world - 1 receives per rank, nothing padded, nothing sent
    to ranks that do not need it); entries of the result that this rank does not own are None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = [None] * n_total
        for gi, pts in local_sets:
            out[gi] = torch.as_tensor(pts, dtype=torch.float64)
        return out
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _dist_device(dist)
    per_rank = -(-n_total // world)
    if len(local_sets) > per_rank:
        raise ValueError("a rank holds %d point sets, more than ceil(%d / %d)" % (len(local_sets), n_total, world))
    table = torch.full((per_rank, 2), -1, dtype=torch.int64, device=dev)
    for k, (gi, pts) in enumerate(local_sets):
        table[k, 0], table[k, 1] = gi, len(pts)
    tables = [torch.empty_like(table) for _ in range(world)]
    dist.all_gather(tables, table)
    tables = torch.stack(tables).cpu().numpy()                       # [world, per_rank, 2]
    out = [None] * n_total
    if owners is not None:
        if len(owners) != n_total:
            raise ValueError("owners must name a rank for each of the %d sets" % n_total)
        mine = {gi: torch.as_tensor(pts, dtype=torch.float64).to(dev).reshape(-1, 3) for gi, pts in local_sets}
        ops, recv_bufs, keep = [], {}, []
        for dst in range(world):                                     # what this rank produced and `dst` consumes, in table order
            if dst == rank:
                continue
            part = [mine[int(gi)] for gi, _ in tables[rank] if gi >= 0 and owners[int(gi)] == dst]
            rows = sum(len(p) for p in part)
            if rows:
                msg = torch.cat(part, 0).contiguous()
                keep.append(msg)
                ops.append(dist.P2POp(dist.isend, msg, dst))
        for src in range(world):                                     # what `src` produced and this rank consumes
            if src == rank:
                continue
            rows = sum(int(cnt) for gi, cnt in tables[src] if gi >= 0 and owners[int(gi)] == rank)
            if rows:
                recv_bufs[src] = torch.empty(rows, 3, dtype=torch.float64, device=dev)
                ops.append(dist.P2POp(dist.irecv, recv_bufs[src], src))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        seen = set()
        for src in range(world):
            pos = 0
            for gi, cnt in tables[src]:
                gi, cnt = int(gi), int(cnt)
                if gi < 0:
                    continue
                seen.add(gi)
                if owners[gi] != rank:
                    continue
                if src == rank:
                    out[gi] = mine[gi].clone()
                else:
                    out[gi] = recv_bufs[src][pos:pos + cnt].clone() if cnt else torch.empty(0, 3, dtype=torch.float64, device=dev)
                    pos += cnt
        if len(seen) != n_total:
            raise RuntimeError("gather_point_sets: some of the %d sets were produced by no rank" % n_total)
        return out
    pmax = max(1, int(tables[..., 1].max()))
    buf = torch.zeros(per_rank, pmax, 3, dtype=torch.float64, device=dev)
    for k, (gi, pts) in enumerate(local_sets):
        buf[k, :len(pts)] = torch.as_tensor(pts, dtype=torch.float64).to(dev)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    for r in range(world):
        for k in range(per_rank):
            gi, cnt = int(tables[r, k, 0]), int(tables[r, k, 1])
            if gi >= 0:
                out[gi] = bufs[r][k, :cnt].clone()
    if any(o is None for o in out):
        raise RuntimeError("gather_point_sets: some of the %d sets were produced by no rank" % n_total)
    return out


def prefetched(items, load, workers=8, window=32):
    """`load(item)` for every item, in order, computed up to `window` items ahead on `workers` host threads (PNG decode and file reads
    release the GIL): the host-side decode of the next views runs while the GPU works on the current one."""
    if load is None:
        yield from items
        return
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as ex:
        pending, it = deque(), iter(items)
        for item in it:
            pending.append(ex.submit(load, item))
            if len(pending) >= window:
                break
        while pending:
            out = pending.popleft().result()
            for item in it:
                pending.append(ex.submit(load, item))
                break
            yield out


def run_side_by_side(jobs, workers=None):
    """[job() for job in jobs] with up to `workers` of them in flight, each on its own host thread AND its own HIP stream: the label
    path's units (views, chains) are chains of tiny dependent kernels with a host read-back here and there -- one of them leaves the
    GPU idle most of the time, several side by side fill it (ctypes calls and the blocking copies release the GIL).  Every job starts
    after the caller's stream (its inputs are ready) and is synchronised before its result is handed back; results keep job order.
    Falls back to a plain loop without a GPU (gloo tests) or for a single job."""
    if workers is None:
        workers = int(os.environ.get("APE_SIDE_WORKERS", "3"))     # measured on one MI355X (200-view label bench): 1: 429, 2: 524, 3: 588, 4: 459 views/s
                                                                   # -- the jobs' Python halves share the GIL, more threads only contend
    if len(jobs) <= 1 or workers <= 1 or not torch.cuda.is_available():
        return [job() for job in jobs]
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.cuda.current_device()
    parent = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(parent)

    def run(job):
        torch.cuda.set_device(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_event(ready)
        with torch.cuda.stream(side):
            out = job()
        side.synchronize()
        return out

    with ThreadPoolExecutor(max_workers=min(workers, len(jobs))) as ex:
        return list(ex.map(run, jobs))


def sharded_chain(items, make_set, fuse, owner, dist=None, load=None):
    """One chain: `items` (its views, in fusion order) -> make_set(item) for this rank's shard_range of them -> one padded all-gather
    -> `fuse(list of point arrays in item order)` on the owner rank only (None elsewhere).  `load`: optional host-side stage
    (decode) run ahead on threads, make_set then receives load(item)."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    lo, hi = shard_range(len(items), rank, world)

    def produce():
        loaded = list(prefetched(items[lo:hi], load))
        return list(zip(range(lo, hi), run_side_by_side([lambda x=x: make_set(x) for x in loaded])))

    local = guarded(dist, produce, "the per-view stage of a sharded chain")         # a rank that fails here must not strand the others in the gather
    sets = gather_point_sets(local, len(items), dist, owners=[owner] * len(items))
    return guarded(dist, lambda: fuse(sets) if rank == owner else None, "the fusion of a sharded chain")


def sharded_chains(chains, make_set, fuse, dist=None, load=None, make_sets=None, fuse_many=None):
    """Several chains at once: `chains` = list of item lists.  The per-item work of ALL chains is spread over the ranks as one flat
    list (balanced even when chains differ in length), ONE padded all-gather distributes the sets, and then every rank fuses the
    chains it owns (`chain_owner`) -- the sequential, order-dependent parts of different chains run side by side on different GPUs,
    which a loop of `sharded_chain` calls cannot do (each of its all-gathers would wait for the previous chain's owner).
    Returns {chain index: fuse(sets of that chain)} for the chains this rank owns.

    `make_sets(list of loaded items) -> list of sets` and `fuse_many(list of set lists) -> list of results`: optional BATCHED forms of the
    two stages (pc_reconstruction/batched.py: one launch advances many clouds); when given they replace the thread-per-unit form."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    flat = [(ci, item) for ci, items in enumerate(chains) for item in items]
    lo, hi = shard_range(len(flat), rank, world)

    def produce():
        loaded = list(prefetched([f[1] for f in flat[lo:hi]], load))
        if make_sets is not None:
            return list(zip(range(lo, hi), make_sets(loaded)))
        return list(zip(range(lo, hi), run_side_by_side([lambda x=x: make_set(x) for x in loaded])))

    local = guarded(dist, produce, "the per-view stage of the sharded chains")
    sets = gather_point_sets(local, len(flat), dist, owners=[chain_owner(ci, world) for ci, _ in flat])     # each surface to its chain's owner only
    mine, pos = [], 0
    for ci, items in enumerate(chains):
        if chain_owner(ci, world) == rank:
            mine.append((ci, sets[pos:pos + len(items)]))
        pos += len(items)
    fused = guarded(dist, lambda: fuse_many([part for _, part in mine]) if fuse_many is not None else
                    run_side_by_side([lambda part=part: fuse(part) for _, part in mine]),       # this rank's chains, in lock step / side by side
                    "the fusion stage of the sharded chains")
    return {ci: res for (ci, _), res in zip(mine, fused)}
