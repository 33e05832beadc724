"""Frame sharding across the GPUs of one node: frames are independent, so each rank (one process per GPU)
owns a contiguous block and the only collective of the live path is ONE all_gather of the per-frame results
`[frames_per_rank, max_obj, 8] f32 = (cls, qw,qx,qy,qz, tx,ty,tz)` -- a few KB, latency-bound on any xGMI link.
Works with any torch.distributed backend (RCCL on the GPU box, gloo in the CPU tests)."""
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
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local.contiguous())
    return torch.cat(parts, 0)
