"""Frame-batch sharding across the GPUs of a node.

Frames are independent once synchronised, so the only multi-GPU structure is a partition of the
frame index space: contiguous ranges, frame f -> rank floor(f*G/F). No data-path collective exists;
ranks only merge counters (frames decoded, iterations executed, max wall time).
"""


def frame_range(rank, world, total_frames):
    """Half-open range [lo, hi) of global frame indices owned by `rank` (contiguous, balanced)."""
    if world < 1 or not (0 <= rank < world) or total_frames < 0:
        raise ValueError("bad sharding arguments")
    lo = (total_frames * rank) // world
    hi = (total_frames * (rank + 1)) // world
    return lo, hi


def owner_of(frame, world, total_frames):
    """Rank that owns global frame index `frame`."""
    if not (0 <= frame < total_frames):
        raise ValueError("frame out of range")
    r = (frame * world) // total_frames
    while frame < frame_range(r, world, total_frames)[0]:
        r -= 1
    while frame >= frame_range(r, world, total_frames)[1]:
        r += 1
    return r


def merge_counters(local, dist=None, device=None):
    """Sum per-rank counters (dict of numbers) and MAX the 'seconds' entry across ranks.
    `dist` is torch.distributed (already initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local)
    import torch
    keys = sorted(k for k in local if k != "seconds")
    sums = torch.tensor([float(local[k]) for k in keys], dtype=torch.float64, device=device)
    tmax = torch.tensor([float(local.get("seconds", 0.0))], dtype=torch.float64, device=device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    out = {k: float(v) for k, v in zip(keys, sums.tolist())}
    out["seconds"] = float(tmax.item())
    return out
