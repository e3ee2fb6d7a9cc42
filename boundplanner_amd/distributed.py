"""Multi-GPU sharding of a batch of independent MPC solves.

Instances are independent, so the batch is cut into contiguous blocks, one per rank (one process
per GPU); there is NO collective on the data path.  The only exchange is the all-gather of the
solutions that BASELINE.json's north_star asks for (RCCL over xGMI when the tensors are on GPUs,
gloo in the CPU tests).  1-GPU and G-GPU results are bitwise identical because an instance's
result does not depend on the batch it is solved in.
"""
import numpy as np


def shard_bounds(B, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; the first B % world ranks get one extra."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def solve_sharded(solve_fn, x0, lbx, ubx, p, dist=None, device=None):
    """Solve this rank's block with `solve_fn(x0, lbx, ubx, p) -> dict(x, f, iters, status, viol)`
    and all-gather every output so that each rank holds the full batch (torch.distributed `dist`
    already initialised; None = single process)."""
    B = x0.shape[0]
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = shard_bounds(B, world, rank)
    r = solve_fn(x0[lo:hi], lbx[lo:hi], ubx[lo:hi], p[lo:hi])
    if world == 1:
        return r
    import torch
    out = {}
    sizes = [shard_bounds(B, world, k) for k in range(world)]
    maxn = max(h - l for l, h in sizes)
    for key in ("x", "f", "iters", "status", "viol"):
        a = np.asarray(r[key])
        pad = np.zeros((maxn,) + a.shape[1:], dtype=a.dtype)
        pad[: hi - lo] = a
        t = torch.from_numpy(pad)
        if device is not None:
            t = t.to(device)
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)
        out[key] = np.concatenate([b.cpu().numpy()[: h - l] for b, (l, h) in zip(bufs, sizes)])
    return out
