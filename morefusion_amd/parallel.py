"""Multi-GPU layout of the path: one process per GPU, scenes sharded
across ranks, no data-path collective until the refined poses are all-gathered once.

ICC couples the objects of ONE scene (every object's no-entry grid is max-ed with the
others' occupancy, contrib/iterative_collision_check_link.py:67-85) but scenes are
independent, and so are the per-object voxelize / 3D-CNN stages: the natural shard unit
is the scene.  The reference has no multi-GPU inference path at all (its only collective
is ChainerMN's gradient all-reduce in train.py:344); the pose all-gather below is the
exchange BASELINE config 4 names: [n_local, 7] float32 per rank (1.8 KB for 8 x 8 objects)
-> one latency-bound ``all_gather_into_tensor`` over RCCL/xGMI at the END of the step,
never per iteration.
"""
import time

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous block partition of ``range(n_items)``; the first ``n_items % world``
    ranks get one extra item.  Returns (start, stop)."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_poses(local_poses, group=None):
    """local_poses [n_local, 7] (q wxyz, t) -> [sum n_local, 7] on every rank, rank order.

    Equal shard sizes use a single ``all_gather_into_tensor``; ragged shards are padded
    to the maximum (the counts travel in the same call as an extra column)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_poses
    world = dist.get_world_size(group)
    n_local = torch.tensor([local_poses.shape[0]], device=local_poses.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    padded = local_poses.new_zeros((n_max, local_poses.shape[1]))
    padded[: local_poses.shape[0]] = local_poses
    out = local_poses.new_empty((world * n_max, local_poses.shape[1]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == n_max for c in counts):
        return out
    return torch.cat([out[r * n_max: r * n_max + c] for r, c in enumerate(counts)], dim=0)


def all_gather_poses_equal(local_poses, out=None, group=None):
    """Fast path for equal shards (the benchmark's weak-scaling layout): one collective,
    no host synchronisation, optional preallocated output."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_poses
    world = dist.get_world_size(group)
    if out is None:
        out = local_poses.new_empty((world * local_poses.shape[0], local_poses.shape[1]))
    dist.all_gather_into_tensor(out, local_poses.contiguous(), group=group)
    return out


def timed_steps(step, steps, warmup, device=None, group=None):
    """The benchmark's timing contract: ``warmup`` untimed calls of ``step()``, then exactly
    ``steps`` calls bracketed by (device sync + barrier) on both sides; returns the elapsed
    seconds of the SLOWEST rank (all-reduce MAX), identical on every rank."""
    cuda = device is not None and torch.device(device).type == "cuda"
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def fence():
        if cuda:
            torch.cuda.synchronize(device)
        if multi:
            dist.barrier(group=group)
        if cuda:
            torch.cuda.synchronize(device)

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if cuda else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(tt.item())
    return elapsed
