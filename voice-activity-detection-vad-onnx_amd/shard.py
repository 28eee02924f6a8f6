"""Multi-GPU sharding of the clip batch: one process per GPU, contiguous shards, NO data-path
collective (clips are independent).  The process group is only used for the timing
barrier / max-over-ranks of bench.py and to gather the ragged per-clip results on rank 0.
Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_bounds(n_items, rank, world):
    """Contiguous [start, stop) of rank's shard; shards differ by at most one item."""
    base, extra = divmod(int(n_items), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def init(backend=None, device=None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (no-op for world 1)."""
    rank, local_rank, world = env_rank()
    if world == 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {}
        if backend is None:
            backend = "nccl" if device is not None else "gloo"
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return dist


def fence(dist, sync=None):
    """sync device -> barrier -> sync device (bench.py brackets its timed region with this)."""
    if sync:
        sync()
    if dist is not None:
        dist.barrier()
    if sync:
        sync()


def max_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ok(dist, ok, device="cpu"):
    """True only if EVERY rank says ok (MIN all-reduce).  Any rank-local step that may fail -- a pinned allocation, an OOM -- is
    followed by this before the ranks enter a section with barriers / all-reduces: a rank that bailed out alone would otherwise leave
    the others waiting in the collective until the launcher's timeout."""
    if dist is None:
        return bool(ok)
    import torch
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def gather_ragged(dist, local_results):
    """Per-clip result lists of every rank, concatenated in rank (= clip) order, on every rank."""
    if dist is None:
        return list(local_results)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, list(local_results))
    out = []
    for p in parts:
        out.extend(p)
    return out


def run_sharded(clips, detect_fn, dist=None):
    """Split `clips` (sequence / array, first axis = clip) across ranks, run `detect_fn(shard)` locally,
    return the full per-clip result list on every rank."""
    rank, _, world = env_rank()
    lo, hi = shard_bounds(len(clips), rank, world)
    local = detect_fn(clips[lo:hi]) if hi > lo else []
    return gather_ragged(dist, local)
