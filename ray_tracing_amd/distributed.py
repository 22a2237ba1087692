"""Multi-GPU rendering: one process per GPU, samples sharded by global sample index, one sum-reduce
of the per-GPU radiance accumulators to rank 0 (RCCL over xGMI when the tensors are on GPUs; the
same code runs over gloo on CPU tensors in the tests).

The reference has no counterpart (single thread, path_integrator.rs:286-308).  Because every random
draw is addressed by (pixel, GLOBAL sample index, bounce, draw), the set of paths rendered by N
ranks is exactly the set one rank would render; only the f32 summation order across ranks differs.
"""
import torch
import torch.distributed as dist


def sample_shard(spp_total, world_size, rank):
    """Contiguous block of global sample indices for `rank`: (first, count).  Ragged when
    spp_total % world_size != 0 (the first ranks get one more)."""
    base, extra = divmod(spp_total, world_size)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def reduce_accumulators(acc, dst=0):
    """Sum the per-rank [h, w, 3] f32 accumulators onto `dst` (in place on dst)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
    return acc


def render_sharded(render_fn, width, height, spp_total, depth, seed, device):
    """render_fn(spp_first, spp_count) -> [h, w, 3] f32 tensor of radiance sums on `device`.
    Returns the frame sum on rank 0 (other ranks: their reduced-away partial)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    first, count = sample_shard(spp_total, world, rank)
    if count > 0:
        acc = render_fn(first, count)
    else:
        acc = torch.zeros(height, width, 3, dtype=torch.float32, device=device)
    return reduce_accumulators(acc, 0)
