"""Frame sharding and the result gather of the multi-GPU path.

Frames are independent (reference: `detect(&self)` only reads immutable fields,
src/detector.rs:17-23,505), so a batch shards by frame with no data-path collective; the one
exchange step is the gather of the per-frame results to rank 0.  Works on any torch.distributed
backend: "nccl" (= RCCL over xGMI) with device tensors in bench.py, "gloo" with CPU tensors in
the tests.
"""
import torch
import torch.distributed as dist

SLAB_RECORDS = 1024  # saddle records per frame in the fixed-size result slab (20 KB / frame)


def shard_range(rank, world, frames_per_rank):
    """Global frame indices owned by `rank` (weak scaling: every rank owns frames_per_rank)."""
    lo = rank * frames_per_rank
    return lo, lo + frames_per_rank


def alloc_result_buffers(n_frames, device):
    """(saddles [n_frames*SLAB_RECORDS, 5] f32, table [n_frames, 4] i32: count, offset, status,
    clusters) -- the caller-owned device buffers of agx_saddles_batch_enqueue_to."""
    return (torch.zeros((n_frames * SLAB_RECORDS, 5), dtype=torch.float32, device=device),
            torch.zeros((n_frames, 4), dtype=torch.int32, device=device))


def gather_results(saddles, table, dst=0, group=None):
    """Gather every rank's (saddles, table) to `dst`.  Returns (list_of_saddles, list_of_tables)
    on dst and (None, None) elsewhere.  world_size 1: no communication."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [saddles], [table]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        gs = [torch.empty_like(saddles) for _ in range(world)]
        gt = [torch.empty_like(table) for _ in range(world)]
        dist.gather(table, gt, dst=dst, group=group)
        dist.gather(saddles, gs, dst=dst, group=group)
        return gs, gt
    dist.gather(table, None, dst=dst, group=group)
    dist.gather(saddles, None, dst=dst, group=group)
    return None, None


def unpack_frames(saddles, table):
    """Per-frame saddle arrays (numpy, 5 columns x, y, k, theta, phi) from one rank's buffers;
    a frame with a blocking status bit (1|2|4) yields None."""
    s = saddles.cpu().numpy()
    t = table.cpu().numpy()
    out = []
    for count, offset, status, _ in t:
        out.append(None if (status & 7) else s[offset:offset + count].copy())
    return out
