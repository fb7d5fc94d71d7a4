"""Multi-GPU: realisations shard embarrassingly, one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for the tests).

There is no exchange step inside the path: every rank owns a contiguous range of the global
realisation index and draws from the same counter-based stream, so the ensemble is bit-identical for any number
of GPUs.  The only collective is the final gather of the residual arrays to rank 0 named by BASELINE.json's
north_star; it is chunked so that rank 0 never needs more than its own slice plus the full result, and can be
skipped entirely when each rank writes its own shard.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank=None, world=None):
    """[start, stop) of the realisation indices owned by `rank`: contiguous, sizes differ by at most one."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def generate_sharded(engine, total, r0=0):
    """this rank's shard of realisations r0 .. r0+total-1 -> (local tensor [R_loc, n_toa], (start, stop))."""
    start, stop = shard_range(total)
    return engine.generate(stop - start, r0=r0 + start), (start, stop)


def gather_to_rank0(local, total=None, dst=0):
    """Gather row-sharded [R_loc, n] tensors to rank `dst` in global row order.  Returns the [total, n] tensor on
    `dst`, None elsewhere.  Shards may differ in size by one row (shard_range); they are padded for the
    collective and trimmed after it."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if total is None:
        t = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        dist.all_reduce(t)
        total = int(t.item())
    sizes = [shard_range(total, r, world) for r in range(world)]
    rows = max(b - a for a, b in sizes)
    assert local.shape[0] == sizes[rank][1] - sizes[rank][0], "local shard does not match shard_range()"
    send = local
    if local.shape[0] != rows:
        send = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:sizes[r][1] - sizes[r][0]] for r in range(world)], dim=0)
