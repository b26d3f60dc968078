"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" for the CPU tests).

The path shards by channel: channels are independent, the reference itself runs
one process per channel.  No collective takes part in decoding; the only exchange is the final
gather of decoded blocks to the collecting rank, in the 40-byte wire form (tg_layout.h), each
peer -> root transfer riding its own xGMI link.  The product's gather is the C-ABI entry
tgpu_comm_gather() (csrc/tg_comm.c, binding.Comm); gather_wire() below is the torch.distributed form
the CPU tests (gloo) and bench.py --torch-gather use.
"""
import torch
import torch.distributed as dist


def shard_channels(nchan, rank, world):
    """contiguous, balanced channel-major shard: channels [lo, hi) belong to `rank`"""
    base, extra = divmod(nchan, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_wire(local, dst=0, group=None, async_op=False, out=None):
    """Gather equally sized wire-record tensors to `dst`.

    local: uint8 tensor (nslots * 40,) on this rank's device (CUDA for nccl, CPU for gloo).
    Returns (list_of_tensors_or_None, work_or_None); the list is only filled on `dst`
    (index = source rank, i.e. channel-shard order)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        if out is None:
            out = [torch.empty_like(local) for _ in range(world)]
    else:
        out = None
    work = dist.gather(local, gather_list=out, dst=dst, group=group, async_op=async_op)
    return out, work


def gather_compact(local, nbytes, dst=0, group=None):
    """Gather the ranks' compact buffers (csrc/tg_cwire.h: a different size per rank and step) to `dst`.

    local: uint8 tensor holding this rank's buffer in its first nbytes bytes.  The sizes go round the ranks first
    (all_gather of one int64: the job's control plane), then every rank sends its buffer padded to the largest --
    torch.distributed.gather wants one size; the product's tgpu_comm_gatherv() sends exact sizes.  Returns
    (sizes, list of tensors trimmed to their size) on `dst`, (sizes, None) elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    szs = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(szs, torch.tensor([int(nbytes)], dtype=torch.int64), group=group)
    sizes = [int(x.item()) for x in szs]
    m = (max(sizes) + 15) & ~15
    send = torch.zeros(m, dtype=torch.uint8, device=local.device)
    send[:nbytes] = local[:nbytes]
    out = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, gather_list=out, dst=dst, group=group)
    return sizes, ([o[:n] for o, n in zip(out, sizes)] if rank == dst else None)
