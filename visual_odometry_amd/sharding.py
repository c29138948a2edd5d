"""How the hot path is partitioned over the GPUs of one node.

Frame pairs are independent units: rank r takes a contiguous block of pairs and no
collective is needed.  A very large descriptor match shards the *query rows* of the
distance matrix (the train set is replicated), so every rank's top-2 table is complete
for its rows and one gather assembles the whole table.
"""
import torch
import torch.distributed as dist


def block_range(n, rank, world):
    """Contiguous [begin, end) of n units owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_rows(local, n_total, dst=0):
    """Gathers the ranks' row blocks (block_range order) of an (n, k) tensor on `dst`.

    Returns the (n_total, k) tensor on dst, None elsewhere.  Without an initialised
    process group the input is returned unchanged."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [block_range(n_total, r, world) for r in range(world)]
    pad = max(e - b for b, e in sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[: e - b] for o, (b, e) in zip(out, sizes)], 0)
