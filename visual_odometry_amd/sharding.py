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
    # gloo gathers host tensors only: device rows are staged through the host there (RCCL
    # gathers them where they are)
    device = torch.device("cpu") if dist.get_backend() == "gloo" else local.device
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=device)
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[: e - b] for o, (b, e) in zip(out, sizes)], 0)


def sharded_knn2(knn2, q_all, t, dst=0):
    """Brute-force 2-nearest-neighbour table of q_all against t with the QUERY ROWS sharded over
    the ranks: this rank runs knn2(q_rows, t) -> (idx2, dist2), two (rows, 2) int32 tensors, on
    its block_range of q_all, and the (rows, 4) blocks [idx2 | dist2] are gathered on `dst` with
    one collective.  Every rank holds all of t, so each block is final and nothing is reduced.
    Returns the (len(q_all), 4) table on dst, None elsewhere."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    b, e = block_range(len(q_all), rank, world)
    idx2, dist2 = knn2(q_all[b:e], t)
    return gather_rows(torch.cat([idx2, dist2], 1), len(q_all), dst=dst)
