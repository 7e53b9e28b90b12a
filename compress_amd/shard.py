"""Multi-GPU sharding of independent blobs: one process per GPU, blob b -> rank b mod N, no
collective on the data path.  torch.distributed is used only for the barrier and the max-over-ranks time."""
import os


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_ids(rank: int, world: int, blobs_per_rank: int):
    """Global blob ids of this rank's shard: rank, rank + world, ... (config 3: 10 000 blobs over 8 GPUs)."""
    return [rank + world * k for k in range(blobs_per_rank)]


def batch_shard_ids(rank: int, world: int, total: int):
    """This rank's blobs of ONE batch of `total` blobs sharded per blob (config 3 as stated: 10 000 blobs over the
    node's GPUs, strong scaling): ids b < total with b mod world == rank - a prefix of shard_ids(rank, world, total)."""
    return list(range(rank, total, world))


def owner(blob_id: int, world: int) -> int:
    return blob_id % world


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a scalar (elapsed time); identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
