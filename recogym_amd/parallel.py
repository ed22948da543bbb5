"""Multi-GPU: users shard by id range, one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).  The step loop needs no collective — every
trajectory is keyed by (seed, user id) — so the only exchange is the all-reduce(SUM) of the
click / impression counters that test_agent and verify_agents turn into a CTR posterior
(reference: bench_agents.py:203-206, evaluate_agent.py:743-745)."""
import torch


def world():
    """-> (rank, world_size, dist module or None)."""
    try:
        import torch.distributed as dist
    except ImportError:                      # pragma: no cover
        return 0, 1, None
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), dist
    return 0, 1, None


def shard_range(n_users, rank, world_size):
    """Contiguous id range [first, first + count) of `rank`; ranges tile [0, n_users)."""
    base, rem = divmod(int(n_users), int(world_size))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def all_reduce_counts(values, device=None):
    """Sum a short list of python ints over all ranks (identity when not distributed)."""
    rank, ws, dist = world()
    if dist is None or ws == 1:
        return [int(v) for v in values]
    backend = dist.get_backend()
    dev = device if (backend == 'nccl' and device is not None) else 'cpu'
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.cpu().tolist()]
