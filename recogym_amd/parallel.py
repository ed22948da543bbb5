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
    if backend == 'nccl':       # RCCL reduces device tensors only — also on a rank whose shard is empty
        dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    else:
        dev = 'cpu'
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.cpu().tolist()]


# ------------------------------------------------------------------------------------------
# launching: one process per GPU on one node
# ------------------------------------------------------------------------------------------
def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(n_procs, script, argv, extra_env=None):
    """`python script --gpus N` without a launcher: when this process was NOT started by
    torch.distributed.run (WORLD_SIZE unset) and n_procs > 1, run `script argv` again under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node n_procs` (127.0.0.1 rendezvous on a
    free port; the container hostname may not resolve) and return its exit code.  Returns None when no
    re-launch is needed: the caller is a rank (or the only process) and carries on."""
    import os
    import subprocess
    import sys
    if int(n_procs) <= 1 or 'WORLD_SIZE' in os.environ:
        return None
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
    env.update(extra_env or {})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(n_procs)}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), script] + list(argv)
    return subprocess.call(cmd, env=env)


def init_from_env(backend=None):
    """Join the process group a launcher described in the environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*).  backend: 'nccl' (= RCCL over xGMI, one GPU per rank) or 'gloo' (CPU tests); default: nccl when a
    GPU is visible.  -> (rank, local_rank, world_size, dist or None)."""
    import os
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if ws <= 1:
        return rank, local_rank, 1, None
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local_rank}'))
    else:
        dist.init_process_group(backend)
    return rank, local_rank, ws, dist
