"""Does RCCL work on this box?  One rank, then two ranks on ONE device (RCCL may refuse a duplicate GPU): init, all_reduce of the
counters recogym_amd.parallel.all_reduce_counts sums.  usage: python tools/rccl_probe.py <world_size>  (starts its own ranks)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from recogym_amd import parallel  # noqa: E402

ws = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if 'WORLD_SIZE' not in os.environ:
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={ws}', '--master-addr', '127.0.0.1',
           '--master-port', str(parallel.free_port()), os.path.abspath(__file__), str(ws)]
    sys.exit(subprocess.call(cmd, env=env))
import torch.distributed as dist  # noqa: E402
rank = int(os.environ['RANK'])
torch.cuda.set_device(0)          # every rank on device 0: this box has one
dist.init_process_group('nccl', device_id=torch.device('cuda:0'))
t = torch.tensor([rank + 1, 10 * (rank + 1), 7], dtype=torch.int64, device='cuda:0')
dist.all_reduce(t, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
want = [sum(r + 1 for r in range(ws)), sum(10 * (r + 1) for r in range(ws)), 7 * ws]
assert t.cpu().tolist() == want, (t.cpu().tolist(), want)
got = parallel.all_reduce_counts([rank + 1, 5])
assert got == [want[0], 5 * ws] or ws == 1, got
if rank == 0:
    print(f'RCCL ok: backend {dist.get_backend()}, world {dist.get_world_size()}, all_reduce -> {t.cpu().tolist()}, all_reduce_counts -> {got}', flush=True)
dist.destroy_process_group()
