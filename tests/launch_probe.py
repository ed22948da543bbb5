"""Worker of tests/test_host_logic.py::test_self_launch_starts_its_own_ranks — the same two calls bench.py makes
(`parallel.self_launch`, then `parallel.init_from_env`), on CPU with the gloo backend: started WITHOUT a launcher as
`python launch_probe.py --procs 2`, it must come back with both ranks' contribution all-reduced."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--procs', type=int, default=1)
    ap.add_argument('--users', type=int, default=1001)
    args = ap.parse_args()
    from recogym_amd import parallel
    rc = parallel.self_launch(args.procs, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    rank, local_rank, world, dist = parallel.init_from_env('gloo')
    assert world == args.procs, (world, args.procs)
    first, count = parallel.shard_range(args.users, rank, world)
    total, ranks = parallel.all_reduce_counts([count, 1])
    ids = parallel.all_reduce_counts([sum(range(first, first + count))])[0]
    if rank == 0:
        print(f'PROBE world={world} users={total} ranks={ranks} idsum={ids}')
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
