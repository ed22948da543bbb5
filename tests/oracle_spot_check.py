"""Sampled-oracle parity at bench size — TEST INFRASTRUCTURE (imports the oracle; nothing under recogym_amd/ imports this).

Trajectories are keyed by (seed, user id): the oracle can replay ANY subset of the ids of a large run.  So: run a bench
workload at size on the DEFAULT device path, pick >= 2 000 user ids spread over the range — among them the longest-lived users
and, for the user-major walk (sigma_omega = 0), users of every fate (walked to their end by round 1; parked at an
uncertified draw or handed over, i.e. taken through the float64 batch and round 2; finished by the wave-per-user last
round: rg_sim_debug_walk_fate) — replay exactly those ids with the oracle (`generate_logs(1, first_user_id=id)`, one
oracle instance per thread) and compare their rows, extracted from the sorted device log, bit for bit on (u, t, z, v, a,
c, phantom), `ps` to 1e-12 (1e-5 for the float32 logit a BanditMF table logs) and, in a second run with the click
probabilities exported, `p_click` to 1e-12 relative.

    python tests/oracle_spot_check.py [c3 c2 c3drift c4shard c5 ...] [--users N] [--sample 2000] [--out FILE]
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import golden_util as gu                                    # noqa: E402


def pick_ids(n_rows_per_user, fate, n_sample, seed):
    """-> (sorted unique user indices, {kind: how many}) : the longest-lived users, users of every walk fate, a uniform spread."""
    import torch
    n = int(n_rows_per_user.numel())
    g = torch.Generator(device='cpu').manual_seed(seed)
    parts, kinds = [], {}

    def take(name, idx, k):
        idx = idx.cpu()
        if idx.numel() > k:
            idx = idx[torch.randperm(idx.numel(), generator=g)[:k]]
        kinds[name] = int(idx.numel())
        parts.append(idx)

    k_long = min(n, max(50, n_sample // 10))
    take('longest_lived', torch.topk(n_rows_per_user, k_long).indices, k_long)
    if fate is not None:
        take('float64_batch_and_round_2', torch.nonzero((fate & 1) != 0).flatten(), n_sample // 5)
        take('finished_by_the_last_round', torch.nonzero((fate & 2) != 0).flatten(), n_sample // 5)
        take('round_1_only', torch.nonzero(fate == 0).flatten(), n_sample // 5)
    have = sum(p.numel() for p in parts)
    take('spread', torch.randperm(n, generator=g)[:max(n_sample - have, n_sample // 4)], n)
    ids = torch.unique(torch.cat(parts))
    if ids.numel() < min(n_sample, n):          # the kinds overlap (a long-lived user has a fate too): top the sample up
        extra = torch.randperm(n, generator=g)[:2 * (n_sample - ids.numel()) + 64]
        ids = torch.unique(torch.cat([ids, extra]))
    return ids.sort().values, kinds


def device_rows_of(sim, ids, out, offsets):
    """Decoded host rows (reference order) of the users `ids` (sorted indices of the reset range) from the sorted device log."""
    import torch
    from recogym_amd import _abi
    from recogym_amd.sim import decode_rows
    dev = out.device
    ids_d = ids.to(dev)
    starts, ends = offsets[ids_d], offsets[ids_d + 1]
    lens = ends - starts
    total = int(lens.sum().item())
    before = torch.cumsum(lens, 0) - lens
    gather = torch.arange(total, device=dev) - torch.repeat_interleave(before, lens) + torch.repeat_interleave(starts, lens)
    raw = out[gather].cpu().numpy()
    ps64, pc = sim.sorted_aux(offsets, out.shape[0])
    uniform = None
    if sim.policy in (_abi.RG_POLICY_UNIFORM_ENV, _abi.RG_POLICY_RANDOM_AGENT):
        uniform = 1.0 / float(sim.config.num_products)
    rows = decode_rows(raw, uniform, None if ps64 is None else ps64[gather].cpu().numpy(), None if pc is None else pc[gather].cpu().numpy())
    return rows, lens.cpu().numpy()


def oracle_rows_of(cfg, kw, ids, first_user, lens, threads):
    """The same users replayed by the oracle, one instance per thread, in id order."""
    from oracle import oracle as orc
    orc.lib()
    ids = [int(i) for i in ids]
    res = [None] * len(ids)
    nxt = [0]
    lock = threading.Lock()
    errs = []

    def work():
        try:
            env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **kw)
            while True:
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= len(ids):
                    return
                res[k] = env.generate_logs(1, 0, first_user_id=first_user + ids[k], capacity=int(lens[k]) + 64)
        except Exception as e:          # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work) for _ in range(max(1, threads))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return np.concatenate(res)


def spot_check(workload, users=None, n_sample=2000, first_user=0, env=None, seed=1, threads=None, arm_kwargs=None, config=None,
               p_click_modes=(False, True)):
    """Run `workload` (bench.WORKLOADS) over `users` users on the default device path, replay a sample with the oracle, compare.
    Raises AssertionError on the first difference; -> list of summaries (one per arm and per p_click setting)."""
    import torch
    import bench
    from recogym_amd.sim import Simulator, default_log_capacity
    cfg = config or bench.make_config(workload)
    n = int(users or bench.WORKLOADS[workload][1])
    threads = threads or min(os.cpu_count() or 1, 64)
    for k, v in (env or {}).items():
        os.environ[k] = v
    out_all = []
    try:
        for arm, kw in (arm_kwargs or bench.arms_of(workload, cfg)):
            ids = kinds = None
            for p_click in p_click_modes:
                sim = Simulator(cfg, n, device='cuda:0', log_capacity=default_log_capacity(cfg, n), p_click=p_click, **kw)
                sim.reset_users(first_user, n)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sim.run()
                torch.cuda.synchronize(); run_s = time.perf_counter() - t0
                cnt = sim.counters()
                assert cnt['live'] == 0 and cnt['log_dropped'] == 0 and cnt['hist_overflow'] == 0 and cnt['exact_overflow'] == 0, cnt
                out, offsets = sim.sorted_log()
                if ids is None:             # the same users in both runs
                    per_user = offsets[1:] - offsets[:-1]
                    fate = None
                    if cfg.sigma_omega == 0 and cnt['memo_hits'] > 0:        # a walked run
                        fate = sim.walk_fate()
                    ids, kinds = pick_ids(per_user, fate, n_sample, seed)
                rows, lens = device_rows_of(sim, ids, out, offsets)
                sim.close(); del sim, out, offsets
                torch.cuda.empty_cache()
                t0 = time.perf_counter()
                want = oracle_rows_of(cfg, kw, ids, first_user, lens, threads)
                oracle_s = time.perf_counter() - t0
                what = f'{workload}/{arm} at {n} users, p_click={p_click}'
                cols = {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')}
                if p_click:
                    cols['p_click'] = want['p_click']
                gu.assert_rows_equal(rows, cols, ps_rtol=1e-5 if 'policy_ps' in kw else 1e-12, what=what)
                assert (rows['phantom'] == want['phantom']).all(), what
                out_all.append(dict(workload=workload, arm=arm, users=n, p_click_exported=p_click, run_seconds=round(run_s, 3),
                                    sampled_users=int(ids.numel()), kinds=kinds, rows_compared=int(len(rows)),
                                    longest_sampled_user_rows=int(lens.max()), oracle_threads=threads, oracle_seconds=round(oracle_s, 1),
                                    counters={k: cnt[k] for k in ('organic', 'bandit', 'clicks', 'phantom', 'exact_draws', 'exact_sweeps',
                                                                  'anchored', 'memo_hits', 'lr_acts', 'lr_exact', 'log_rows', 'log_dropped')},
                                    verdict='rows identical to the oracle (u, t, z, v, a, c, phantom bit-exact; ps'
                                            + (', p_click' if p_click else '') + ' within tolerance)'))
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)
    return out_all


if __name__ == '__main__':
    argv = sys.argv[1:]

    def opt(name, default):
        if name in argv:
            i = argv.index(name)
            v = argv[i + 1]
            del argv[i:i + 2]
            return v
        return default
    users = int(opt('--users', 0))
    sample = int(opt('--sample', 2000))
    path = opt('--out', '')
    fh = open(path, 'a') if path else None
    for wl in (argv or ['c3']):
        for line in spot_check(wl, users or None, sample):
            s = json.dumps(line)
            print(s, flush=True)
            if fh:
                fh.write(s + '\n'); fh.flush()
