#!/usr/bin/env python
"""bench.py — env steps/s (= log rows emitted/s) of the reco-gym-v1 step loop on MI355X.

    python bench.py --gpus N --steps K --warmup W      N > 1 without a launcher: bench.py starts its own N
                                                       ranks under torch.distributed.run (127.0.0.1 rendezvous)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (also fine)

A "step" is one pass of the hot path over one batch of synthetic input: the rank's users are reset and
simulated to completion (every Markov transition, organic product draw, click draw, policy action and log row
of `env.generate_logs(U, agent)`), inputs resident in HBM.  The default workload is BASELINE.json's headline
configuration (configs[2], the one `metric` is quoted on): reco-gym-v1, P=10 000 products, K=20, sigma_omega=0,
OrganicUserEventCounterAgent in the loop, 10 M users per GPU.  Users shard across ranks by id range with no
data-path collective; one RCCL all-reduce of the click/impression counters closes each step.
`--scaling weak` (default) keeps the per-GPU users fixed as N grows, `--scaling strong` keeps the TOTAL fixed
(10 M users over N GPUs, the way north_star states the target).

Prints ONE JSON line (rank 0).  `value` counts real rows (organic + bandit; the per-user phantom row is
excluded) over all ranks / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (env overrides, users per GPU (weak) , users in total (strong), policy)
    'c3': (dict(num_products=10000, K=20, sigma_omega=0.0), 10_000_000, 10_000_000, 'ouc'),
    'c3drift': (dict(num_products=10000, K=20, sigma_omega=0.1), 10_000_000, 10_000_000, 'ouc'),
    'c2': (dict(num_products=1000, K=20, sigma_omega=0.0), 1_000_000, 1_000_000, 'random'),
    'c4shard': (dict(num_products=100000, K=64, sigma_omega=0.1), 1_250_000, 10_000_000, 'none'),
    'tiny': (dict(num_products=100, K=20, sigma_omega=0.0), 20_000, 20_000, 'ouc'),
}

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2516.6      # v_mfma_f32_32x32x16_f16 / bf16, dense (256 CUs x 4 SIMDs x 1024 flop/cycle x 2.4 GHz)
F64_VALU_PEAK_TFLOPS = 78.6
HBM_PEAK_GBPS = 8000.0


def policy_kwargs(pol):
    from recogym_amd import _abi
    if pol == 'ouc':
        return dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=42,
                    ouc=dict(select_randomly=True, epsilon=0.0, exploit_explore=True, reverse_pop=False))
    if pol == 'random':
        return dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=42)
    return {}


def make_config(workload):
    from recogym_amd.envs.configuration import Configuration
    from recogym_amd.envs.reco_env_v1 import env_1_args
    return Configuration({**env_1_args, 'random_seed': 42, **WORKLOADS[workload][0]})


def make_sim(workload, users, device, log_rows):
    from recogym_amd.sim import Simulator
    cfg = make_config(workload)
    return cfg, Simulator(cfg, users, device=device, log_capacity=log_rows, **policy_kwargs(WORKLOADS[workload][3]))


def cpu_baseline(workload, seconds_target=12.0):
    """The oracle (plain-C float64 port of the reference loop, oracle/recogym_oracle.c) on a bounded sample of the
    same workload on ALL host cores: trajectories are keyed by (seed, user id), so every thread replays its own id
    range with its own oracle instance (ctypes releases the GIL; no allocation inside the loop).  Test
    infrastructure used as the reported CPU baseline only.  The unmodified NumPy reference cannot run on the GPU
    box (no /root/reference there): its events/s measured in the build container is quoted beside it."""
    import threading
    from oracle import oracle as orc
    cfg = make_config(workload)
    kw = policy_kwargs(WORKLOADS[workload][3])
    cores = max(1, min(os.cpu_count() or 1, 64))
    orc.lib()                                   # build / load once, before the threads start

    def measure(n_threads, seconds):
        results = [None] * n_threads

        def work(k):
            env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **kw)
            first = 1_000_000 * k               # disjoint id ranges
            users, events, batch = 0, 0, 25
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < seconds and users < 20000:
                rows = env.generate_logs(batch, first_user_id=first + users, capacity=batch * 2000 + 10000)
                events += int((rows['phantom'] == 0).sum())
                users += batch
            results[k] = (users, events)

        threads = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        wall = time.perf_counter() - t0
        return sum(r[0] for r in results), sum(r[1] for r in results), wall

    u1, e1, w1 = measure(1, 3.0)
    users, events, wall = measure(cores, seconds_target)
    out = dict(value=events / wall, unit='events/s', cores=cores, kind='port',
               per_core=events / wall / cores, one_thread=e1 / w1,
               sample=f'{users} users / {events} events of the same workload in {wall:.1f} s on {cores} threads '
                      f'(oracle/recogym_oracle.c, float64, one oracle instance per thread); one thread alone: '
                      f'{e1 / w1:.0f} events/s')
    try:      # the NumPy reference itself, measured where /root/reference exists (tools/time_reference.py)
        ref = json.load(open(os.path.join(ROOT, 'profiles', 'r2', 'numpy_reference_cpu.json')))
        case = {'c3': 'c3', 'c3drift': 'c3', 'c2': 'c2', 'c4shard': 'c4_capped', 'tiny': 'c1'}[workload]
        rc = ref['cases'][case]
        out['reference_numpy'] = dict(
            one_core_events_per_s=rc['one_core_events_per_s'], all_core_events_per_s=rc['all_core_events_per_s'],
            processes=rc['processes'], case=case, env=rc['env'], agent=rc['agent'], users=rc['users'],
            provenance=f"unmodified criteo-research/reco-gym (NumPy {ref['numpy']}, numba absent) timed by "
                       f"tools/time_reference.py in the build container ({ref['cpu_count']} vCPU), NOT on this box; "
                       f"profiles/r2/numpy_reference_cpu.json")
    except Exception:
        out['reference_numpy'] = None
    return out


def kernel_rooflines(cfg, prof, c, users, cached, pol):
    """Per-kernel rooflines from the HIP-event timings the library records on its launch stream
    (rg_sim_set_profiling).  Algorithmic units follow SURVEY.md §8d (stated in DESIGN.md §6)."""
    P, K = cfg.num_products, cfg.K
    f16_split = (3 * K + 1) <= 64
    out = {}
    # organic product sweeps on the matrix pipe: 2*P*K flop per swept draw.  With sigma_omega = 0 only a user's
    # first draw sweeps (the rest search the per-user cache); otherwise every lock-step organic draw does.
    swept = users if cached else c['organic']
    if prof['draw_mfma_ms'] > 0:
        tf = 2.0 * P * K * swept / (prof['draw_mfma_ms'] * 1e-3) / 1e12
        peak = F16_MFMA_PEAK_TFLOPS if K <= 64 else FP32_MFMA_PEAK_TFLOPS
        out['draw_sweep'] = dict(kernel='k_draw_bf16p' if K <= 21 else 'k_draw_* (K class)', bound='mfma',
                                 ms=round(prof['draw_mfma_ms'], 2), units=int(swept), unit_name='swept draws',
                                 achieved=round(tf, 2), peak=peak, unit='TFLOP/s', frac=round(tf / peak, 4),
                                 executed_mfma_tflops=round(tf * (64.0 if f16_split else 144.0) / K, 1) if K <= 21 else None)
    walked = prof.get('walk1_ms', 0.0) > 0
    # the user-major walk (sigma_omega = 0, run to the end): SURVEY.md 8d bytes per event — bandit: omega 4K + state 8 +
    # row 16 + action 3 (+35 history with the OUC agent in the loop); organic: state 8 + the cached draw's 128 B of
    # super-chunk sums, 48 B of chunk sums, omega32 (4K) and the 16-byte row
    if walked:
        b_b = 4 * K + 8 + 16 + 3 + (35 if pol == 'ouc' else 0)
        b_o = 8 + 128 + 48 + 4 * K + 16
        by = b_b * c['bandit'] + b_o * c['organic']
        ms = prof['walk1_ms'] + prof['walk2_ms']
        gbps = by / (ms * 1e-3) / 1e9
        out['walk'] = dict(kernel='k_walk', bound='hbm', ms=round(ms, 2), round1_ms=round(prof['walk1_ms'], 2),
                           round2_ms=round(prof['walk2_ms'], 2), units=int(c['bandit'] + c['organic']), unit_name='events',
                           bytes_per_unit=round(by / max(c['bandit'] + c['organic'], 1), 1), achieved=round(gbps, 1),
                           peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(gbps / HBM_PEAK_GBPS, 4),
                           note='VALU-issue / latency bound on per-user state that lives in L2 and the Infinity Cache '
                                '(DESIGN.md 4), not on HBM bandwidth')
        if prof['draw_search_ms'] > 0:
            out['cache_finalize'] = dict(kernel='k_cache_finalize', bound='hbm', ms=round(prof['draw_search_ms'], 2), units=int(users),
                                         unit_name='users', bytes_per_unit=256 + 8 * K + 256,
                                         achieved=round((512 + 8 * K) * users / (prof['draw_search_ms'] * 1e-3) / 1e9, 1),
                                         peak=HBM_PEAK_GBPS, unit='GB/s',
                                         frac=round((512 + 8 * K) * users / (prof['draw_search_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4))
    # cached draw (sigma_omega = 0, t >= 1, lock-step form): per draw the user's 32 super-chunk sums (128 B), the chosen
    # super-chunk's chunk sums (48 B), omega32 (4K) and the 16-byte row: HBM gather
    if cached and not walked and prof['draw_search_ms'] > 0:
        n = c['organic'] - users
        by = 128 + 48 + 4 * K + 16
        gbps = by * n / (prof['draw_search_ms'] * 1e-3) / 1e9
        out['draw_cached'] = dict(kernel='k_draw_cached', bound='hbm', ms=round(prof['draw_search_ms'], 2), units=int(n),
                                  unit_name='cached draws', bytes_per_unit=by, achieved=round(gbps, 1),
                                  peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(gbps / HBM_PEAK_GBPS, 4))
    # float64 resolve: per sweep P * (2K + ~16) float64 flop-equivalents on the VALU
    if prof['draw_exact_ms'] > 0 and c['exact_sweeps'] > 0:
        tf = c['exact_sweeps'] * float(P) * (2 * K + 16) / (prof['draw_exact_ms'] * 1e-3) / 1e12
        out['draw_exact_f64'] = dict(kernel='k_exact_sums_m + k_exact_pick', bound='f64 mfma + valu', ms=round(prof['draw_exact_ms'], 2),
                                     units=int(c['exact_sweeps']), unit_name='float64 sweeps', achieved=round(tf, 2),
                                     peak=F64_VALU_PEAK_TFLOPS, unit='TFLOP/s', frac=round(tf / F64_VALU_PEAK_TFLOPS, 4),
                                     resolved_draws=int(c['exact_draws']))
    # advance: SURVEY.md 8d bytes per event — bandit: omega 4K + state 8 + row 16 + action 3 (+35 history with the
    # OUC agent in the loop); organic: state 8 (+ omega write 4K when it drifts)
    if prof['advance_ms'] > 0:
        b_b = 4 * K + 8 + 16 + 3 + (35 if pol == 'ouc' else 0)
        b_o = 8 + (4 * K if cfg.sigma_omega != 0 else 0)
        by = b_b * c['bandit'] + b_o * c['organic']
        gbps = by / (prof['advance_ms'] * 1e-3) / 1e9
        out['advance'] = dict(kernel='k_advance', bound='hbm', ms=round(prof['advance_ms'], 2),
                              units=int(c['bandit'] + c['organic']), unit_name='events',
                              bytes_per_unit=round(by / max(c['bandit'] + c['organic'], 1), 1),
                              achieved=round(gbps, 1), peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(gbps / HBM_PEAK_GBPS, 4))
    return out


def measured_traffic(name):
    """HBM bytes per unit of a kernel from the committed rocprofv3 --pmc passes (profiles/r2/pmc_traffic.json;
    PMC counters cannot be read from inside this process) — None when that kernel was not profiled."""
    try:
        pt = json.load(open(os.path.join(ROOT, 'profiles', 'r2', 'pmc_traffic.json')))
        return pt['kernels'][name]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--users', type=int, default=0, help='users per GPU (weak) / in total (strong); default: workload size')
    ap.add_argument('--no-log', action='store_true', help='counters only (no 16 B/row log writes)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-drift-line', action='store_true', help='skip the sigma_omega > 0 companion measurement')
    args = ap.parse_args()

    from recogym_amd import parallel
    rc = parallel.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:
        sys.exit(rc)

    import torch
    rank, local_rank, world, dist = parallel.init_from_env('nccl')
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    device = torch.device(f'cuda:{local_rank}')
    torch.cuda.set_device(device)

    import __graft_entry__ as graft
    if rank == 0:
        graft.build()
    if dist:
        dist.barrier()

    from recogym_amd.sim import default_log_capacity
    _, per_gpu, total, pol = WORKLOADS[args.workload]
    if args.scaling == 'weak':
        users = args.users or per_gpu
        first_user = rank * users       # disjoint id ranges: identical to one big run
        users_total = users * world
    else:
        users_total = args.users or total
        first_user, users = parallel.shard_range(users_total, rank, world)

    def build(workload, n):
        log_rows = 0 if args.no_log else default_log_capacity(make_config(workload), n)
        return make_sim(workload, n, device, log_rows)

    cfg, sim = build(args.workload, users)

    def one_step(s=None):
        s = s or sim
        s.reset_users(first_user, users)
        s.run()
        c = s.counters()
        vec = torch.tensor([c['organic'], c['bandit'], c['clicks'], c['phantom']],
                           dtype=torch.int64, device=device)
        if dist:
            dist.all_reduce(vec)        # the CTR reduction of test_agent / verify_agents
        return c, vec

    def sync():
        if dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed(s, steps, warmup):
        # (the warm-up also runs the few torch ops of the timed loop once: their kernels are loaded lazily, ~20 ms the
        # first time, which is 2 ms per step of a 10-step run and more than a whole step of the C2 workload)
        totals = torch.zeros(4, dtype=torch.int64, device=device)
        for _ in range(warmup):
            _, vec = one_step(s)
            totals += vec
        totals.zero_()
        sync()
        t0 = time.perf_counter()
        last = None
        for _ in range(steps):
            last, vec = one_step(s)
            totals += vec
        sync()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), totals.cpu().numpy(), last

    elapsed, totals, last = timed(sim, args.steps, args.warmup)
    events = int(totals[0] + totals[1])
    assert last['hist_overflow'] == 0 and last['log_dropped'] == 0 and last['live'] == 0, last

    def profile(s, config):
        s.set_profiling(True)
        s.reset_users(first_user, users)
        s.run()
        prof = s.profile()
        c = s.counters()
        s.set_profiling(False)
        cached = config.sigma_omega == 0 and os.environ.get('RECOGYM_CACHE', '1') != '0' and config.K <= 63
        return prof, c, kernel_rooflines(config, prof, c, users, cached, pol)

    # --- per-kernel rooflines, HIP events on the launch stream; `roofline` = the kernel with the most time ---
    roofline = kernels = None
    if rank == 0:
        prof, c, kernels = profile(sim, cfg)
        dom = max((k for k in kernels if kernels[k]['bound'] in ('hbm', 'mfma')), key=lambda k: kernels[k]['ms'])
        roofline = dict(kernels[dom])
        launches = max(prof['steps'], 1)
        if roofline['unit'] == 'GB/s':
            roofline['achieved_is'] = ('algorithmic bytes (SURVEY.md 8d figure per unit x units) / total kernel time over '
                                       f'{launches} launches; bound by scattered per-user gathers, not by streaming bandwidth')
        roofline['launches'] = launches
        roofline['avg_launch_ms'] = round(roofline['ms'] / launches, 4)
        if dom == 'walk':
            roofline['achieved_is'] = ('algorithmic bytes (SURVEY.md 8d figure per event x events) / total kernel time of the run\'s '
                                       'walk launches (rounds of unequal size: per-launch figures are the run\'s divided by their number)')
        pmc = measured_traffic(roofline['kernel'].split(' ')[0]) if args.workload in ('c3', 'c4shard') else None
        if dom == 'walk':
            # rounds of one run: 1, the parked users' round 2, and round 3 for what draining waves handed over
            launches = (3 if os.environ.get('RECOGYM_WALK_HANDOVER', '16') != '0' else 2) if prof['walk2_ms'] > 0 else 1
            roofline['launches'] = launches
            roofline['avg_launch_ms'] = round(roofline['ms'] / launches, 4)
        roofline['traffic'] = None if pmc is None else pmc['hbm_bytes_per_unit'] * roofline['units'] / launches
        roofline['traffic_source'] = None if pmc is None else pmc['source']
        roofline['tail_ms'] = round(prof['tail_ms'], 2)
        roofline['exact_fraction'] = round(c['exact_draws'] / max(c['organic'], 1), 5)
        roofline['whole_job_hbm_algorithmic_GBps'] = round(
            (events / args.steps / world) * (4 * cfg.K + 8 + 16 + 3 + (35 if pol == 'ouc' else 0)) / 1e9 / (elapsed / args.steps), 1)
    sim.close()
    del sim
    torch.cuda.empty_cache()

    # --- the sigma_omega > 0 companion of the headline workload: no per-user cache is possible there, every
    # organic draw sweeps all P products on the matrix pipe — the line that shows the sweep kernel's quality ---
    drift = None
    if args.workload == 'c3' and not args.no_drift_line:
        dcfg, dsim = build('c3drift', users)
        d_el, d_tot, d_last = timed(dsim, 1, 1)
        if rank == 0:
            d_prof, d_c, d_k = profile(dsim, dcfg)
            drift = dict(workload='c3drift: the same with sigma_omega=0.1 (omega drifts at every organic transition)',
                         value=float(d_tot[0] + d_tot[1]) / d_el, unit='events/s', ms_per_step=1e3 * d_el,
                         steps=1, warmup=1, kernels=d_k,
                         exact_fraction=round(d_c['exact_draws'] / max(d_c['organic'], 1), 5))
        dsim.close()
        del dsim
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:     # the CPU leg runs at N = 1 only
        cpu = cpu_baseline(args.workload)

    if rank == 0:
        w = WORKLOADS[args.workload][0]
        out = {
            'metric': f'env steps/sec (organic+bandit events emitted/sec), reco-gym-v1 P={w["num_products"]} K={w["K"]}',
            'value': events / elapsed,
            'unit': 'events/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f64 (state, click, transition, policy, propensities); organic logits on MFMA as a 2-way fp16 split '
                     'of fp32 operands (fp32 accumulate), every index certified against f64',
            'data': 'synthetic',
            'config': {'workload': f'{args.workload}: reco-gym-v1 P={cfg.num_products} K={cfg.K} '
                                   f'sigma_omega={cfg.sigma_omega} policy={pol}',
                       'users_per_gpu': users, 'users_total': users_total,
                       'events_per_step': events // args.steps,
                       'log': 'off' if args.no_log else '16 B/row device log + float64 ps side array',
                       'ctr': float(totals[2]) / max(float(totals[1] + totals[3]), 1.0)},
            'roofline': roofline,
            'kernels': kernels,
            'sigma_omega_gt0': drift,
            'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
