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

Other workloads (`--workload`): c2 / c4shard = BASELINE configs 2 / one rank's share of 4; c3drift = config 3 with
the reference's default drift; c5 = BASELINE config 5 on one rank's share: the `verify_agents` A/B loop with a frozen
BanditMFSquare table arm and a frozen 10^4-class LogregMulticlassIps arm over the SAME users (a step = both arms).

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
    # verify_agents(env, users, {BanditMFSquare, LogregMulticlassIps}) with the env's defaults (sigma_omega = 0.1)
    'c5': (dict(num_products=10000, K=20), 1_250_000, 10_000_000, 'c5'),
    # the same A/B loop at a size the REFERENCE can train: both policies fitted by the reference's own code (tests/make_golden.py
    # c5_trained -> tests/golden/c5_trained_p100.npz: LogregMulticlassIps build() and BanditMFSquare train() on 1 000 users)
    'c5trained': (dict(num_products=100, K=20), 1_250_000, 10_000_000, 'c5trained'),
    'tiny': (dict(num_products=100, K=20, sigma_omega=0.0), 20_000, 20_000, 'ouc'),
    'tiny5': (dict(num_products=300, K=20), 20_000, 20_000, 'c5'),
}

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2516.6      # v_mfma_f32_32x32x16_f16 / bf16, dense (256 CUs x 4 SIMDs x 1024 flop/cycle x 2.4 GHz)
F64_VALU_PEAK_TFLOPS = 78.6
HBM_PEAK_GBPS = 8000.0
EXP_PEAK_PER_S = 1024 * 4 * 2.4e9  # v_exp_f32: quarter rate, 4 lanes per cycle and SIMD (MI355X_MICROARCH.md: transcendentals)
VALU_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4   # a SIMD issues one full-rate vector wave-instruction per 4 cycles: 256 CUs x 4 SIMDs
PROFILE_ROUNDS = ('r6', 'r5', 'r4', 'r3', 'r2')      # profiles/<round>/pmc_traffic.json, newest first


def policy_kwargs(pol):
    from recogym_amd import _abi
    if pol == 'ouc':
        return dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=42,
                    ouc=dict(select_randomly=True, epsilon=0.0, exploit_explore=True, reverse_pop=False))
    if pol == 'random':
        return dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=42)
    return {}


def arms_of(workload, cfg):
    """[(arm name, Simulator policy kwargs)].  Every workload but c5 has one arm.  c5: the two frozen policies of BASELINE
    config 5 — a BanditMFSquare inference table (last viewed product -> action, its logit as `ps`) and a
    LogregMulticlassIps model with one class per product.  The models are random (training is the reference's host
    code, SURVEY.md §2: a random table and N(0, 0.1) coefficients), seeded, the same on every rank."""
    pol = WORKLOADS[workload][3]
    if pol == 'c5trained':
        import numpy as np
        from recogym_amd.agents import LastViewTableAgent, LogregFrozenAgent
        z = np.load(os.path.join(ROOT, 'tests', 'golden', 'c5_trained_p100.npz'))
        assert int(z['bmf_product_embedding'].shape[0]) == cfg.num_products
        mf = LastViewTableAgent.from_bandit_mf(cfg, z['bmf_product_embedding'], z['bmf_user_embedding'])
        lr = LogregFrozenAgent(cfg, z['logreg_coef'], z['logreg_intercept'], z['logreg_classes'])
        strip = lambda d: {k: v for k, v in d.items() if k != 'ouc'}
        return [('bandit_mf_fitted', strip(mf.device_policy())), ('logreg_ips_fitted', strip(lr.device_policy()))]
    if pol != 'c5':
        return [(pol, policy_kwargs(pol))]
    import numpy as np
    from recogym_amd import _abi
    P = cfg.num_products
    rng = np.random.RandomState(0)
    table = rng.randint(0, P, size=P).astype(np.int32)
    ps = rng.rand(P)
    coef_t = (rng.standard_normal((P, P)) * 0.1)
    intercept = rng.standard_normal(P) * 0.1
    return [('bandit_mf_frozen', dict(policy=_abi.RG_POLICY_LAST_VIEW_TABLE, policy_seed=0, policy_table=table, policy_ps=ps)),
            ('logreg_ips_frozen', dict(policy=_abi.RG_POLICY_LOGREG_FROZEN, policy_seed=0,
                                       logreg=dict(coef_t=coef_t, intercept=intercept, classes=np.arange(P, dtype=np.int32))))]


def make_config(workload):
    from recogym_amd.envs.configuration import Configuration
    from recogym_amd.envs.reco_env_v1 import env_1_args
    return Configuration({**env_1_args, 'random_seed': 42, **WORKLOADS[workload][0]})


def make_sim(workload, users, device, log_rows):
    """(config, simulator) of a one-arm workload (tools use this)."""
    from recogym_amd.sim import Simulator
    cfg = make_config(workload)
    return cfg, Simulator(cfg, users, device=device, log_capacity=log_rows, **arms_of(workload, cfg)[0][1])


def cpu_baseline(workload, seconds_target=12.0):
    """The CPU leg.  `reference_numpy`: the unmodified NumPy reference — it cannot travel to the GPU box (no
    /root/reference there), so its events/s are the ones tools/time_reference.py measured in the build container,
    quoted with that provenance.  `value` (kind "port"): the oracle (plain-C float64 restatement of the reference loop,
    oracle/recogym_oracle.c) timed LIVE on this box's host cores on a bounded sample of the same workload: trajectories
    are keyed by (seed, user id), so every thread replays its own id range with its own oracle instance (ctypes
    releases the GIL; no allocation inside the loop).  Test infrastructure used as the reported baseline only."""
    import threading
    from oracle import oracle as orc
    cfg = make_config(workload)
    kw = arms_of(workload, cfg)[0][1]
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

    out = {}
    try:      # the NumPy reference itself, measured where /root/reference exists (tools/time_reference.py)
        ref_path = next(p for p in (os.path.join(ROOT, 'profiles', r, 'numpy_reference_cpu.json') for r in ('r4', 'r2')) if os.path.exists(p))
        ref = json.load(open(ref_path))
        case = {'c3': 'c3', 'c3drift': 'c3', 'c2': 'c2', 'c4shard': 'c4_capped', 'tiny': 'c1', 'c5': 'c3', 'tiny5': 'c1', 'c5trained': 'c2'}[workload]
        rc = ref['cases'][case]
        out['reference_numpy'] = dict(
            one_core_events_per_s=rc['one_core_events_per_s'], all_core_events_per_s=rc['all_core_events_per_s'],
            processes=rc['processes'], case=case, env=rc['env'], agent=rc['agent'], users=rc['users'],
            provenance=f"unmodified criteo-research/reco-gym (NumPy {ref['numpy']}, numba absent) timed by "
                       f"tools/time_reference.py in the build container ({ref['cpu_count']} vCPU), NOT on this box; "
                       f"{os.path.relpath(ref_path, ROOT)}")
        if rc.get('port_same_box'):      # the C port on the same samples in the same container: a same-box port / NumPy ratio
            ps = rc['port_same_box']
            out['reference_numpy'].update(
                port_same_box_one_thread_events_per_s=ps['one_thread_events_per_s'],
                port_same_box_all_thread_events_per_s=ps['all_thread_events_per_s'],
                port_over_numpy_one_core=ps['port_over_numpy_one_core'], port_over_numpy_all_core=ps['port_over_numpy_all_core'])
    except Exception:
        out['reference_numpy'] = None
    u1, e1, w1 = measure(1, 3.0)
    users, events, wall = measure(cores, seconds_target)
    rn = out.get('reference_numpy') or {}
    if rn.get('port_over_numpy_one_core'):
        # what the NumPy reference would do on THIS box's cores: the live port figures divided by the same-box ratio
        rn['estimated_on_this_box'] = dict(one_core_events_per_s=e1 / w1 / rn['port_over_numpy_one_core'],
                                           all_core_events_per_s=events / wall / rn['port_over_numpy_all_core'], cores=cores,
                                           how='this box\'s live C-port timing / the port-over-NumPy ratio measured in the build container')
    out.update(value=events / wall, unit='events/s', cores=cores, kind='port',
               per_core=events / wall / cores, one_thread=e1 / w1,
               sample=f'{users} users / {events} events of the same workload in {wall:.1f} s on {cores} threads '
                      f'(oracle/recogym_oracle.c, float64, one oracle instance per thread); one thread alone: '
                      f'{e1 / w1:.0f} events/s')
    return out


def survey_bytes_per_event(cfg, pol):
    """fp32 state per event — omega 4K + (state, t) 8 + packed row 16 + action 3 (0.78 x 4 B), + 35 B of
    view history with the OrganicUserEventCounter agent in the loop, + 0.2 x 4K omega write where omega drifts."""
    b = 4 * cfg.K + 8 + 16 + 3 + (35 if pol == 'ouc' else 0)
    if cfg.sigma_omega != 0:
        b += round(0.8 * cfg.K)
    return b


def kernel_rooflines(cfg, prof, c, users, cached, pol):
    """Per-kernel rooflines from the HIP-event timings the library records on its launch stream
    (rg_sim_set_profiling).  Algorithmic units follow SURVEY.md §8d (stated in DESIGN.md §6)."""
    P, K = cfg.num_products, cfg.K
    f16_split = (3 * K + 1) <= 64
    events = c['bandit'] + c['organic']
    out = {}
    # organic product sweeps on the matrix pipe: 2*P*K flop per swept draw.  With sigma_omega = 0 only a user's
    # first draw sweeps (the rest search the per-user cache); otherwise every lock-step organic draw does.
    swept = users if cached else c['organic']
    if prof['draw_mfma_ms'] > 0:
        tf = 2.0 * P * K * swept / (prof['draw_mfma_ms'] * 1e-3) / 1e12
        peak = F16_MFMA_PEAK_TFLOPS if K <= 64 else FP32_MFMA_PEAK_TFLOPS
        exp_ms = 1e3 * float(P) * swept / EXP_PEAK_PER_S
        tp = (not cached) and (K <= 20 or 21 < K <= 64) and prof.get('sweep_lds', 0) and prof['draw_search_ms'] > 0
        out['draw_sweep'] = dict(kernel=('k_sweep_xh' if (cached and K <= 20 and os.environ.get('RECOGYM_XH', '1') != '0') else
                                         ('k_draw_tp (unsliced rounds; k_draw_bf16p the sliced ones)' if tp else 'k_draw_bf16p')) if K <= 21 else
                                        ('k_draw_tpw (unsliced rounds; k_draw_f16w the sliced ones)' if tp else 'k_draw_* (K class)'), bound='mfma',
                                 ms=round(prof['draw_mfma_ms'], 2), units=int(swept), unit_name='swept draws',
                                 achieved=round(tf, 2), peak=peak, unit='TFLOP/s', frac=round(tf / peak, 4),
                                 executed_mfma_tflops=round(tf * ((112.0 if K > 8 else 48.0) if (cached and K <= 20 and os.environ.get('RECOGYM_XH', '1') != '0')
                                                                  else (64.0 if f16_split else 144.0)) / K, 1) if K <= 21 else None,
                                 # one v_exp_f32 per logit at a quarter of the fp32 lane rate: the kernel's real ceiling
                                 exp_bound_ms=round(exp_ms, 2), frac_of_exp_bound=round(exp_ms / prof['draw_mfma_ms'], 4))
        if tp:
            # k_pick: the draw inside its tile (128 products; the wide sweep's super-tiles: ~1 300) on the matrix cores, 32 draws of one
            # tile per wave: 2 x tile x K flop per draw
            tile_products = 128 if K <= 20 else prof.get('tp_tile_products', 1344)
            tfp = 2.0 * tile_products * K * swept / (prof['draw_search_ms'] * 1e-3) / 1e12
            out['draw_pick'] = dict(kernel='k_pick (+ k_draw_search on sliced rounds)', bound='mfma', ms=round(prof['draw_search_ms'], 2), units=int(swept),
                                    unit_name='draws', achieved=round(tfp, 3), peak=peak, unit='TFLOP/s', frac=round(tfp / peak, 5),
                                    note='latency-bound: a wave per 32 draws of one tile, ~9 dependent round trips per group')
    walked = prof.get('walk1_ms', 0.0) > 0
    b_survey = survey_bytes_per_event(cfg, pol)
    # the user-major walk (sigma_omega = 0, run to the end).  `bytes_per_unit` is SURVEY.md 8d's figure; what the
    # implementation really moves is the measured `traffic_bytes_per_unit` of the roofline entry (PMC passes)
    if walked:
        ms = prof['walk1_ms'] + prof['walk2_ms']
        gbps = b_survey * events / (ms * 1e-3) / 1e9
        out['walk'] = dict(kernel='k_walk', kernels='k_walk2 (rounds 1-2) + k_walk_solo (last round)', bound='hbm', ms=round(ms, 2), round1_ms=round(prof['walk1_ms'], 2),
                           later_rounds_ms=round(prof['walk2_ms'], 2), units=int(events), unit_name='events',
                           bytes_per_unit=b_survey,
                           achieved=round(gbps, 1), peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(gbps / HBM_PEAK_GBPS, 4),
                           note='not bound by HBM bandwidth: the per-user state lives in L2 and the Infinity Cache, and the kernel is '
                                'bound by its instruction stream and its dependent loads (DESIGN.md 4): `issue_roofline` prices the '
                                'vector wave-instructions it executes against the vector ALUs\' issue rate')
        iss = measured_issue(pol)
        if iss is not None:
            issue_ms = 1e3 * iss['valu_wave_instr_per_unit'] * events / VALU_WAVE_INSTR_PER_S
            out['walk']['issue_roofline'] = dict(valu_wave_instr_per_event=round(iss['valu_wave_instr_per_unit'], 2),
                                                 salu_wave_instr_per_event=round(iss.get('salu_wave_instr_per_unit', 0.0), 2),
                                                 peak_valu_wave_instr_per_s=VALU_WAVE_INSTR_PER_S, issue_bound_ms=round(issue_ms, 2),
                                                 frac=round(issue_ms / ms, 4), source=iss['source'],
                                                 note='SQ_INSTS_VALU of the profiled run per event x this run\'s events x 4 cycles / (1024 SIMDs x '
                                                      '2.4 GHz) over the walk kernels\' time: the fraction of the time the vector ALUs must issue')
        if prof['draw_search_ms'] > 0:
            # since round 4 the sweep leaves this output itself; the two kernels only visit the users whose reference moved during
            # the sweep (a count the library does not export): time only, no roofline
            out['cache_finalize'] = dict(kernel='k_cache_finalize + k_cache_prefix', bound='none', ms=round(prof['draw_search_ms'], 2),
                                         units=None, unit_name='users whose reference moved during the sweep (not counted)')
    # cached draw (sigma_omega = 0, t >= 1, lock-step form): per draw the user's 32 super-chunk sums (128 B), the chosen
    # super-chunk's chunk sums (48 B), omega32 (4K) and the 16-byte row: HBM gather
    if cached and not walked and prof['draw_search_ms'] > 0:
        n = c['organic'] - users
        by = 128 + 48 + 4 * K + 16
        gbps = by * n / (prof['draw_search_ms'] * 1e-3) / 1e9
        out['draw_cached'] = dict(kernel='k_draw_cached', bound='hbm', ms=round(prof['draw_search_ms'], 2), units=int(n),
                                  unit_name='cached draws', bytes_per_unit=by, achieved=round(gbps, 1),
                                  peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(gbps / HBM_PEAK_GBPS, 4))
    # float64 resolve: per sweep P * (2K + ~16) float64 flop-equivalents
    if prof['draw_exact_ms'] > 0 and c['exact_sweeps'] > 0:
        tf = c['exact_sweeps'] * float(P) * (2 * K + 16) / (prof['draw_exact_ms'] * 1e-3) / 1e12
        out['draw_exact_f64'] = dict(kernel='k_exact_sums_* + k_exact_pick', bound='mfma', pipe='float64 matrix + vector pipes (they share issue: DESIGN.md 4)', ms=round(prof['draw_exact_ms'], 2),
                                     units=int(c['exact_sweeps']), unit_name='float64 sweeps', achieved=round(tf, 2),
                                     peak=F64_VALU_PEAK_TFLOPS, unit='TFLOP/s', frac=round(tf / F64_VALU_PEAK_TFLOPS, 4),
                                     resolved_draws=int(c['exact_draws']))
    # frozen LogReg acts: the act of a user is recomputed when its view history changed; per act the coef^T rows of its
    # viewed products over all classes at the precision the screening pass stores them in (fp16: 2 B per weight, the default;
    # RECOGYM_LOGREG=int8: 1 B — round 6, measured slower —, =fp32: 4 B) — what the kernel has to read (the float64 refine touches a few values per act)
    if prof.get('logreg_ms', 0.0) > 0 and c.get('lr_acts', 0) > 0:
        n_classes = P
        mode = os.environ.get('RECOGYM_LOGREG', 'fp16')
        wbytes = 4.0 if mode == 'fp32' or n_classes % 8 else (1.0 if mode == 'int8' else 2.0)
        by = wbytes * n_classes * c['lr_rows']
        gbps = by / (prof['logreg_ms'] * 1e-3) / 1e9
        out['logreg_acts'] = dict(kernel='k_logreg_select + k_logreg_screen + k_logreg_decide' if wbytes <= 2.0 else 'k_logreg_select + k_logreg_acts',
                                  bound='hbm', ms=round(prof['logreg_ms'], 2),
                                  units=int(c['lr_acts']), unit_name='acts', bytes_per_unit=round(by / c['lr_acts'], 1),
                                  rows_per_act=round(c['lr_rows'] / c['lr_acts'], 2), float64_refined_acts=int(c.get('lr_exact', 0)),
                                  launches=int(prof['steps']), us_per_launch=round(1e3 * prof['logreg_ms'] / max(prof['steps'], 1), 1),
                                  achieved=round(gbps, 1), peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(gbps / HBM_PEAK_GBPS, 4),
                                  note='late rounds have few acts (about one per wave slot): the time of such a round\'s act kernels is the '
                                       'latency of one act, not throughput')
    # advance: SURVEY.md 8d bytes per event
    if prof['advance_ms'] > 0:
        gbps = b_survey * events / (prof['advance_ms'] * 1e-3) / 1e9
        out['advance'] = dict(kernel='k_advance_run + k_drift (rounds: a user\'s bandit run per launch)', bound='hbm', ms=round(prof['advance_ms'], 2),
                              launches=int(prof['steps']),
                              units=int(events), unit_name='events', bytes_per_unit=b_survey,
                              achieved=round(gbps, 1), peak=HBM_PEAK_GBPS, unit='GB/s', frac=round(gbps / HBM_PEAK_GBPS, 4))
    return out


def measured_issue(pol):
    """Vector / scalar wave-instructions per event of the walk kernels from the committed SQ counter pass
    (profiles/<round>/pmc_traffic.json, entry `k_walk_issue`) — None when there is none for this policy."""
    for rnd in PROFILE_ROUNDS:
        try:
            pt = json.load(open(os.path.join(ROOT, 'profiles', rnd, 'pmc_traffic.json')))
        except Exception:
            continue
        e = pt.get('kernels', {}).get('k_walk_issue')
        if e is not None and e.get('policy', 'ouc') == pol:
            return e
    return None


def measured_traffic(workload, name):
    """HBM bytes per unit of a kernel from the committed rocprofv3 --pmc passes (profiles/<round>/pmc_traffic.json; PMC
    counters cannot be read from inside this process) — None when that kernel was not profiled on this workload."""
    for rnd in PROFILE_ROUNDS:
        try:
            pt = json.load(open(os.path.join(ROOT, 'profiles', rnd, 'pmc_traffic.json')))
        except Exception:
            continue
        e = pt.get('kernels', {}).get(name)
        if e is None:
            continue
        wl = e.get('workload')
        if wl == workload or (wl is None and workload in ('c3', 'c4shard')):
            return e
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
    ap.add_argument('--no-materialise', action='store_true', help='skip the sort_log / log_columns timings')
    ap.add_argument('--no-other-workloads', action='store_true', help='skip the one-step lines of BASELINE configs 2, 4 (one rank\'s share) and 5')
    ap.add_argument('--single-run', action='store_true',
                    help='exactly ONE simulation per arm in the process (the HIP-event profile is taken on the timed run): what '
                         'the rocprofv3 --pmc passes run, so that a counter summed over the trace belongs to one run')
    ap.add_argument('--shard', default='', help='R/W: with --gpus 1, simulate rank R\'s id range of a W-rank strongly scaled job')
    ap.add_argument('--digest', action='store_true', help='add the order-independent checksum of the last run\'s log rows')
    args = ap.parse_args()
    if args.single_run:
        args.steps, args.warmup = 1, 0
        args.no_cpu_baseline = args.no_drift_line = args.no_materialise = args.no_other_workloads = True

    from recogym_amd import parallel
    rc = parallel.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:
        sys.exit(rc)

    import torch
    # (RECOGYM_BENCH_BACKEND=gloo RECOGYM_BENCH_ONE_DEVICE=1: the N > 1 code path on a ONE-GPU box, every rank on cuda:0 — a test
    # of this script's multi-rank logic, tests/test_hip_parity.py; never a measurement)
    one_device = os.environ.get('RECOGYM_BENCH_ONE_DEVICE') == '1'
    rank, local_rank, world, dist = parallel.init_from_env(os.environ.get('RECOGYM_BENCH_BACKEND', 'nccl'))
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if one_device:
        local_rank = 0
    device = torch.device(f'cuda:{local_rank}')
    torch.cuda.set_device(device)

    import __graft_entry__ as graft
    if rank == 0:
        graft.build()
    if dist:
        dist.barrier()

    from recogym_amd.sim import Simulator, default_log_capacity
    _, per_gpu, total, pol = WORKLOADS[args.workload]
    shard_note = None
    if args.shard:
        assert world == 1, '--shard emulates a rank on ONE GPU'
        r, w = (int(x) for x in args.shard.split('/'))
        users_total = args.users or total
        first_user, users = parallel.shard_range(users_total, r, w)
        shard_note = f'rank {r} of {w} (strong scaling of {users_total} users), on one GPU'
    elif args.scaling == 'weak':
        users = args.users or per_gpu
        first_user = rank * users       # disjoint id ranges: identical to one big run
        users_total = users * world
    else:
        users_total = args.users or total
        first_user, users = parallel.shard_range(users_total, rank, world)

    # N > 1: what every rank saw of the job — world size as torch.distributed reports it, its device — so that the line of a real
    # multi-GPU run proves the collective spanned N ranks on N devices
    rank_info = None
    if dist:
        mine = dict(rank=rank, world_size=int(dist.get_world_size()), backend=str(dist.get_backend()), device=str(device),
                    device_name=torch.cuda.get_device_name(device), host=os.uname().nodename)
        try:
            mine['pci_bus_id'] = str(torch.cuda.get_device_properties(device).pci_bus_id)
        except Exception:
            pass
        rank_info = [None] * world
        dist.all_gather_object(rank_info, mine)

    def build(workload, n):
        cfg = make_config(workload)
        log_rows = 0 if args.no_log else default_log_capacity(cfg, n)
        return cfg, [(name, Simulator(cfg, n, device=device, log_capacity=log_rows, **kw)) for name, kw in arms_of(workload, cfg)]

    cfg, arms = build(args.workload, users)

    def one_step(arm_list, profiled=False):
        """every arm over the same users; -> (last counters per arm, summed [organic, bandit, clicks, phantom])"""
        vec = torch.zeros(4, dtype=torch.int64, device=device)
        per_arm = []
        for _, s in arm_list:
            if profiled:
                s.set_profiling(True)
            s.reset_users(first_user, users)
            s.run()
            c = s.counters()
            per_arm.append(c)
            v = torch.tensor([c['organic'], c['bandit'], c['clicks'], c['phantom']], dtype=torch.int64, device=device)
            if dist:
                dist.all_reduce(v)      # the CTR reduction of test_agent / verify_agents, once per agent
            vec += v
        return per_arm, vec

    per_rank_ms = []          # ms per step of every rank in the last timed() (N > 1)

    def sync():
        if dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed(arm_list, steps, warmup, profiled=False):
        # (the warm-up also runs the few torch ops of the timed loop once: their kernels are loaded lazily, ~20 ms the
        # first time, which is 2 ms per step of a 10-step run and more than a whole step of the C2 workload)
        totals = torch.zeros(4, dtype=torch.int64, device=device)
        for _ in range(warmup):
            _, vec = one_step(arm_list)
            totals += vec
        totals.zero_()
        sync()
        t0 = time.perf_counter()
        last = None
        for _ in range(steps):
            last, vec = one_step(arm_list, profiled)
            totals += vec
        sync()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if dist:
            every = [torch.zeros_like(el) for _ in range(world)]
            dist.all_gather(every, el)
            per_rank_ms[:] = [round(1e3 * float(x.item()) / max(steps, 1), 3) for x in every]
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), totals.cpu().numpy(), last

    elapsed, totals, last = timed(arms, args.steps, args.warmup, profiled=args.single_run)
    events = int(totals[0] + totals[1])
    ranks_ms = list(per_rank_ms)
    for c in last:
        assert c['hist_overflow'] == 0 and c['log_dropped'] == 0 and c['live'] == 0 and c['exact_overflow'] == 0, c

    def is_cached(config):
        return config.sigma_omega == 0 and os.environ.get('RECOGYM_CACHE', '1') != '0' and config.K <= 63

    def profile(arm_list, config, already=False):
        """HIP-event profile of one more run of every arm (or of the timed run itself: --single-run)"""
        out = {}
        counters = []
        for (name, s), c_timed in zip(arm_list, last if already else [None] * len(arm_list)):
            if already:
                c = c_timed
            else:
                s.set_profiling(True)
                s.reset_users(first_user, users)
                s.run()
                c = s.counters()
            prof = s.profile()
            try:
                prof['sweep_lds'] = s.get_option('sweep_lds_kernel') and s.get_option('sweep_lds')
            except Exception:
                prof['sweep_lds'] = 0
            s.set_profiling(False)
            counters.append((c, prof))
            arm_pol = name if name in ('ouc', 'random', 'none') else name
            for k, v in kernel_rooflines(config, prof, c, users, is_cached(config), arm_pol).items():
                out[k if len(arm_list) == 1 else f'{name}.{k}'] = dict(v, arm=name) if len(arm_list) > 1 else v
        return counters, out

    # --- per-kernel rooflines, HIP events on the launch stream; `roofline` = the kernel with the most time ---
    roofline = kernels = None
    materialise = None
    if rank == 0:
        counters, kernels = profile(arms, cfg, already=args.single_run)
        dom = max((k for k in kernels if kernels[k]['bound'] in ('hbm', 'mfma')), key=lambda k: kernels[k]['ms'])
        roofline = dict(kernels[dom])
        c, prof = counters[0] if len(counters) == 1 else counters[[n for n, _ in arms].index(roofline['arm'])]
        launches = max(prof['steps'], 1)
        roofline['dominant'] = dom
        if roofline['kernel'] == 'k_walk':
            # rounds of one run: 1, the parked users' round 2, and round 3 for what draining waves handed over
            launches = (3 if arms[0][1].get_option('walk_handover') != 0 else 2) if prof['walk2_ms'] > 0 else 1
            roofline['achieved_is'] = ('SURVEY.md 8d bytes per event x events of the run / total time of the run\'s walk launches '
                                       '(rounds of unequal size: avg_launch_ms is the run\'s time divided by their number)')
        elif roofline['unit'] == 'GB/s':
            roofline['achieved_is'] = (f'algorithmic bytes per unit x units of the run / total kernel time over {launches} launches')
        roofline['launches'] = launches
        roofline['avg_launch_ms'] = round(roofline['ms'] / launches, 4)
        pmc = measured_traffic(args.workload, roofline['kernel'].split(' ')[0])
        if pmc is not None and roofline['unit'] == 'GB/s':
            # HBM bytes of ONE run's launches of this kernel (FETCH_SIZE doubled as the microarchitecture guide prescribes
            # for gfx950, + WRITE_SIZE; per-unit figure of the profiled run x this run's units)
            roofline['traffic'] = pmc['hbm_bytes_per_unit'] * roofline['units']
            roofline['traffic_per_launch'] = roofline['traffic'] / launches
            roofline['traffic_bytes_per_unit'] = round(pmc['hbm_bytes_per_unit'], 1)
            roofline['wasted_traffic_ratio'] = round(pmc['hbm_bytes_per_unit'] / roofline['bytes_per_unit'], 2)
            roofline['traffic_source'] = pmc['source']
        else:
            roofline['traffic'] = None
            roofline['traffic_source'] = None if pmc is None else pmc['source']
            if pmc is not None:
                roofline['traffic_bytes_per_unit'] = round(pmc['hbm_bytes_per_unit'], 1)
        roofline['tail_ms'] = round(prof['tail_ms'], 2)
        roofline['exact_fraction'] = round(c['exact_draws'] / max(c['organic'], 1), 5)
        b_ev = survey_bytes_per_event(cfg, pol)
        roofline['whole_job_hbm_algorithmic_GBps'] = round((events / args.steps / world) * b_ev / 1e9 / (elapsed / args.steps), 1)
        roofline['whole_job_frac'] = round(roofline['whole_job_hbm_algorithmic_GBps'] / HBM_PEAK_GBPS, 4)

        # --- what the reference's row order costs (abstract.py:299-327 is part of generate_logs): rg_sim_sort_log on the
        # whole log of the last run; decoding into the reference's columns on a 200 k-user run of the same workload ---
        if not args.no_materialise and not args.no_log:
            s0 = arms[0][1]
            srt, off = s0.sorted_log()              # (first call: the 16 GB result buffer comes from hipMalloc, not the cache)
            del srt, off
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            srt, off = s0.sorted_log()
            torch.cuda.synchronize(device)
            sort_ms = 1e3 * (time.perf_counter() - t0)
            n_rows = int(srt.shape[0])
            del srt, off
            torch.cuda.empty_cache()
            # the whole log of the benched run to the host, as the reference's columns: sort + decode on the device, chunks over
            # PCIe into pinned buffers the simulator keeps (first call: incl. their allocation; second: steady state)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            cols = s0.log_columns(copy=False)
            torch.cuda.synchronize(device)
            dt_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            cols = s0.log_columns(copy=False)
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            n_cols = int(len(cols['t']))
            by_row = sum(int(v.dtype.itemsize) for v in cols.values())
            materialise = dict(sort_log_ms=round(sort_ms, 2), sort_log_rows=n_rows,
                               sort_log_rows_per_s=round(n_rows / (sort_ms * 1e-3), 0),
                               log_columns_rows=n_cols, log_columns_rows_per_s=round(n_cols / dt, 0),
                               log_columns_first_call_rows_per_s=round(n_cols / dt_first, 0),
                               log_columns_seconds=round(dt, 3), log_columns_first_call_seconds=round(dt_first, 3),
                               host_bytes_per_row=by_row, pcie_GBps=round(n_cols * by_row / dt / 1e9, 1),
                               pinned_host_buffers=bool(getattr(s0, '_host_cols_pinned', False)),
                               note='outside the timed region, on the WHOLE log of one benched run: rg_sim_sort_log (reference row '
                                    'order), then Simulator.log_columns — sort + decode into the reference\'s columns on the device, '
                                    'chunks over PCIe (decode of chunk i beside the copy of chunk i - 1) into pinned host buffers kept '
                                    'between calls, returned as zero-copy NumPy views; first call = incl. allocating / page-locking them')
            del cols
            s0._host_cols = None
            s0._stage_cols = None
            torch.cuda.empty_cache()
    digest = None
    if args.digest:
        digest = [arm.log_digest() for _, arm in arms]
    for _, s in arms:
        s.close()
    del arms
    torch.cuda.empty_cache()

    # --- the sigma_omega > 0 companion of the headline workload: no per-user cache is possible there, every
    # organic draw sweeps all P products on the matrix pipe — the line that shows the sweep kernel's quality ---
    drift = None
    if args.workload == 'c3' and not args.no_drift_line:
        dcfg, darms = build('c3drift', users)
        d_steps = 2
        d_el, d_tot, d_last = timed(darms, d_steps, 1)
        if rank == 0:
            d_counters, d_k = profile(darms, dcfg)
            d_c = d_counters[0][0]
            drift = dict(workload='c3drift: the same with sigma_omega=0.1 (omega drifts at every organic transition)',
                         value=float(d_tot[0] + d_tot[1]) / d_el, unit='events/s', ms_per_step=1e3 * d_el / d_steps,
                         steps=d_steps, warmup=1, kernels=d_k,
                         exact_fraction=round(d_c['exact_draws'] / max(d_c['organic'], 1), 5))
        for _, s in darms:
            s.close()
        del darms
        torch.cuda.empty_cache()

    # --- the other BASELINE configurations, three timed steps each (after one warm-up), so that the driver's clock sees them too:
    # config 2, one rank's share of config 4, config 5 (both arms) — compact entries: value, ms per step, the kernel with the most time ---
    others = None
    if args.workload == 'c3' and world == 1 and not args.no_other_workloads and not args.shard:
        others = {}
        saved_fu = (first_user, users)
        for wl in ('c2', 'c4shard', 'c5'):
            first_user, users = 0, WORKLOADS[wl][1]
            ocfg, oarms = build(wl, users)
            o_steps = 3
            o_el, o_tot, o_last = timed(oarms, o_steps, 1)
            for c in o_last:
                assert c['hist_overflow'] == 0 and c['log_dropped'] == 0 and c['live'] == 0 and c['exact_overflow'] == 0, c
            _, o_k = profile(oarms, ocfg)
            o_dom = max(o_k, key=lambda k: o_k[k]['ms'])
            others[wl] = dict(workload=f'{wl}: reco-gym-v1 P={ocfg.num_products} K={ocfg.K} sigma_omega={ocfg.sigma_omega} policy={WORKLOADS[wl][3]}',
                              users=users, value=float(o_tot[0] + o_tot[1]) / o_el, unit='events/s', ms_per_step=round(1e3 * o_el / o_steps, 2),
                              events_per_step=int(o_tot[0] + o_tot[1]) // o_steps, steps=o_steps, warmup=1,
                              dominant=o_dom, dominant_kernel=o_k[o_dom]['kernel'], dominant_ms=o_k[o_dom]['ms'],
                              dominant_bound=o_k[o_dom]['bound'], dominant_frac=o_k[o_dom].get('frac'),
                              frac_of_exp_bound=o_k[o_dom].get('frac_of_exp_bound'),
                              kernels_ms={k: v['ms'] for k, v in o_k.items()})
            for _, s_o in oarms:
                s_o.close()
            del oarms
            torch.cuda.empty_cache()
        first_user, users = saved_fu

    # --- N > 1: the OTHER scaling form in the same run (north_star states the target as 10 M users in total over 1/2/4/8
    # GPUs: strong; the contract's `value` keeps per-GPU work fixed: weak), and the latency of the one collective the path has ---
    other = allreduce = None
    if dist and not args.shard and not args.single_run:
        o_form = 'strong' if args.scaling == 'weak' else 'weak'
        if o_form == 'strong':
            o_total = args.users * world if args.users else total
            o_first, o_users = parallel.shard_range(o_total, rank, world)
        else:
            o_users = (args.users // world if args.users else per_gpu)
            o_first, o_total = rank * o_users, o_users * world
        saved = (first_user, users)
        first_user, users = o_first, o_users          # (one_step reads these)
        _, o_arms = build(args.workload, o_users)
        o_el, o_tot, _ = timed(o_arms, args.steps, args.warmup)
        other = dict(scaling=o_form, value=float(o_tot[0] + o_tot[1]) / o_el, unit='events/s', ms_per_step=1e3 * o_el / args.steps,
                     users_total=int(o_total), users_per_gpu=int(o_users), events_per_step=int(o_tot[0] + o_tot[1]) // args.steps,
                     per_rank_ms_per_step=list(per_rank_ms))
        for _, sim_o in o_arms:
            sim_o.close()
        del o_arms
        torch.cuda.empty_cache()
        first_user, users = saved
        # the CTR reduction itself: 100 all-reduces of the 24-byte {clicks, impressions, phantom} vector, each timed to completion
        v = torch.zeros(3, dtype=torch.int64, device=device)
        for _ in range(10):
            dist.all_reduce(v)
        torch.cuda.synchronize(device)
        lat = []
        for _ in range(100):
            dist.barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            dist.all_reduce(v)
            torch.cuda.synchronize(device)
            lat.append(1e6 * (time.perf_counter() - t0))
        lat.sort()
        allreduce = dict(bytes=24, median_us=round(lat[50], 1), p10_us=round(lat[10], 1), p90_us=round(lat[90], 1), samples=100,
                         note='one all_reduce(SUM) per agent per evaluation closes test_agent / verify_agents (bench_agents.py:203-206): '
                              'host-timed from launch to completion, after a barrier')

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
                                   f'sigma_omega={cfg.sigma_omega} policy={pol}'
                                   + (' (verify_agents: frozen BanditMFSquare table arm + frozen LogregMulticlassIps arm, same users)'
                                      if pol == 'c5' else '')
                                   + (' (verify_agents: both arms FITTED BY THE REFERENCE\'s code on 1 000 users, same users)'
                                      if pol == 'c5trained' else ''),
                       'users_per_gpu': users, 'users_total': users_total, 'first_user': first_user,
                       'events_per_step': events // args.steps,
                       'log': 'off' if args.no_log else '16 B/row device log + float64 ps side array',
                       'ctr': float(totals[2]) / max(float(totals[1] + totals[3]), 1.0)},
            'roofline': roofline,
            'kernels': kernels,
            'materialise': materialise,
            'sigma_omega_gt0': drift,
            'other_workloads': others,
            'cpu_baseline': cpu,
        }
        if materialise is not None:
            # the reference's generate_logs returns ORDERED rows (abstract.py:299-327): the same figure with rg_sim_sort_log on the
            # whole log of a step inside it (the sort is timed on the benched run's log, second call)
            ms_ord = 1e3 * elapsed / args.steps + materialise['sort_log_ms']
            out['value_ordered'] = (events / args.steps) / (ms_ord * 1e-3)
            out['ms_per_step_ordered'] = ms_ord
        if roofline is not None and kernels and 'issue_roofline' in kernels.get('walk', {}):
            roofline['issue_frac'] = kernels['walk']['issue_roofline']['frac']          # (the walk's vector-issue roofline beside its HBM one)
            roofline['issue_bound_ms'] = kernels['walk']['issue_roofline']['issue_bound_ms']
        if world > 1:
            out['ranks'] = rank_info
            out['per_rank_ms_per_step'] = ranks_ms
            out['other_scaling'] = other
            out['allreduce_us'] = None if allreduce is None else allreduce['median_us']
            out['allreduce'] = allreduce
        if shard_note:
            out['config']['shard'] = shard_note
        if digest is not None:
            out['digest'] = digest
            out['totals'] = dict(organic=int(totals[0]), bandit=int(totals[1]), clicks=int(totals[2]), phantom=int(totals[3]))
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
