"""Parity of the HIP step loop (through the C ABI) with the oracle and with the committed
reference fixtures.  Bit-exact on (u, t, z, v, a, c); `ps` (float64 side array of the log, exact
1/P for the uniform policies) and the click probability `p_click` of every real bandit row to 1e-12
relative — the tolerance BASELINE.json's north_star asks to be stated for click probabilities: both
sides evaluate ff(beta[a].omega + mu_b[a]) in float64 with k-ascending sums, the device's exp /
divide differ from libm's by an ulp or two; needs a real MI355X."""
import numpy as np
import pytest
import torch

import golden_util as gu
from recogym_amd import _abi
from recogym_amd.envs.configuration import Configuration
from recogym_amd.envs.reco_env_v1 import env_1_args

pytestmark = pytest.mark.gpu


PCLICK_RTOL = 1e-12


def run_sim(config, n_users, n_organic=0, first_user=0, **pol):
    from recogym_amd.sim import Simulator
    pol.setdefault('p_click', True)
    sim = Simulator(config, n_users + n_organic, device='cuda:0', **pol)
    sim.reset_users(first_user, n_users + n_organic, organic_only_below=first_user + n_organic)
    sim.run()
    rows = sim.rows()
    cnt = sim.counters()
    sim.close()
    return rows, cnt


@pytest.mark.parametrize('name', gu.fixtures('philox_'))
def test_hip_reproduces_reference_fixture(name):
    """The committed logs of the unmodified reference (counter RNG injected), row for row."""
    meta, cols = gu.load(name)
    extra = {} if gu.env0_tables(meta) is None else dict(env0=gu.env0_tables(meta))      # reco-gym-v0 fixtures: env_kind = 1
    rows, cnt = run_sim(gu.env_config(meta), meta['n_users'], meta['n_organic'],
                        **gu.policy_args(meta, cols), **extra)
    # ps: float64 on both sides (BanditMF logs a float32 torch logit: compared at float32 resolution)
    gu.assert_rows_equal(rows, cols, ps_rtol=1e-5 if meta['agent'] == 'bmf' else 1e-12, what=name)
    if 'p_click' in cols:          # every real bandit row's click probability vs the reference's own value
        real = (cols['z'] == 1) & ~np.isnan(cols['p_click'])
        assert real.sum() == cnt['bandit'] and np.isfinite(rows['p_click'][real]).all()
    assert cnt['organic'] == int((cols['z'] == 0).sum())
    assert cnt['bandit'] + cnt['phantom'] == int((cols['z'] == 1).sum())
    assert cnt['clicks'] == int((cols['c'] == 1).sum())
    assert cnt['live'] == 0 and cnt['log_dropped'] == 0


CASES = [
    # (env overrides, n_users, n_organic, policy args)
    (dict(num_products=10, K=5, random_seed=1), 3000, 100, {}),
    (dict(num_products=1, K=1, random_seed=2), 500, 0, {}),
    (dict(num_products=63, K=3, random_seed=3, sigma_omega=0.3), 1500, 0, {}),
    (dict(num_products=64, K=64, random_seed=4), 800, 0, {}),
    (dict(num_products=65, K=33, random_seed=5, change_omega_for_bandits=True), 800, 7, {}),
    (dict(num_products=1000, K=20, random_seed=6, sigma_omega=0.0), 2000, 0, {}),
    (dict(num_products=10000, K=20, random_seed=7), 300, 0, {}),
    (dict(num_products=4097, K=17, random_seed=8, number_of_flips=20), 400, 0,
     dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=99)),
    (dict(num_products=40, K=8, random_seed=9, prob_leave_organic=0.05, prob_organic_to_bandit=0.5,
          prob_bandit_to_organic=0.2), 4000, 0, {}),
    (dict(num_products=200, K=20, random_seed=10), 1500, 0,
     dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=21,
          ouc=dict(gu.OUC_DEFAULTS))),
    (dict(num_products=30, K=6, random_seed=11), 1000, 0,
     dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=22,
          ouc=dict(gu.OUC_DEFAULTS, epsilon=0.25))),
    (dict(num_products=30, K=6, random_seed=12), 600, 0,
     dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=23,
          ouc=dict(gu.OUC_DEFAULTS, select_randomly=False))),
    (dict(num_products=25, K=6, random_seed=13), 600, 0,
     dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=24,
          ouc=dict(gu.OUC_DEFAULTS, exploit_explore=False, epsilon=0.5, reverse_pop=True))),
    # more than 65536 products: the view history keeps separate (product u32, count u16) arrays
    # instead of packed 16+16-bit entries
    (dict(num_products=70000, K=4, random_seed=14), 120, 0,
     dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=25, ouc=dict(gu.OUC_DEFAULTS))),
    # BASELINE config 4's shape: P = 100 000, K = 64, omega drifting (beyond the reference's int16 ceiling)
    (dict(num_products=100000, K=64, random_seed=15, sigma_omega=0.1), 12, 0, {}),
    # BASELINE config 3's shape with its agent in the loop: OrganicUserEventCounter at P = 10 000
    (dict(num_products=10000, K=20, random_seed=16, sigma_omega=0.0), 200, 0,
     dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=26, ouc=dict(gu.OUC_DEFAULTS))),
]


@pytest.mark.parametrize('case', range(len(CASES)))
def test_hip_matches_oracle(case):
    """Seeded configurations the fixtures do not cover (edge sizes: P=1, P around the 64-lane
    chunk, K odd/large, flips, warm-up users, every policy variant), vs the oracle run here."""
    from oracle import oracle as orc
    over, n_users, n_org, pol = CASES[case]
    cfg = Configuration({**env_1_args, **over})
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(n_users, n_org)
    rows, cnt = run_sim(cfg, n_users, n_org, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps', 'p_click')},
                         ps_rtol=1e-12, what=f'case {case}')
    assert (rows['phantom'] == want['phantom']).all()
    oc = want_env.counters()
    assert (cnt['organic'], cnt['bandit'], cnt['clicks'], cnt['phantom']) == \
        (oc['organic'], oc['bandit'], oc['clicks'], oc['phantom'])


@pytest.mark.parametrize('shape', [(10, 2, 3000, 50, 'uniform'), (1000, 10, 20000, 0, 'ouc'), (96, 3, 4000, 0, 'random'), (2, 1, 500, 0, 'uniform')])
def test_env0_matches_the_oracle(shape):
    """reco-gym-v0 on the device (rg_config.env_kind = 1: k_draw_env0, the table click of k_advance) against the oracle's
    restatement of recogym/envs/reco_env_v0.py, which the MT fixtures pin on the unmodified reference: rows bit for bit, the
    click probability of every real bandit row (a table value: exact)."""
    from oracle import oracle as orc
    from recogym_amd.envs.reco_env_v0 import env_0_args
    from recogym_amd.envs.static_params import draw_env0_tables
    P, nc, n, n_org, pk = shape
    cfg = Configuration({**env_0_args, 'random_seed': 500 + P, 'num_products': P, 'num_clusters': nc, 'phi_var': 0.3})
    t0 = draw_env0_tables(cfg)
    pol = {'uniform': {}, 'random': dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=3),
           'ouc': dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=4, ouc=dict(gu.OUC_DEFAULTS))}[pk]
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, env0=t0, **pol)
    want = want_env.generate_logs(n, n_org)
    rows, cnt = run_sim(cfg, n, n_org, env0=t0, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps', 'p_click')}, ps_rtol=1e-12, what=f'env0 {shape}')
    assert (rows['phantom'] == want['phantom']).all()
    oc = want_env.counters()
    assert (cnt['organic'], cnt['bandit'], cnt['clicks'], cnt['phantom']) == (oc['organic'], oc['bandit'], oc['clicks'], oc['phantom'])
    assert cnt['clicks'] > 0 and cnt['live'] == 0


@pytest.mark.parametrize('lockstep', [False, True])
@pytest.mark.parametrize('case', [0, 5, 6, 7, 8, 9, 15])
def test_fp32_decided_clicks_match_the_oracle(case, lockstep, monkeypatch):
    """Without the click-probability export k_walk decides a click from an fp32 evaluation of
    ff(beta[a].omega + mu_b[a]) wherever its error margin allows and falls back to float64 inside the margin (measured
    in k_advance too: no gain there, not kept): the logged clicks must be the oracle's, walked or in lock-step."""
    from oracle import oracle as orc
    over, n_users, n_org, pol = CASES[case]
    if lockstep:
        monkeypatch.setenv('RECOGYM_WALK', '0')
        monkeypatch.setenv('RECOGYM_TAIL', '0')
    cfg = Configuration({**env_1_args, **over})
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol).generate_logs(n_users, n_org)
    rows, cnt = run_sim(cfg, n_users, n_org, p_click=False, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')}, ps_rtol=1e-12,
                         what=f'fp32 clicks, case {case}, lockstep {lockstep}')
    assert cnt['clicks'] == int((want['c'] == 1).sum()) > 0


@pytest.mark.parametrize('case', [0, 4, 7, 9, 10])
def test_repacked_state_matches_oracle(case, monkeypatch):
    """The state repack (live users' omega / view history / user ids copied into list order every
    few steps; on by default from 2^18 users) is a pure relabelling: forced on for small runs, every
    3 steps, the rows still equal the oracle's."""
    from oracle import oracle as orc
    monkeypatch.setenv('RECOGYM_REPACK_MIN', '1')
    monkeypatch.setenv('RECOGYM_REPACK', '3')
    over, n_users, n_org, pol = CASES[case]
    cfg = Configuration({**env_1_args, **over})
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol).generate_logs(n_users, n_org)
    rows, cnt = run_sim(cfg, n_users, n_org, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')},
                         ps_rtol=1e-6, what=f'repacked case {case}')
    assert (rows['phantom'] == want['phantom']).all()
    assert cnt['hist_overflow'] == 0 and cnt['live'] == 0


@pytest.mark.parametrize('case', [0, 2, 4, 8, 9, 12])
def test_lock_step_to_the_end_matches_oracle(case, monkeypatch):
    """rg_sim_run normally finishes the last users of a run with the per-user tail kernel (every
    small case here runs mostly in it); with RECOGYM_TAIL=0 the lock-step kernels run to the end."""
    from oracle import oracle as orc
    monkeypatch.setenv('RECOGYM_TAIL', '0')
    over, n_users, n_org, pol = CASES[case]
    cfg = Configuration({**env_1_args, **over})
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(n_users, n_org)
    rows, cnt = run_sim(cfg, n_users, n_org, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')},
                         ps_rtol=1e-6, what=f'lock-step case {case}')
    oc = want_env.counters()
    assert (cnt['organic'], cnt['bandit'], cnt['clicks'], cnt['phantom']) == \
        (oc['organic'], oc['bandit'], oc['clicks'], oc['phantom'])


@pytest.mark.parametrize('hops', [0, 1, 3, 1000])
@pytest.mark.parametrize('case', [0, 2, 7, 8, 9, 10, 12])
def test_run_ahead_rounds_match_the_oracle(case, hops, monkeypatch):
    """A run to the end goes in ROUNDS (k_advance_run): a launch takes every listed user from its organic event (or the bandit event
    the last round left it at) through its bandit run — up to `hops` events, to the first transition that is not "bandit" or the
    first event that can click — so users sit at different event indices; draws and rows are keyed by the user's own index.
    Whatever the cap (0 = lock-step, an event per launch; 1 = rounds of one event; 3 = runs cut short; 1000 = never cut), the log
    is the oracle's.  RECOGYM_TAIL=0: the rounds, not the per-user tail kernel, run these small cases to the end."""
    from oracle import oracle as orc
    monkeypatch.setenv('RECOGYM_TAIL', '0')
    monkeypatch.setenv('RECOGYM_RUN_AHEAD', str(hops))
    over, n_users, n_org, pol = CASES[case]
    if case == 0:
        n_org = 100
    cfg = Configuration({**env_1_args, **over})
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(n_users, n_org)
    rows, cnt = run_sim(cfg, n_users, n_org, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps', 'p_click')},
                         ps_rtol=1e-12, what=f'run-ahead {hops}, case {case}')
    assert (rows['phantom'] == want['phantom']).all()
    oc = want_env.counters()
    assert (cnt['organic'], cnt['bandit'], cnt['clicks'], cnt['phantom']) == \
        (oc['organic'], oc['bandit'], oc['clicks'], oc['phantom'])
    assert cnt['live'] == 0 and cnt['log_dropped'] == 0 and cnt['log_rows'] == cnt['organic'] + cnt['bandit']


@pytest.mark.parametrize('tail', ['0', None])
@pytest.mark.parametrize('hops', [0, 2, 32])
@pytest.mark.parametrize('name', ['philox_normal_time', 'philox_normal_time_ouc', 'philox_logreg', 'philox_bandit_mf',
                                  'philox_p32767_k64_drift', 'philox_random_agent'])
def test_run_ahead_rounds_reproduce_reference_fixtures(name, hops, tail, monkeypatch):
    """The same against logs of the unmodified reference: per-user clocks (NormalTimeGenerator: the clock and the drift's scale move
    with every event of a run), the frozen LogReg act (one act serves a whole bandit run), the last-viewed-product table, the wide
    sweep; with the tail kernel taking over from the rounds (default) and without."""
    if tail is not None:
        monkeypatch.setenv('RECOGYM_TAIL', tail)
    monkeypatch.setenv('RECOGYM_RUN_AHEAD', str(hops))
    meta, cols = gu.load(name)
    rows, cnt = run_sim(gu.env_config(meta), meta['n_users'], meta['n_organic'], **gu.policy_args(meta, cols))
    gu.assert_rows_equal(rows, cols, ps_rtol=1e-5 if meta['agent'] == 'bmf' else 1e-12, what=f'{name}, run-ahead {hops}, tail {tail}')
    assert cnt['organic'] == int((cols['z'] == 0).sum())
    assert cnt['bandit'] + cnt['phantom'] == int((cols['z'] == 1).sum())
    assert cnt['clicks'] == int((cols['c'] == 1).sum())
    assert cnt['live'] == 0 and cnt['log_dropped'] == 0


def test_repacked_fixture_last_view_table(monkeypatch):
    """Same, for the policy whose per-user state (last product viewed) also moves."""
    monkeypatch.setenv('RECOGYM_REPACK_MIN', '1')
    monkeypatch.setenv('RECOGYM_REPACK', '2')
    meta, cols = gu.load('philox_bandit_mf')
    rows, cnt = run_sim(gu.env_config(meta), meta['n_users'], meta['n_organic'],
                        **gu.policy_args(meta, cols))
    gu.assert_rows_equal(rows, cols, ps_rtol=1e-5, what='repacked philox_bandit_mf')


def test_omega_export_after_repack(monkeypatch):
    """omega() / states() address users, not slots, after a repack (sigma_omega = 0: a live user's
    omega is still its reset draw; rows of users that left read 0)."""
    from oracle import oracle as orc
    from recogym_amd.sim import Simulator
    monkeypatch.setenv('RECOGYM_REPACK_MIN', '1')
    monkeypatch.setenv('RECOGYM_REPACK', '2')
    cfg = Configuration({**env_1_args, 'random_seed': 8, 'num_products': 20, 'K': 7, 'sigma_omega': 0.0,
                         'prob_leave_organic': 0.1, 'prob_leave_bandit': 0.1})
    sim = Simulator(cfg, 400, device='cuda:0')
    sim.reset_users(50, 400)
    for _ in range(9):
        sim.step()
    om = sim.omega().cpu().numpy()
    st = sim.states().cpu().numpy()
    sim.close()
    o = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX)
    want = o.generate_logs(400, first_user_id=50)
    n_rows = np.bincount(want['u'] - 50, minlength=400)
    alive9 = n_rows - 1 > 9            # rows minus the phantom row = events; alive at t = 9 needs a 10th event
    assert 20 < alive9.sum() < 380
    assert ((st != _abi.RG_STATE_STOP) == alive9).all()
    bad = np.flatnonzero(~alive9 & (om != 0).any(axis=1))
    def reset_omega(i):
        o.reset(50 + int(i))
        return o.omega
    diag = ''
    if len(bad):
        own = sum(np.allclose(om[i], reset_omega(i)) for i in bad[:20])
        live_ok = sum(np.allclose(om[i], reset_omega(i)) for i in np.flatnonzero(alive9)[:20])
        diag = f'{len(bad)} dead rows non-zero; {own}/20 hold the user\'s own omega; live rows right {live_ok}/20'
    assert len(bad) == 0, diag
    for i in np.flatnonzero(alive9)[:40]:
        o.reset(50 + int(i))
        np.testing.assert_allclose(om[i], o.omega, rtol=1e-13, atol=1e-15)


def test_user_sharding_is_invisible():
    """Trajectories are keyed by (seed, user id): simulating users [0,N) at once or in four
    shards with first_user offsets gives identical rows (the multi-GPU contract)."""
    cfg = Configuration({**env_1_args, 'random_seed': 77, 'num_products': 300, 'K': 12})
    whole, _ = run_sim(cfg, 2000)
    parts = [run_sim(cfg, 500, first_user=i * 500)[0] for i in range(4)]
    glued = np.concatenate(parts)
    assert len(glued) == len(whole)
    for k in whole.dtype.names:
        assert np.array_equal(glued[k], whole[k], equal_nan=True), k


def test_omega_matches_oracle_after_reset():
    from oracle import oracle as orc
    from recogym_amd.sim import Simulator
    cfg = Configuration({**env_1_args, 'random_seed': 31, 'num_products': 20, 'K': 7})
    sim = Simulator(cfg, 50, device='cuda:0')
    sim.reset_users(1000, 50)
    om = sim.omega().cpu().numpy()
    o = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX)
    for i in (0, 17, 49):
        o.reset(1000 + i)
        np.testing.assert_allclose(om[i], o.omega, rtol=1e-13, atol=1e-15)
    st = sim.states().cpu().numpy()
    assert (st == _abi.RG_STATE_ORGANIC).all()


def test_more_than_a_million_users_shard_consistently():
    """Grid-stride paths (> 4096 blocks of users): one 1.3 M-user run equals the sum of three
    shards — counters and an order-independent checksum of every log row."""
    from recogym_amd.sim import Simulator
    cfg = Configuration({**env_1_args, 'random_seed': 5, 'num_products': 10, 'K': 5})

    def run(first, n):
        sim = Simulator(cfg, n, device='cuda:0')
        sim.reset_users(first, n)
        sim.run()
        c = sim.counters()
        rows = sim.raw_log().to(torch.int64)
        assert rows.shape[0] == c['organic'] + c['bandit']
        chk = [(rows[:, i] * (rows[:, 1] + 7)).sum().item() for i in range(3)]
        sim.close()
        return c, chk

    n = 1_300_000
    whole_c, whole_chk = run(0, n)
    parts = [run(0, 500_000), run(500_000, 500_000), run(1_000_000, 300_000)]
    for k in ('organic', 'bandit', 'clicks', 'phantom'):
        assert whole_c[k] == sum(p[0][k] for p in parts), k
    assert whole_c['phantom'] == n and whole_c['live'] == 0 and whole_c['log_dropped'] == 0
    assert 95 < (whole_c['organic'] + whole_c['bandit']) / n < 108      # ~1/0.01 events per user
    for i in range(3):
        assert whole_chk[i] % 2 ** 64 == sum(p[1][i] for p in parts) % 2 ** 64


@pytest.mark.parametrize('mode', ['f64', 'fp32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(10, 5), (1000, 20), (4100, 8), (700, 33)])
def test_every_draw_kernel_matches_the_oracle(mode, shape, monkeypatch):
    """The organic-draw implementations (float64 only, fp32 MFMA + certificate, three-way split-bf16
    MFMA + certificate, two-way split-fp16 MFMA + certificate — the default where 3K + 1 <= 64) each
    reproduce the oracle's rows."""
    from oracle import oracle as orc
    monkeypatch.setenv('RECOGYM_DRAW', mode)
    P, K = shape
    cfg = Configuration({**env_1_args, 'random_seed': 100 + P, 'num_products': P, 'K': K})
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX).generate_logs(600)
    rows, cnt = run_sim(cfg, 600)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')},
                         ps_rtol=1e-6, what=f'{mode} {shape}')
    if mode == 'f64':
        assert cnt['exact_draws'] == 0      # counted only for draws handed over by an MFMA kernel
    else:
        assert cnt['exact_draws'] < 0.2 * cnt['organic']


@pytest.mark.parametrize('draw', ['f64', 'f16'])
def test_product_per_lane_float64_kernel_matches_the_oracle(draw, monkeypatch):
    """RECOGYM_EXACT_TILE=1 selects the first float64 resolve kernel (a product per lane, Gamma^T tiles
    in LDS; the only one for K > 64) instead of the user-per-lane one."""
    from oracle import oracle as orc
    monkeypatch.setenv('RECOGYM_EXACT_TILE', '1')
    monkeypatch.setenv('RECOGYM_DRAW', draw)
    cfg = Configuration({**env_1_args, 'random_seed': 321, 'num_products': 1500, 'K': 20})
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX).generate_logs(500)
    rows, cnt = run_sim(cfg, 500)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')},
                         ps_rtol=1e-6, what=f'tile kernel, {draw}')


@pytest.mark.parametrize('shape', [(10000, 20, 40_000), (1000, 20, 300_000)])
def test_certified_draws_equal_float64_draws_at_scale(shape, monkeypatch):
    """The float64-only draw path is the oracle's arithmetic (checked row for row on small runs) and
    does not depend on the run size; the certified fp16-split MFMA path must log exactly the same
    rows on runs far beyond what the CPU oracle can replay (BASELINE config 3's table: ~10^6 draws
    at P = 10 000, and 7.5 * 10^6 at P = 1000 through the fused form of the kernel)."""
    from recogym_amd.sim import Simulator
    P, K, n = shape
    cfg = Configuration({**env_1_args, 'random_seed': 17, 'num_products': P, 'K': K})

    def run(mode):
        monkeypatch.setenv('RECOGYM_DRAW', mode)
        sim = Simulator(cfg, n, device='cuda:0', policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=5,
                        ouc=dict(gu.OUC_DEFAULTS))
        sim.reset_users(1000, n)
        sim.run()
        c = sim.counters()
        rows = sim.raw_log().to(torch.int64)
        assert rows.shape[0] == c['organic'] + c['bandit']
        chk = [(rows[:, i] * (rows[:, 1] + 7) * (rows[:, 0] + 13)).sum().item() for i in range(4)]
        sim.close()
        return c, chk

    fast_c, fast_chk = run('f16')
    ref_c, ref_chk = run('f64')
    assert 0 < fast_c['exact_draws'] < 0.1 * fast_c['organic'] and ref_c['exact_draws'] == 0
    for k in ('organic', 'bandit', 'clicks', 'phantom'):
        assert fast_c[k] == ref_c[k], k
    assert fast_chk == ref_chk


@pytest.mark.parametrize('K', [2, 9, 11, 15, 16, 21, 22, 27, 32, 40, 65])
def test_every_K_class_matches_the_oracle(K):
    """One run per embedding-size class of the draw kernels (fp16 split: N1 = 1..4 with KH = 4 / 10 / 16;
    bf16 split classes up to K = 32; fp32 MFMA beyond; float64 resolve classes KB = 1..16 and the tile
    kernel for K > 64), default kernel selection, rows vs the oracle."""
    from oracle import oracle as orc
    cfg = Configuration({**env_1_args, 'random_seed': 500 + K, 'num_products': 333, 'K': K,
                         'sigma_omega': 0.07})
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX).generate_logs(250)
    rows, cnt = run_sim(cfg, 250)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')},
                         ps_rtol=1e-6, what=f'K={K}')
    assert cnt['live'] == 0
    # a kernel class whose sums are off still matches (the certificate hands the draws to float64) but stops being
    # the fast path: K = 65 took 25 % of its draws in float64 when its LDS tile was shorter than a chunk pair
    assert cnt['exact_draws'] < 0.1 * cnt['organic'], (cnt['exact_draws'], cnt['organic'])


@pytest.mark.parametrize('P', [2, 31, 33, 127, 129, 255, 383, 513, 2049])
def test_product_counts_around_tile_boundaries_match_the_oracle(P):
    """P just below / above the 32-product chunk, the 128-product LDS tile, the 64-product float64
    chunk and the super-chunk granularity, with the headline K = 20 kernel class."""
    from oracle import oracle as orc
    cfg = Configuration({**env_1_args, 'random_seed': 900 + P, 'num_products': P, 'K': 20})
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX).generate_logs(300)
    rows, cnt = run_sim(cfg, 300)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')},
                         ps_rtol=1e-6, what=f'P={P}')


@pytest.mark.parametrize('mode', ['f16', 'bf16', 'fp32'])
@pytest.mark.parametrize('sigma_mu', [30.0, 200.0])
def test_wide_logit_range_re_references_and_still_matches_the_oracle(mode, sigma_mu, monkeypatch):
    """mu_organic spread over hundreds of nats: the running reference of the MFMA kernels has to be
    re-based inside a product sweep (logits far above the first chunk's maximum), exp2 saturates on
    the way, and with sigma_mu = 200 whole super-chunks underflow — whatever the fast path makes of
    it, the certificate must only let float64-identical indices through."""
    from oracle import oracle as orc
    monkeypatch.setenv('RECOGYM_DRAW', mode)
    cfg = Configuration({**env_1_args, 'random_seed': 77, 'num_products': 3000, 'K': 20,
                         'sigma_mu_organic': sigma_mu})
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX).generate_logs(400)
    rows, cnt = run_sim(cfg, 400)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')},
                         ps_rtol=1e-6, what=f'{mode} sigma_mu={sigma_mu}')


@pytest.mark.parametrize('fin', ['1', '0'])
@pytest.mark.parametrize('sigma_mu', [30.0, 200.0])
def test_error_free_sweep_re_references_and_still_matches_the_oracle(sigma_mu, fin, monkeypatch):
    """The same spread of mu_organic at sigma_omega = 0 through run_walk_pipe, whose sweep is k_sweep_xh: its reference moves
    inside the sweep (an integer in the exact accumulator; past |q| ~ 200 the user is no longer eligible for the tight delta),
    the {sum, reference} records of such users are back-filled from the staged super-chunk prefixes at the first move, and
    k_cache_finalize / k_cache_prefix rescale them.  `fin = 0`: the sweep leaves the finalize output to those kernels for
    every user (RECOGYM_FIN_IN_SWEEP=0)."""
    from oracle import oracle as orc
    monkeypatch.delenv('RECOGYM_DRAW', raising=False)
    monkeypatch.setenv('RECOGYM_PIPE_MIN', '256')
    monkeypatch.setenv('RECOGYM_FIN_IN_SWEEP', fin)
    cfg = Configuration({**env_1_args, 'random_seed': 78, 'num_products': 3000, 'K': 20, 'sigma_mu_organic': sigma_mu, 'sigma_omega': 0.0})
    pol = dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=31, ouc=dict(gu.OUC_DEFAULTS))
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol).generate_logs(700)
    rows, cnt = run_sim(cfg, 700, p_click=False, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')}, ps_rtol=1e-12, what=f'xh sigma_mu={sigma_mu} fin={fin}')


def test_repack_and_tail_kernel_do_not_change_the_log_at_scale(monkeypatch):
    """300 000 users (above the 2^18 threshold where the state repack is on by default): the run with
    the repack every 16 steps and the per-user tail kernel must log the same rows as plain lock-step
    to the end without repack."""
    from recogym_amd.sim import Simulator
    cfg = Configuration({**env_1_args, 'random_seed': 23, 'num_products': 100, 'K': 20, 'sigma_omega': 0.05})
    n = 300_000

    def run(repack, tail):
        monkeypatch.setenv('RECOGYM_REPACK', repack)
        monkeypatch.setenv('RECOGYM_TAIL', tail)
        sim = Simulator(cfg, n, device='cuda:0', policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=5,
                        ouc=dict(gu.OUC_DEFAULTS))
        sim.reset_users(0, n)
        sim.run()
        c = sim.counters()
        rows = sim.raw_log().to(torch.int64)
        assert rows.shape[0] == c['organic'] + c['bandit']
        chk = [(rows[:, i] * (rows[:, 1] + 7) * (rows[:, 0] + 13)).sum().item() for i in range(4)]
        sim.close()
        return c, chk

    a_c, a_chk = run('16', '4096')
    b_c, b_chk = run('0', '0')
    for k in ('organic', 'bandit', 'clicks', 'phantom'):
        assert a_c[k] == b_c[k], k
    assert a_chk == b_chk
    # both of these went in run-ahead rounds (k_advance_run: a user's whole bandit run per launch); an event per launch must log
    # the same rows
    monkeypatch.setenv('RECOGYM_RUN_AHEAD', '0')
    c_c, c_chk = run('16', '4096')
    for k in ('organic', 'bandit', 'clicks', 'phantom', 'step'):
        assert a_c[k] == c_c[k], k
    assert a_chk == c_chk


def test_fused_and_sliced_draw_forms_agree_at_scale(monkeypatch):
    """The draw kernel runs in two forms: whole product sweeps with the search fused in (steps with
    >= 1024 user tiles) and product slices + k_draw_search (fewer).  Small oracle-checked runs only
    ever take the second; here both forms are forced on the same 150 000-user run and must produce
    the same log (order-independent checksum of every row) and counters."""
    from recogym_amd.sim import Simulator
    cfg = Configuration({**env_1_args, 'random_seed': 9, 'num_products': 1000, 'K': 20})
    n = 150_000

    def run(slices):
        monkeypatch.setenv('RECOGYM_SLICES', slices)
        sim = Simulator(cfg, n, device='cuda:0', policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=3,
                        ouc=dict(gu.OUC_DEFAULTS))
        sim.reset_users(0, n)
        sim.run()
        c = sim.counters()
        rows = sim.raw_log().to(torch.int64)
        assert rows.shape[0] == c['organic'] + c['bandit']
        chk = [(rows[:, i] * (rows[:, 1] + 7) * (rows[:, 0] + 13)).sum().item() for i in range(4)]
        sim.close()
        return c, chk

    fused_c, fused_chk = run('1')
    sliced_c, sliced_chk = run('4')
    for k in ('organic', 'bandit', 'clicks', 'phantom'):
        assert fused_c[k] == sliced_c[k], k
    assert fused_chk == sliced_chk
    assert fused_c['live'] == 0 and fused_c['hist_overflow'] == 0


def test_full_size_log_invariants():
    """BASELINE config 2 at full size (P=1000, K=20, 1 M users, RandomAgent): the oracle cannot
    run this in seconds, so check the size-independent properties of the reference's log
     on every one of the ~1e8 rows, on the device."""
    from recogym_amd.sim import Simulator
    cfg = Configuration({**env_1_args, 'random_seed': 42, 'num_products': 1000, 'K': 20,
                         'sigma_omega': 0.0})
    n = 1_000_000
    sim = Simulator(cfg, n, device='cuda:0', policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=7)
    sim.reset_users(0, n)
    sim.run()
    cnt = sim.counters()
    log, off = sim.sorted_log()
    log = log.to(torch.int64)
    u, t, code = log[:, 0], log[:, 1], log[:, 2] & 0xFFFFFFFF
    is_b = (code & _abi.RG_EV_BANDIT) != 0
    click = (code & _abi.RG_EV_CLICK) != 0
    phantom = (code & _abi.RG_EV_PHANTOM) != 0
    idx = code & _abi.RG_EV_INDEX_MASK
    rows = log.shape[0]
    assert rows == cnt['organic'] + cnt['bandit'] + cnt['phantom'] and cnt['phantom'] == n
    assert int(off[-1]) == rows
    # users appear in id order, every user's t runs 0, 1, 2, ... without gaps
    first = off[:-1]
    last = off[1:] - 1
    assert bool((u[first] == torch.arange(n, device=u.device)).all())
    assert bool((u[1:] >= u[:-1]).all())
    same = u[1:] == u[:-1]
    assert bool((t[1:][same] == t[:-1][same] + 1).all()) and bool((t[first] == 0).all())
    # first row organic; last row = the phantom bandit row with c = 0; no other phantom rows
    assert bool((~is_b[first]).all())
    assert bool((is_b[last] & phantom[last] & ~click[last]).all())
    assert int(phantom.sum()) == n
    # a click is always followed by an organic row of the same user (abstract.py:180-185)
    cpos = torch.nonzero(click).squeeze(1)
    assert bool((~is_b[cpos + 1] & (u[cpos + 1] == u[cpos])).all())
    assert int(click.sum()) == cnt['clicks']
    # indices in range, organic rows carry no click flag
    assert int(idx.max()) < 1000 and not bool((click & ~is_b).any())
    # statistics of the Markov chain: ~100.6 events per user, 21.8 % organic, CTR ~1.1 %
    ev = cnt['organic'] + cnt['bandit']
    assert 99.0 < ev / n < 102.5
    assert 0.21 < cnt['organic'] / ev < 0.23
    assert 0.009 < cnt['clicks'] / (cnt['bandit'] + cnt['phantom']) < 0.014
    # uniform policy: actions cover the catalogue evenly (chi-square-ish bound)
    a_hist = torch.bincount(idx[is_b], minlength=1000).double()
    assert float((a_hist.max() - a_hist.min()) / a_hist.mean()) < 0.05
    sim.close()


def test_frozen_logreg_with_more_than_256_classes_matches_the_oracle():
    """The wave-cooperative class-score loop walks the classes in blocks of 256 (lane = class, four
    blocks of 64 per pass): 700 classes take three passes with a ragged last one.  Synthetic
    coefficients with deliberate exact ties between classes (first maximum wins, like numpy's argmax)."""
    from oracle import oracle as orc
    P, C = 900, 700
    rng = np.random.RandomState(3)
    classes = np.sort(rng.choice(P, C, replace=False)).astype(np.int32)
    coef_t = rng.standard_normal((P, C))
    coef_t[:, 300] = coef_t[:, 17]              # exact ties across class blocks
    coef_t[:, 699] = coef_t[:, 511]
    intercept = rng.standard_normal(C) * 0.1
    intercept[300] = intercept[17]
    intercept[699] = intercept[511]
    pol = dict(policy=_abi.RG_POLICY_LOGREG_FROZEN, policy_seed=0,
               logreg=dict(coef_t=coef_t, intercept=intercept, classes=classes))
    cfg = Configuration({**env_1_args, 'random_seed': 41, 'num_products': P, 'K': 12})
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(150)
    rows, cnt = run_sim(cfg, 150, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps', 'p_click')},
                         ps_rtol=1e-12, what='logreg 700 classes')
    used = np.unique(rows['a'][rows['z'] == 1])
    assert (used >= classes[256]).any() and (used >= classes[512]).any()     # later class blocks do win


@pytest.mark.parametrize('workload,users', [('c4shard', 262144), ('c3drift', 1000000), ('c3', 1000000), ('c5#1', 262144),
                                            ('wide:K=30:P=20000', 524288), ('wide:K=45:P=20000', 524288), ('wide:K=64:P=5000', 524288)])
def test_two_runs_of_a_bench_shape_give_the_same_log(workload, users):
    """Rows are keyed by (seed, user, event): two runs of the same job must agree row for row whatever the order of their atomics.
    Round 6's wide-K sweep (k_draw_tpw) read the first mu quads of a tile up to three issue slots before the barrier behind which
    that tile is known to have landed — almost always fine, ~300 wrong draws in the 2.8 10^7 of a C4 shard when not; no small
    case and no 2 000-user oracle sample sees a rate of 10^-5, two runs of a chip-filling shape against each other do
    (profiles/r6/determinism_call51.jsonl: 330 rows; call 53: 0)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from recogym_amd.sim import Simulator, default_log_capacity
    if workload.startswith('wide:'):          # the other k-step classes of k_draw_tpw, chip-filling populations
        over = {k: int(v) for k, v in (kv.split('=') for kv in workload.split(':')[1:])}
        cfg = Configuration({**env_1_args, 'random_seed': 42, 'num_products': over['P'], 'K': over['K'], 'sigma_omega': 0.1})
        kw = bench.policy_kwargs('none')
    else:                                     # ("c5#1": the second arm of C5 — the frozen LogReg policy at 10^4 classes)
        wl, _, arm = workload.partition('#')
        cfg = bench.make_config(wl)
        name, kw = bench.arms_of(wl, cfg)[int(arm or 0)]
    sim = Simulator(cfg, users, device='cuda:0', log_capacity=default_log_capacity(cfg, users), **kw)
    logs = []
    for _ in range(3):
        sim.reset_users(0, users)
        sim.run()
        log, _off = sim.sorted_log()
        logs.append(log.clone())
    sim.close()
    for other in logs[1:]:
        assert other.shape == logs[0].shape
        assert int((other != logs[0]).any(dim=1).sum().item()) == 0


@pytest.mark.parametrize('sigma_omega', [0.0, 0.1])
def test_the_highest_user_ids_sort_like_any_other(sigma_omega):
    """User ids up to 2^32 - 1 (rg_sim_reset_users' bound): the ordered log's LDS-tiled scatter keys its table of a tile's users by
    id with 0xFFFFFFFF as the empty key — a tile holding that user takes the plain path — and every row must come out as the
    oracle's (user-major walk at sigma_omega = 0, rounds otherwise)."""
    from oracle import oracle as orc
    cfg = Configuration({**env_1_args, 'random_seed': 9, 'num_products': 40, 'K': 5, 'sigma_omega': sigma_omega})
    n, first = 700, (1 << 32) - 700
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX).generate_logs(n, first_user_id=first)
    rows, cnt = run_sim(cfg, n, first_user=first)
    assert int(rows['u'].astype(np.int64).max()) == (1 << 32) - 1
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps', 'p_click')}, ps_rtol=1e-12, what='highest user ids')


def test_frozen_logreg_steps_with_more_acts_than_the_capped_grids_hold():
    """k_logreg_screen / k_logreg_decide walk a step's act list grid-stride on grids capped at a few blocks per CU (round 6): with
    40 000 users the first steps list ~10^4 acts — 8 x 10^4 (act, class range) items, several per wave — and every row must
    still be the oracle's."""
    from oracle import oracle as orc
    P = C = 64
    rng = np.random.RandomState(11)
    coef_t = rng.standard_normal((P, C)) * 0.3
    intercept = rng.standard_normal(C) * 0.1
    pol = dict(policy=_abi.RG_POLICY_LOGREG_FROZEN, policy_seed=0,
               logreg=dict(coef_t=coef_t, intercept=intercept, classes=np.arange(C, dtype=np.int32)))
    cfg = Configuration({**env_1_args, 'random_seed': 77, 'num_products': P, 'K': 8})
    n = 40000
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol).generate_logs(n)
    rows, cnt = run_sim(cfg, n, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps', 'p_click')},
                         ps_rtol=1e-12, what='logreg, 40 000 users')
    assert cnt['live'] == 0 and cnt['log_dropped'] == 0


@pytest.mark.parametrize('P', [10, 200, 1024])
def test_sampled_frozen_logreg_matches_the_oracle(P):
    """LogregMulticlassIps with select_randomly = True on the device (k_logreg_sample: softmax of the float64 class scores, the
    event's second policy uniform against the cumulative probabilities, ps = the sampled class's probability; the trailing
    row's own draw) against the oracle, which the reference's own log pins (tests/golden/hostpath_logreg_random.npz):
    actions bit for bit, ps to 1e-12."""
    from oracle import oracle as orc
    rng = np.random.RandomState(P)
    coef_t = rng.standard_normal((P, P)) * (1.5 if P <= 200 else 0.6)
    intercept = rng.standard_normal(P) * 0.5
    pol = dict(policy=_abi.RG_POLICY_LOGREG_FROZEN, policy_seed=77,
               logreg=dict(coef_t=coef_t, intercept=intercept, classes=np.arange(P, dtype=np.int32), select_randomly=True))
    cfg = Configuration({**env_1_args, 'random_seed': 600 + P, 'num_products': P, 'K': 8})
    n = 400 if P < 1024 else 150
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(n, 5)
    rows, cnt = run_sim(cfg, n, 5, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps', 'p_click')}, ps_rtol=1e-12, what=f'sampled logreg P={P}')
    assert (rows['phantom'] == want['phantom']).all()
    assert len(np.unique(rows['a'][rows['z'] == 1])) > min(P, 20) // 2          # it does sample


@pytest.mark.parametrize('screen', ['int8', 'fp16', 'fp32', 'fp16_cap3'])
def test_frozen_logreg_at_config_5_scale_matches_the_oracle(screen, monkeypatch):
    """BASELINE config 5's policy shape: one class per product, 4 096 products.  The act screens every class score from
    the half copy of coef^T (RECOGYM_LOGREG=int8: from the 8-bit copy of round 6, q = rint(w / scale) + 128 with scale = wmax / 127 per
    product row — opt-in, measured slower; =fp32: from the fp32 copy), keeps the classes within twice the rounding
    bound of the best and lets float64 scores in scipy's order decide among them.  Small coefficients (N(0, 0.1)) make
    near-ties common; classes duplicated exactly (first maximum wins) and almost exactly (1e-9 apart: far inside the
    fp16 bound, decided by float64) must come out as the oracle's argmax."""
    from oracle import oracle as orc
    if screen == 'fp16_cap3':
        # the screen's scratch holds a row per act of a step (n / 2 + 4096 of them); here only 3: every step that lists more acts
        # sends the rest through k_logreg_acts (fp32 scores, float64 inside the bound)
        monkeypatch.setenv('RECOGYM_LR_PART_CAP', '3')
        screen = 'fp16'
    monkeypatch.setenv('RECOGYM_LOGREG', screen)
    P = C = 4096
    rng = np.random.RandomState(5)
    coef_t = rng.standard_normal((P, C)) * 0.1
    intercept = rng.standard_normal(C) * 0.1
    for a, b in ((10, 3000), (511, 512), (77, 4095)):           # exact duplicates: the smaller class index wins
        coef_t[:, b] = coef_t[:, a]; intercept[b] = intercept[a]
    for a, b in ((20, 2000), (1023, 1024)):                     # near duplicates, the LATER class a hair better
        coef_t[:, b] = coef_t[:, a]; intercept[b] = intercept[a] + 1e-9
    intercept[[10, 3000, 511, 512, 77, 4095, 20, 2000, 1023, 1024]] += 0.6      # ... and often the best
    pol = dict(policy=_abi.RG_POLICY_LOGREG_FROZEN, policy_seed=0,
               logreg=dict(coef_t=coef_t, intercept=intercept, classes=np.arange(C, dtype=np.int32)))
    cfg = Configuration({**env_1_args, 'random_seed': 55, 'num_products': P, 'K': 10})
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(120)
    rows, cnt = run_sim(cfg, 120, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps', 'p_click')},
                         ps_rtol=1e-12, what=f'logreg 4096 classes, {screen} screen')
    used = set(np.unique(rows['a'][rows['z'] == 1]).tolist())
    assert used & {10, 511, 77} and used & {2000, 1024} and not used & {3000, 512, 4095}
    assert cnt['lr_acts'] > 300 and cnt['lr_exact'] > 0 and cnt['lr_rows'] >= cnt['lr_acts']


LDS_EXTRA_CASES = [
    # the (KH, N1) classes of k_draw_tp at tables of >= 4 product tiles (what it serves): (4, 1) K = 3 / 5, (4, 2) K = 6 / 8,
    # (10, 2) K = 10, (10, 3) K = 13, (10, 4) K = 17 / 20; P on and off tile boundaries
    (dict(num_products=600, K=5, random_seed=60), 1500, 20, {}),
    (dict(num_products=700, K=10, random_seed=61, sigma_omega=0.2), 1200, 0, {}),
    (dict(num_products=641, K=13, random_seed=62), 900, 5, dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=7)),
    (dict(num_products=512, K=8, random_seed=63, sigma_omega=0.5), 900, 0, {}),
    (dict(num_products=2000, K=20, random_seed=64, sigma_omega=0.3, sigma_mu_organic=12.0), 500, 0,
     dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=27, ouc=dict(gu.OUC_DEFAULTS))),
    # k_draw_tpw, the wide classes (N1 = 7, 10: K = 30, 45; 13 is oracle case 14 / 3): several super-tiles, P off a tile boundary
    (dict(num_products=5000, K=30, random_seed=65, sigma_omega=0.2), 600, 0, {}),
    (dict(num_products=9001, K=45, random_seed=66), 500, 3, dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=8)),
    (dict(num_products=900, K=3, random_seed=67, sigma_omega=0.3), 1200, 0, {}),
    (dict(num_products=530, K=6, random_seed=68), 1000, 0,
     dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=22, ouc=dict(gu.OUC_DEFAULTS, epsilon=0.25))),
    (dict(num_products=1500, K=8, random_seed=69, prob_leave_organic=0.05, prob_organic_to_bandit=0.5, prob_bandit_to_organic=0.2), 1500, 0, {}),
]


@pytest.mark.parametrize('run_ahead', ['32', '0'])
@pytest.mark.parametrize('case', [3, 4, 6, 7, 14, 'x0', 'x1', 'x2', 'x3', 'x4', 'x5', 'x6', 'x7', 'x8', 'x9'])
def test_lds_search_sweep_matches_the_oracle(case, run_ahead, monkeypatch):
    """k_draw_tp / k_draw_tpw + k_pick (rg_draw_lds.hip: a user's tile prefixes in LDS, the draw's tile a count on them, the product
    inside the tile on the matrix cores, 32 draws of one tile per wave) are the sweep of every UNSLICED step of a run whose draws
    cannot be cached; populations the oracle finishes in seconds take the sliced form, so the unsliced one is forced here
    (RECOGYM_SLICES=1) on every (KH, N1) class they are instantiated for — K <= 20 and the wide classes up to K = 64 — in rounds and
    in lock-step."""
    from oracle import oracle as orc
    from recogym_amd.sim import Simulator
    monkeypatch.setenv('RECOGYM_SLICES', '1')
    monkeypatch.setenv('RECOGYM_RUN_AHEAD', run_ahead)
    monkeypatch.setenv('RECOGYM_TAIL', '0')            # (the tail kernel would take these small populations from the first poll on)
    over, n_users, n_org, pol = CASES[case] if isinstance(case, int) else LDS_EXTRA_CASES[int(case[1:])]
    cfg = Configuration({**env_1_args, **over})
    probe = Simulator(cfg, 64, device='cuda:0', **pol)
    assert probe.get_option('sweep_lds_kernel') == 1, 'k_draw_tp does not serve this configuration'
    probe.close()
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(n_users, n_org)
    rows, cnt = run_sim(cfg, n_users, n_org, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps', 'p_click')},
                         ps_rtol=1e-12, what=f'lds case {case}')
    assert (rows['phantom'] == want['phantom']).all()
    oc = want_env.counters()
    assert (cnt['organic'], cnt['bandit'], cnt['clicks'], cnt['phantom']) == (oc['organic'], oc['bandit'], oc['clicks'], oc['phantom'])
    # most draws are certified by the kernel itself (the float64 resolve is the exception, not the path)
    assert cnt['exact_draws'] < 0.2 * cnt['organic'] + 50, (cnt['exact_draws'], cnt['organic'])


@pytest.mark.parametrize('mode', ['f16', 'f16_lds', 'bf16', 'fp32'])
@pytest.mark.parametrize('shape', [(10000, 20), (3000, 20), (1500, 40)])
def test_certificate_is_sound_for_uniforms_next_to_cdf_boundaries(mode, shape, monkeypatch):
    """Adversarial check of the margin certificate.  The uniform of every user's draw is
    placed next to a boundary of ITS float64 cdf: u = cdf[b] * (1 +- eps), eps from 1e-9 (closer than any
    fp32 sum can resolve) to 3e-3, through the test hook rg_sim_debug_set_uniforms.  Then
      * every draw the matrix-core kernel certified must equal the float64 decision (soundness);
      * nothing within 5e-7 of a boundary may be certified (the certificate's floor is 2^-20 of the total — the fp32
        roundings of the stored prefixes — next to delta times the masses before AND behind the boundary; the COMPUTED
        boundary a certified draw keeps that distance from is itself off the true one by the actual roundings);
      * draws >= 1e-3 away from both neighbouring boundaries mostly are certified (the test is not
        vacuous), and the logged index of EVERY user equals the float64 one (uncertified draws are
        resolved by the float64 kernels)."""
    import ctypes as C
    from recogym_amd.envs.static_params import draw_tables
    from recogym_amd.sim import Simulator
    if mode == 'f16_lds':          # the unsliced sweep whose search runs on tile prefixes in LDS (k_draw_tp; K <= 20)
        if shape[1] > 20:
            pytest.skip('k_draw_tp serves K <= 20')
        monkeypatch.setenv('RECOGYM_SLICES', '1')
        mode = 'f16'
    monkeypatch.setenv('RECOGYM_DRAW', mode)
    P, K = shape
    n = 4096
    cfg = Configuration({**env_1_args, 'random_seed': 1234 + P + K, 'num_products': P, 'K': K})
    gamma, mu_o, _, _ = draw_tables(cfg)
    rng = np.random.RandomState(99)
    omega = rng.standard_normal((n, K))
    # the reference's arithmetic (reco_env_v1.py:119-128) in float64
    logits = omega @ gamma.T + mu_o.reshape(1, -1)
    logits -= logits.max(axis=1, keepdims=True)
    e = np.exp(logits)
    prob = e / e.sum(axis=1, keepdims=True)
    cdf = np.cumsum(prob, axis=1)
    cdf /= cdf[:, -1:]
    # a boundary per user, drawn by mass (so that heavy and light products both occur), then u beside it
    b = np.array([np.searchsorted(cdf[i], rng.random_sample(), 'right') for i in range(n)])
    b = np.clip(b, 0, P - 2)
    eps = 10.0 ** rng.uniform(-9, -2.5, n)
    sign = rng.choice([-1.0, 1.0], n)
    u = np.clip(cdf[np.arange(n), b] * (1.0 + sign * eps), 0.0, np.nextafter(1.0, 0.0))
    want_v = np.array([np.searchsorted(cdf[i], u[i], 'right') for i in range(n)])
    # distance of u to its two neighbouring boundaries, relative to u
    lo = np.where(want_v > 0, cdf[np.arange(n), np.maximum(want_v - 1, 0)], -np.inf)
    hi = np.where(want_v < P - 1, cdf[np.arange(n), np.minimum(want_v, P - 1)], np.inf)     # (no boundary behind the last product)
    margin = np.minimum(u - lo, hi - u) / np.maximum(u, 1e-300)

    sim = Simulator(cfg, n, device='cuda:0')
    sim.reset_users(0, n)
    d_om = torch.from_numpy(omega).to('cuda:0')
    d_u = torch.from_numpy(u).to('cuda:0')
    _abi.check(sim.lib.rg_sim_debug_set_omega(sim._h, d_om.data_ptr(), sim._stream()), 'debug_set_omega')
    _abi.check(sim.lib.rg_sim_debug_set_uniforms(sim._h, d_u.data_ptr()), 'debug_set_uniforms')
    sim.step()
    flags = torch.zeros(n, dtype=torch.uint8, device='cuda:0')
    _abi.check(sim.lib.rg_sim_debug_uncertified(sim._h, flags.data_ptr(), sim._stream()), 'debug_uncertified')
    torch.cuda.synchronize()
    _abi.check(sim.lib.rg_sim_debug_set_uniforms(sim._h, None), 'debug_set_uniforms')
    uncert = flags.cpu().numpy().astype(bool)
    raw = sim.log[:n].cpu().numpy().view(np.uint32)
    got_v = np.zeros(n, dtype=np.int64)
    got_v[raw[:, 0]] = raw[:, 2] & _abi.RG_EV_INDEX_MASK          # step 0: one organic row per user
    sim.close()
    cert = ~uncert
    # a uniform closer than 3e-14 (relative) to a boundary may legitimately fall either way between two
    # float64 evaluations that differ in summation order; none is placed there (eps >= 1e-9)
    bad = np.flatnonzero(cert & (got_v != want_v))
    assert bad.size == 0, (f'{bad.size} CERTIFIED draws differ from float64; first: user {bad[0]} got {got_v[bad[0]]} '
                           f'want {want_v[bad[0]]} margin {margin[bad[0]]:.3e}')
    assert not cert[margin < 5e-7].any(), 'a draw within 5e-7 of a cdf boundary was certified'
    assert (margin < 5e-7).sum() > 200 and cert.sum() > 200
    far = margin > 1e-3
    assert far.sum() > 20 and cert[far].mean() > 0.9
    # float64 resolve of the rest.  Where the neighbouring products' masses are below float64 resolution of the
    # running sum (margin < 1e-12) two float64 evaluations with different summation trees may differ: excluded
    clear = margin > 1e-12
    bad = np.flatnonzero(clear & (got_v != want_v))
    assert bad.size == 0, (f'{bad.size} draws differ from float64; first: user {bad[0]} got {got_v[bad[0]]} want '
                           f'{want_v[bad[0]]} margin {margin[bad[0]]:.3e} certified {cert[bad[0]]}')
    assert clear.mean() > 0.95


@pytest.mark.parametrize('walk', ['default', 'solo_only', 'k_walk', 'pipe', 'pipe_solo'])
@pytest.mark.parametrize('shape', [(10000, 20), (3000, 20), (640, 7)])
def test_walk_certificate_is_sound_for_uniforms_next_to_cdf_boundaries(shape, walk, monkeypatch):
    """The same adversarial placement for the user-major walk (sigma_omega = 0): EVERY organic draw of a user gets the
    uniform placed next to a boundary of the user's float64 cdf, eps from 1e-9 to 3e-3 — two thirds of them inside the band
    the independent-error form of the certificate rejected and the correlated form (cert_correlated) may accept.  Whatever
    decided a draw (the search's certificate, the memo built from it, the float64-anchored certificate, the float64
    pick), every logged product must be float64's.  `solo_only`: hand-over at 64 live lanes, so the wave-per-user kernel
    takes nearly all users; `k_walk`: round 2's kernel; `pipe` / `pipe_solo`: run_walk_pipe (RECOGYM_PIPE_MIN lowered), whose
    sweep is k_sweep_xh — the error-free leading accumulator, a delta ~8x smaller and rho = 2^-23: the band these uniforms are
    placed in is the one that certificate newly ACCEPTS."""
    import ctypes as C
    from recogym_amd.envs.static_params import draw_tables
    from recogym_amd.sim import Simulator
    for k in ('RECOGYM_DRAW', 'RECOGYM_WALK', 'RECOGYM_WALK_HANDOVER', 'RECOGYM_PIPE_MIN', 'RECOGYM_XH'):
        monkeypatch.delenv(k, raising=False)
    if walk in ('solo_only', 'pipe_solo'):
        monkeypatch.setenv('RECOGYM_WALK_HANDOVER', '64')
    if walk in ('pipe', 'pipe_solo'):
        monkeypatch.setenv('RECOGYM_PIPE_MIN', '256')
    if walk == 'k_walk':
        monkeypatch.setenv('RECOGYM_WALK', '1')
    P, K = shape
    n = 4096
    cfg = Configuration({**env_1_args, 'random_seed': 4321 + P + K, 'num_products': P, 'K': K, 'sigma_omega': 0.0})
    gamma, mu_o, _, _ = draw_tables(cfg)
    rng = np.random.RandomState(7)
    omega = rng.standard_normal((n, K))
    logits = omega @ gamma.T + mu_o.reshape(1, -1)
    logits -= logits.max(axis=1, keepdims=True)
    e = np.exp(logits)
    cdf = np.cumsum(e / e.sum(axis=1, keepdims=True), axis=1)
    cdf /= cdf[:, -1:]
    b = np.clip(np.array([np.searchsorted(cdf[i], rng.random_sample(), 'right') for i in range(n)]), 0, P - 2)
    eps = 10.0 ** rng.uniform(-9, -2.5, n)
    sign = rng.choice([-1.0, 1.0], n)
    u = np.clip(cdf[np.arange(n), b] * (1.0 + sign * eps), 0.0, np.nextafter(1.0, 0.0))
    want_v = np.array([np.searchsorted(cdf[i], u[i], 'right') for i in range(n)])
    lo = np.where(want_v > 0, cdf[np.arange(n), np.maximum(want_v - 1, 0)], -np.inf)
    hi = np.where(want_v < P - 1, cdf[np.arange(n), np.minimum(want_v, P - 1)], np.inf)     # (no boundary behind the last product)
    margin = np.minimum(u - lo, hi - u) / np.maximum(u, 1e-300)

    sim = Simulator(cfg, n, device='cuda:0')
    sim.reset_users(0, n)
    d_om = torch.from_numpy(omega).to('cuda:0')
    d_u = torch.from_numpy(u).to('cuda:0')
    _abi.check(sim.lib.rg_sim_debug_set_omega(sim._h, d_om.data_ptr(), sim._stream()), 'debug_set_omega')
    _abi.check(sim.lib.rg_sim_debug_set_uniforms(sim._h, d_u.data_ptr()), 'debug_set_uniforms')
    sim.run()
    torch.cuda.synchronize()
    _abi.check(sim.lib.rg_sim_debug_set_uniforms(sim._h, None), 'debug_set_uniforms')
    rows = sim.rows()
    c = sim.counters()
    sim.close()
    org = rows['z'] == 0
    uu, vv = rows['u'][org].astype(np.int64), rows['v'][org].astype(np.int64)
    assert org.sum() == c['organic'] and np.unique(uu).size == n
    clear = margin > 1e-12
    bad = np.flatnonzero(clear[uu] & (vv != want_v[uu]))
    assert bad.size == 0, (f'{bad.size} organic rows differ from float64; first: user {uu[bad[0]]} got {vv[bad[0]]} want '
                           f'{want_v[uu[bad[0]]]} margin {margin[uu[bad[0]]]:.3e}')
    assert clear.mean() > 0.95
    # not vacuous: most users' draws were decided by the fp32 certificate (float64 took fewer draws than users whose
    # margin is below 3e-4, the widest band either form leaves)
    assert 0 < c['exact_draws'] and c['exact_sweeps'] < (margin < 3e-4).sum() + 64


@pytest.mark.parametrize('variant', ['default', 'fastclick', 'k_walk', 'sliced', 'nowalk', 'repack', 'lockstep', 'nowalk_sliced'])
@pytest.mark.parametrize('shape', [(129, 9, 1500, 40), (2049, 20, 500, 0), (10000, 20, 150, 0), (33, 3, 2500, 0)])
def test_sigma_omega_zero_sum_cache_matches_the_oracle(shape, variant, monkeypatch):
    """sigma_omega = 0 (BASELINE configs 2 and 3): a user's omega never changes, so the exp-sums of its first
    product sweep are kept per user and every later draw is only the search over them (same certificate,
    same float64 resolve, whose sums are also kept per user).  Rows vs the oracle, with the state repack
    forced (the cache is indexed by user, not by slot), in lock-step to the end, and with step 0 taken by
    the product-sliced form of the sweep."""
    from oracle import oracle as orc
    P, K, n, n_org = shape
    # 'default' / 'sliced': the user-major walk (k_walk) after a fused / product-sliced sweep; the others keep the
    # lock-step loop over the cache (k_draw_cached), which also serves the step-by-step API
    if variant in ('nowalk', 'repack', 'lockstep', 'nowalk_sliced'):
        monkeypatch.setenv('RECOGYM_WALK', '0')
    if variant == 'repack':
        monkeypatch.setenv('RECOGYM_REPACK_MIN', '1')
        monkeypatch.setenv('RECOGYM_REPACK', '3')
    if variant == 'lockstep':
        monkeypatch.setenv('RECOGYM_TAIL', '0')
    if variant in ('sliced', 'nowalk_sliced'):
        monkeypatch.setenv('RECOGYM_SLICES', '4')
    if variant == 'k_walk':          # round 2's walk kernel (it still serves K > 32 and the dense forms of the policy)
        monkeypatch.setenv('RECOGYM_WALK', '1')
    pol = dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=31, ouc=dict(gu.OUC_DEFAULTS))
    cfg = Configuration({**env_1_args, 'random_seed': 700 + P, 'num_products': P, 'K': K, 'sigma_omega': 0.0})
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(n, n_org)
    # 'fastclick': without the click-probability export the walk decides the clicks in fp32 wherever that is certain
    rows, cnt = run_sim(cfg, n, n_org, **pol, **(dict(p_click=False) if variant == 'fastclick' else {}))
    cols = ('u', 't', 'z', 'v', 'a', 'c', 'ps') + (() if variant == 'fastclick' else ('p_click',))
    gu.assert_rows_equal(rows, {k: want[k] for k in cols},
                         ps_rtol=1e-12, what=f'sigma0 cache {shape} {variant}')
    assert (rows['phantom'] == want['phantom']).all()
    assert cnt['exact_sweeps'] <= cnt['exact_draws'] and cnt['live'] == 0


def test_sum_cache_does_not_change_the_log_at_scale(monkeypatch):
    """200 000 users, P = 2 000, sigma_omega = 0: with the per-user cache (default) and without it
    (RECOGYM_CACHE=0: every draw sweeps all products) the log is the same — order-independent checksum of
    every row and the counters — and float64 sweeps are shared between the draws of a user."""
    from recogym_amd.sim import Simulator
    cfg = Configuration({**env_1_args, 'random_seed': 19, 'num_products': 2000, 'K': 20, 'sigma_omega': 0.0})
    n = 200_000

    def run(cache):
        monkeypatch.setenv('RECOGYM_CACHE', cache)
        sim = Simulator(cfg, n, device='cuda:0', policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=5,
                        ouc=dict(gu.OUC_DEFAULTS))
        sim.reset_users(0, n)
        sim.run()
        c = sim.counters()
        rows = sim.raw_log().to(torch.int64)
        assert rows.shape[0] == c['organic'] + c['bandit']
        chk = [(rows[:, i] * (rows[:, 1] + 7) * (rows[:, 0] + 13)).sum().item() for i in range(4)]
        sim.close()
        return c, chk

    on_c, on_chk = run('1')
    off_c, off_chk = run('0')
    for k in ('organic', 'bandit', 'clicks', 'phantom'):
        assert on_c[k] == off_c[k], k
    assert on_chk == off_chk
    # (round 5: the cached run's sweep is k_sweep_xh — a certificate ~10x tighter than the lock-step sweep's: fewer draws go to float64)
    assert 0 < on_c['exact_draws'] <= 1.05 * off_c['exact_draws']
    assert off_c['exact_sweeps'] == off_c['exact_draws'] and 0 < on_c['exact_sweeps'] < on_c['exact_draws']


def test_user_major_walk_equals_the_lock_step_loop_at_scale(monkeypatch):
    """sigma_omega = 0, 300 000 users, P = 1 000, OrganicUserEventCounter in the loop: the user-major walk
    (default: sweep -> k_walk -> batched float64 sums of the parked users -> k_walk) logs exactly the rows of
    the lock-step loop (RECOGYM_WALK=0) — order-independent checksum of every raw row and the counters; the
    sorted logs are identical row for row on a prefix of the users."""
    from recogym_amd.sim import Simulator
    cfg = Configuration({**env_1_args, 'random_seed': 29, 'num_products': 1000, 'K': 20, 'sigma_omega': 0.0})
    n = 300_000

    def run(walk):
        monkeypatch.setenv('RECOGYM_WALK', walk)
        sim = Simulator(cfg, n, device='cuda:0', policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=5,
                        ouc=dict(gu.OUC_DEFAULTS))
        sim.reset_users(7, n)
        sim.run()
        c = sim.counters()
        rows = sim.raw_log().to(torch.int64)
        assert rows.shape[0] == c['organic'] + c['bandit']
        chk = [(rows[:, i] * (rows[:, 1] + 7) * (rows[:, 0] + 13)).sum().item() for i in range(4)]
        srt, off = sim.sorted_log()
        head = srt[:int(off[2000])].cpu().numpy().copy()
        sim.close()
        return c, chk, head

    w_c, w_chk, w_head = run('1')
    l_c, l_chk, l_head = run('0')
    for k in ('organic', 'bandit', 'clicks', 'phantom', 'live', 'hist_overflow', 'log_dropped'):
        assert w_c[k] == l_c[k], k
    assert w_chk == l_chk
    assert np.array_equal(w_head, l_head)
    assert w_c['step'] == l_c['step'] and 0 < w_c['exact_sweeps'] <= w_c['exact_draws']


def _ff(x):
    """reco_env_v1.py:32-41 in float64."""
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    return sig(5.0 * sig(2.0 * sig(0.3 * x) - 2.0) - 6.0)


@pytest.mark.parametrize('shape', [(10000, 20, 1.0), (1000, 20, 3.0), (300, 7, 1.0), (64, 33, 2.0)])
def test_fp32_click_decision_is_sound_for_uniforms_next_to_the_click_threshold(shape):
    """Adversarial check of the walk's fp32 click decision (click_decide32): the uniform of 2^18
    (user, action) pairs is placed at (1 - ctr) +- eps, eps from 1e-12 to 1e-3, through rg_sim_debug_click_decisions.
      * every click the fp32 form DECIDED equals the float64 decision (reco_env_v1.py:104-116 in numpy float64);
      * nothing within 1e-6 of the threshold is decided in fp32 (its error budget is < 2e-7, its margin >= 2e-5);
      * beyond 3e-4 nearly everything is decided (the test is not vacuous);
      * the device's own float64 decision equals numpy's wherever the uniform is further than 1e-13 from the threshold."""
    from recogym_amd.envs.static_params import draw_tables
    from recogym_amd.sim import Simulator
    P, K, scale = shape
    n = 1 << 18
    cfg = Configuration({**env_1_args, 'random_seed': 4321 + P + K, 'num_products': P, 'K': K, 'sigma_omega': 0.0})
    _, _, beta, mu_b = draw_tables(cfg)
    mu_b = np.asarray(mu_b, dtype=np.float64).reshape(-1)
    rng = np.random.RandomState(7)
    omega = rng.standard_normal((n, K)) * scale
    act = rng.randint(0, P, n)
    x = np.einsum('ik,ik->i', beta[act], omega) + mu_b[act]
    ctr = _ff(x)
    p0 = 1.0 - ctr
    thr = p0 / (p0 + ctr)                                    # what choice([0, 1], p=[1 - ctr, ctr]) compares u with
    eps = 10.0 ** rng.uniform(-12, -3, n)
    u = np.clip(thr + rng.choice([-1.0, 1.0], n) * eps, 0.0, np.nextafter(1.0, 0.0))
    want = thr <= u
    dist = np.abs(u - thr)

    sim = Simulator(cfg, n, device='cuda:0')
    sim.reset_users(0, n)
    d_om = torch.from_numpy(omega).to('cuda:0')
    d_a = torch.from_numpy(act.astype(np.int32)).to('cuda:0')
    d_u = torch.from_numpy(u).to('cuda:0')
    out = torch.zeros(n, dtype=torch.uint8, device='cuda:0')
    _abi.check(sim.lib.rg_sim_debug_set_omega(sim._h, d_om.data_ptr(), sim._stream()), 'debug_set_omega')
    _abi.check(sim.lib.rg_sim_debug_click_decisions(sim._h, d_a.data_ptr(), d_u.data_ptr(), out.data_ptr(), sim._stream()),
               'debug_click_decisions')
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    sim.close()
    decided, c32, c64 = (o & 1).astype(bool), ((o >> 1) & 1).astype(bool), ((o >> 2) & 1).astype(bool)
    bad = np.flatnonzero(decided & (c32 != want))
    assert bad.size == 0, (f'{bad.size} fp32-DECIDED clicks differ from float64; first: pair {bad[0]} distance '
                           f'{dist[bad[0]]:.3e} x {x[bad[0]]:.3f}')
    assert not decided[dist < 1e-6].any(), 'a click within 1e-6 of its threshold was decided in fp32'
    assert (dist < 1e-6).sum() > 10000 and decided.sum() > 10000
    assert decided[dist > 3e-4].mean() > 0.99 and (dist > 3e-4).sum() > 1000
    clear = dist > 1e-13
    assert (c64[clear] == want[clear]).all() and clear.mean() > 0.9


@pytest.mark.parametrize('P', [5000, 70000])
def test_ouc_integer_prefix_walk_is_sound_for_uniforms_next_to_count_boundaries(P):
    """Adversarial check of the OrganicUserEventCounter act's integer fast path (policy_act): view
    histories of 1 .. 200 distinct products (one to thirteen 16-entry lines) are written through
    rg_sim_debug_set_history and the action uniform of every user is placed at cdf[b] (1 +- eps) of ITS float64 cdf
    (organic_user_count.py:45-96: p = counts / sum, cumsum, / last, searchsorted 'right'), eps from 0 (exactly on a
    boundary) and 1e-16 to 1e-4.
      * every action and propensity equals the reference arithmetic, integer-decided or not;
      * nothing within 1e-11 (relative) of a boundary is decided by integers (the band is 2^-36 = 1.5e-11);
      * beyond 1e-10 everything is."""
    from recogym_amd.sim import Simulator
    n, stride = (1 << 16) if P <= 5000 else (1 << 13), 208
    cfg = Configuration({**env_1_args, 'random_seed': 99, 'num_products': P, 'K': 4, 'sigma_omega': 0.0})
    rng = np.random.RandomState(11)
    nd = np.where(rng.rand(n) < 0.5, rng.randint(1, 16, n), rng.randint(16, 201, n)).astype(np.uint32)
    prod = np.zeros((n, stride), dtype=np.uint32)
    cnt = np.zeros((n, stride), dtype=np.uint32)
    u1 = np.zeros(n)
    want_a = np.zeros(n, dtype=np.int64)
    want_ps = np.zeros(n)
    rel = np.zeros(n)
    for i in range(n):
        k = int(nd[i])
        pr = np.sort(rng.choice(P, k, replace=False))
        c = np.where(rng.rand(k) < 0.2, rng.randint(1, 400, k), rng.randint(1, 6, k))
        prod[i, :k] = pr
        cnt[i, :k] = c
        f = np.zeros(P)
        f[pr] = c
        p = f / np.sum(f)                                    # the reference's dense vector
        cdf = p.cumsum()
        cdf /= cdf[-1]
        b = pr[rng.randint(0, k)]                            # a boundary that exists: the end of a viewed product
        e = 0.0 if rng.rand() < 0.1 else 10.0 ** rng.uniform(-16, -4)
        u1[i] = min(max(cdf[b] * (1.0 + rng.choice([-1.0, 1.0]) * e), 0.0), np.nextafter(1.0, 0.0))
        a = int(cdf.searchsorted(u1[i], side='right'))
        a = min(a, P - 1)
        want_a[i], want_ps[i] = a, p[a]
        near = cdf[pr]                                       # distinct boundary values
        rel[i] = np.min(np.abs(near - u1[i]) / np.maximum(near, 1e-300))
    sim = Simulator(cfg, n, device='cuda:0', policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=3,
                    ouc=dict(gu.OUC_DEFAULTS))
    sim.reset_users(0, n)
    dev = 'cuda:0'
    d_nd, d_p, d_c = (torch.from_numpy(x.view(np.int32)).to(dev) for x in (nd, prod, cnt))
    d_u = torch.from_numpy(u1).to(dev)
    got_a = torch.zeros(n, dtype=torch.int32, device=dev)
    got_ps = torch.zeros(n, dtype=torch.float64, device=dev)
    flags = torch.zeros(n, dtype=torch.uint8, device=dev)
    _abi.check(sim.lib.rg_sim_debug_set_history(sim._h, d_nd.data_ptr(), d_p.data_ptr(), d_c.data_ptr(), stride, sim._stream()),
               'debug_set_history')
    _abi.check(sim.lib.rg_sim_debug_ouc_acts(sim._h, d_u.data_ptr(), got_a.data_ptr(), got_ps.data_ptr(), flags.data_ptr(),
                                             sim._stream()), 'debug_ouc_acts')
    torch.cuda.synchronize()
    ga, gp, fl = got_a.cpu().numpy().astype(np.int64), got_ps.cpu().numpy(), flags.cpu().numpy().astype(bool)
    sim.close()
    bad = np.flatnonzero(ga != want_a)
    assert bad.size == 0, (f'{bad.size} actions differ; first: user {bad[0]} got {ga[bad[0]]} want {want_a[bad[0]]} '
                           f'rel {rel[bad[0]]:.3e} integer-decided {fl[bad[0]]} nd {nd[bad[0]]}')
    np.testing.assert_allclose(gp, want_ps, rtol=1e-15, atol=0)
    assert not fl[rel < 1e-11].any(), 'an act within 1e-11 of a count boundary was decided by integers'
    assert (rel < 1e-11).sum() > n // 20
    assert fl[rel > 1e-10].all() and (rel > 1e-10).sum() > n // 20


def _bench_line(argv, capsys, monkeypatch):
    import json, sys
    import bench
    monkeypatch.setattr(sys, 'argv', ['bench.py'] + argv)
    capsys.readouterr()
    bench.main()
    out = capsys.readouterr().out.strip().splitlines()
    return json.loads(out[-1])


@pytest.mark.parametrize('workload,users', [('c3', 300_000), ('c3drift', 60_000)])
def test_bench_shards_of_a_strongly_scaled_job_add_up_to_the_whole_run(workload, users, capsys, monkeypatch):
    """SURVEY.md §8e's G = 1 vs G = 8 digest, executed as shards on one device: `bench.py --scaling strong --shard R/W`
    simulates rank R's id range of a W-rank job.  For W = 2 and W = 4 the order-independent row checksums of the shards
    add up (mod 2^64) to the checksum of the whole run, and so do the counters the ranks all-reduce."""
    base = ['--workload', workload, '--users', str(users), '--scaling', 'strong', '--steps', '1', '--warmup', '0',
            '--no-cpu-baseline', '--no-drift-line', '--no-materialise', '--digest']
    whole = _bench_line(base, capsys, monkeypatch)
    assert whole['config']['users_per_gpu'] == users
    for W in (2, 4):
        dig = [0] * len(whole['digest'][0])         # (u, t, code, ps bits, float64 ps bits), phantom rows included
        tot = dict(organic=0, bandit=0, clicks=0, phantom=0)
        n = 0
        for r in range(W):
            line = _bench_line(base + ['--shard', f'{r}/{W}'], capsys, monkeypatch)
            n += line['config']['users_per_gpu']
            for i in range(len(dig)):
                dig[i] = (dig[i] + line['digest'][0][i]) % (1 << 64)
            for k in tot:
                tot[k] += line['totals'][k]
        assert n == users
        assert tot == whole['totals'], (W, tot, whole['totals'])
        assert dig == whole['digest'][0], (W, dig, whole['digest'][0])


@pytest.mark.parametrize('policy', ['uniform', 'random', 'ouc', 'table'])
@pytest.mark.parametrize('handover', ['64', '1'])
def test_last_round_wave_per_user_kernel_matches_the_oracle(policy, handover, monkeypatch):
    """k_walk_solo (the last round of the sigma_omega = 0 walk: a wave per user, a lane per consecutive event of the user's
    current run) for every policy the walk serves.  RECOGYM_WALK_HANDOVER=64 makes every wave hand its users over as soon
    as the queue is empty, so that most of a small run's events are emitted by the solo kernel — organic runs (memo, search,
    float64 pick of the parked draw, view-history insertions in a row), bandit runs cut by clicks and transitions, phantom
    rows, warm-up users; =1 leaves it only the very last users.  Rows vs the oracle, bit for bit."""
    from oracle import oracle as orc
    monkeypatch.setenv('RECOGYM_WALK_HANDOVER', handover)
    P, K, n, n_org = 700, 12, 1500, 30
    cfg = Configuration({**env_1_args, 'random_seed': 900 + len(policy), 'num_products': P, 'K': K, 'sigma_omega': 0.0})
    rng = np.random.RandomState(4)
    pol = {'uniform': {},
           'random': dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=77),
           'ouc': dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=31, ouc=dict(gu.OUC_DEFAULTS)),
           'table': dict(policy=_abi.RG_POLICY_LAST_VIEW_TABLE, policy_seed=0, policy_table=rng.randint(0, P, P).astype(np.int32),
                         policy_ps=rng.rand(P))}[policy]
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(n, n_org)
    rows, cnt = run_sim(cfg, n, n_org, p_click=False, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')},
                         ps_rtol=1e-5 if policy == 'table' else 1e-12, what=f'solo {policy} handover {handover}')
    assert (rows['phantom'] == want['phantom']).all()
    oc = want_env.counters()
    assert (cnt['organic'], cnt['bandit'], cnt['clicks'], cnt['phantom']) == (oc['organic'], oc['bandit'], oc['clicks'], oc['phantom'])
    assert cnt['live'] == 0 and cnt['log_dropped'] == 0 and cnt['hist_overflow'] == 0


def test_memo_and_anchored_certificate_carry_the_walk_at_scale(monkeypatch):
    """400 000 users, P = 10 000, sigma_omega = 0, OrganicUserEventCounter: most organic draws are answered by the per-user
    memo of certified draws, nearly all draws the plain certificate rejects by the float64-anchored one, a sliver by the
    float64 pick — and the log is the lock-step float64-only path's (order-independent checksum of every row)."""
    from recogym_amd.sim import Simulator
    cfg = Configuration({**env_1_args, 'random_seed': 23, 'num_products': 10000, 'K': 20, 'sigma_omega': 0.0})
    n = 400_000
    pol = dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=5, ouc=dict(gu.OUC_DEFAULTS))

    def run(env):
        for k in ('RECOGYM_DRAW', 'RECOGYM_WALK'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sim = Simulator(cfg, n, device='cuda:0', **pol)
        sim.reset_users(0, n)
        sim.run()
        c, dig = sim.counters(), sim.log_digest()
        sim.close()
        return c, dig

    c, dig = run({})
    c64, dig64 = run({'RECOGYM_DRAW': 'f64', 'RECOGYM_WALK': '0'})
    assert dig == dig64
    for k in ('organic', 'bandit', 'clicks', 'phantom'):
        assert c[k] == c64[k], k
    assert c['memo_hits'] > 0.5 * c['organic']                      # the memo answers most draws
    # the anchors take the rejected draws of rounds 1-2 (97 % at 10 M users); the wave-per-user last round — a larger
    # share of a small run — goes to the float64 pick directly
    assert c['anchored'] > 0.5 * c['exact_draws'] > 0
    assert c['exact_draws'] < 0.05 * c['organic'] and 0 < c['exact_sweeps'] < n


@pytest.mark.parametrize('form', ['g1', 'g3_one_stream', 'g4_two_streams', 'g4_three_streams', 'g5_handover64', 'g4_small_grids'])
@pytest.mark.parametrize('policy', ['uniform', 'ouc'])
def test_pipelined_walk_matches_the_oracle(policy, form, monkeypatch):
    """run_walk_pipe — the sigma_omega = 0 run as a pipeline over user groups (sweep -> finalize -> round 1 on one stream, float64
    batch -> prefixes -> round 2 of the group before on a second, every list length read on the device, one last round) — at a
    size the oracle replays: RECOGYM_PIPE_MIN lowers the group size from 131 072 users to 256.  Group counts that do and do not
    divide the users, one / two / three streams, every wave handing over at once, grids smaller than the device.  Rows vs the
    oracle, bit for bit."""
    from oracle import oracle as orc
    env = {'g1': dict(RECOGYM_PIPE='1'),
           'g3_one_stream': dict(RECOGYM_PIPE='3', RECOGYM_PIPE_MODE='0'),
           'g4_two_streams': dict(RECOGYM_PIPE='4', RECOGYM_PIPE_MODE='1'),
           'g4_three_streams': dict(RECOGYM_PIPE='4', RECOGYM_PIPE_MODE='2'),
           'g5_handover64': dict(RECOGYM_PIPE='5', RECOGYM_PIPE_MODE='1', RECOGYM_WALK_HANDOVER='64'),
           'g4_small_grids': dict(RECOGYM_PIPE='4', RECOGYM_PIPE_MODE='2', RECOGYM_PIPE_OCC1='1', RECOGYM_PIPE_OCC2='1', RECOGYM_PIPE_XBLOCKS='3')}[form]
    monkeypatch.setenv('RECOGYM_PIPE_MIN', '256')
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    P, K, n, n_org = 2500, 20, 3100, 40
    cfg = Configuration({**env_1_args, 'random_seed': 4100 + len(form), 'num_products': P, 'K': K, 'sigma_omega': 0.0})
    pol = {'uniform': {},
           'ouc': dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=31, ouc=dict(gu.OUC_DEFAULTS))}[policy]
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(n, n_org)
    rows, cnt = run_sim(cfg, n, n_org, p_click=False, **pol)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')}, ps_rtol=1e-12, what=f'pipe {policy} {form}')
    assert (rows['phantom'] == want['phantom']).all()
    oc = want_env.counters()
    assert (cnt['organic'], cnt['bandit'], cnt['clicks'], cnt['phantom']) == (oc['organic'], oc['bandit'], oc['clicks'], oc['phantom'])
    assert cnt['live'] == 0 and cnt['log_dropped'] == 0 and cnt['hist_overflow'] == 0
    assert cnt['exact_sweeps'] > 0            # some users went through the float64 batch and round 2


@pytest.mark.parametrize('form', ['helpers0', 'helpers1', 'helpers7_own_click_batches', 'helpers7_bias4', 'helpers3_click_batch2'])
@pytest.mark.parametrize('policy', ['uniform', 'random', 'ouc', 'ouc_wide', 'last_view'])
def test_walk_helpers_do_not_change_the_log(policy, form, monkeypatch):
    """k_walk2's helpers (round 5): the idle lanes of a bandit iteration evaluate the NEXT events of the bandit runs in it, and
    an owner skips what its helpers found.  Every form — none, one, seven per owner; the click batch inside a bandit iteration
    or in its own; other event-kind biases and batch sizes — must leave the oracle's rows, bit for bit: policies whose act
    reads the view history line (OrganicUserEventCounter at 60 and 3 000 products), the last view (frozen table), or nothing;
    organic-only users (their runs end the user), long histories (few products: lines fill up), phantom rows."""
    from oracle import oracle as orc
    env = {'helpers0': dict(RECOGYM_WALK_HELPERS='0'),
           'helpers1': dict(RECOGYM_WALK_HELPERS='1'),
           'helpers7_own_click_batches': dict(RECOGYM_WALK_HELPERS='7', RECOGYM_WALK_CLICK_JOIN='0'),
           'helpers7_bias4': dict(RECOGYM_WALK_HELPERS='7', RECOGYM_WALK_BIAS='4', RECOGYM_WALK_SEARCH_BATCH='8'),
           'helpers3_click_batch2': dict(RECOGYM_WALK_HELPERS='3', RECOGYM_WALK_CLICK_BATCH='2', RECOGYM_WALK_REFILL='1')}[form]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    P, K, n, n_org = {'ouc': 60, 'ouc_wide': 3000}.get(policy, 700), 20, 2600, 37
    cfg = Configuration({**env_1_args, 'random_seed': 5200 + len(form) + len(policy), 'num_products': P, 'K': K, 'sigma_omega': 0.0})
    pol = {'uniform': {},
           'random': dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=77),
           'ouc': dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=31, ouc=dict(gu.OUC_DEFAULTS)),
           'ouc_wide': dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=32, ouc=dict(gu.OUC_DEFAULTS)),
           'last_view': None}[policy]
    if pol is None:
        rs = np.random.RandomState(9)
        pol = dict(policy=_abi.RG_POLICY_LAST_VIEW_TABLE, policy_seed=0, policy_table=rs.randint(0, P, size=P).astype(np.int32),
                   policy_ps=rs.uniform(0.1, 1.0, size=P))
    want_env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    want = want_env.generate_logs(n, n_org)
    rows, cnt = run_sim(cfg, n, n_org, p_click=False, **pol)
    # (the frozen table's propensities are float32 on the device, as the BanditMF fixtures': compared at that resolution)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')}, ps_rtol=1e-6 if policy == 'last_view' else 1e-12,
                         what=f'helpers {policy} {form}')
    assert (rows['phantom'] == want['phantom']).all()
    oc = want_env.counters()
    assert (cnt['organic'], cnt['bandit'], cnt['clicks'], cnt['phantom']) == (oc['organic'], oc['bandit'], oc['clicks'], oc['phantom'])
    assert cnt['live'] == 0 and cnt['log_dropped'] == 0


def test_bench_multi_rank_line_reports_both_scaling_forms_and_the_allreduce(tmp_path):
    """`bench.py --gpus 2` end to end on THIS one-GPU box (both ranks on cuda:0, gloo instead of RCCL: the script's multi-rank
    logic, not a measurement): one JSON line with the weak form as `value`, the strong form beside it, the per-rank step times
    and the latency of the counter all-reduce; both forms count the events of the id ranges they simulate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RECOGYM_BENCH_BACKEND='gloo', RECOGYM_BENCH_ONE_DEVICE='1')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'tiny', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline', '--no-materialise'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['config']['users_total'] == 40_000
    o = line['other_scaling']
    assert o['scaling'] == 'strong' and o['users_total'] == 20_000 and o['users_per_gpu'] == 10_000
    assert len(line['per_rank_ms_per_step']) == 2 and len(o['per_rank_ms_per_step']) == 2
    assert line['allreduce_us'] > 0 and line['allreduce']['samples'] == 100
    # ~100 events per user in both forms
    assert 0.8 < (line['config']['events_per_step'] / 40_000) / (o['events_per_step'] / 20_000) < 1.25


@pytest.mark.parametrize('workload,users', [('c3', 2_000_000), ('c2', 1_000_000), ('c3drift', 1_000_000), ('c4shard', 250_000),
                                            ('c5', 250_000), ('c5trained', 1_000_000)])
def test_sampled_oracle_parity_at_bench_size(workload, users):
    """The oracle replays >= 2 000 user ids of a bench-size run of the default path — the longest-lived users, users of every
    fate of the user-major walk (round 1 only / float64 batch + round 2 / finished by the wave-per-user last round), a uniform
    spread — and their rows from the sorted device log are the oracle's bit for bit (ps, and p_click in a second run, to
    1e-12): tests/oracle_spot_check.py.  c5: both frozen arms at 10^4 classes; c5trained: both arms fitted by the reference."""
    import oracle_spot_check as osc
    # (the oracle scores 10^4 classes per bandit event of c5's LogReg arm — 500 s for 2 000 users on 64 threads: 150 users
    # here; profiles/r4/oracle_spot_check_full_size.jsonl holds the 2 000-user check of every workload at its full bench size)
    n_sample = 150 if workload == 'c5' else 2000
    for line in osc.spot_check(workload, users, n_sample=n_sample, p_click_modes=(False,) if workload == 'c5' else (False, True)):
        assert line['sampled_users'] >= n_sample and line['rows_compared'] > 100 * n_sample
        if workload in ('c3', 'c2'):
            k = line['kinds']
            assert k['float64_batch_and_round_2'] > 0 and k['finished_by_the_last_round'] > 0 and k['round_1_only'] > 0, k


@pytest.mark.parametrize('form', ['walk_pipe', 'walk_pipe_solo', 'run_walk', 'k_walk', 'rounds', 'lockstep', 'tail'])
def test_raw_log_rows_beyond_2_31_are_kept(form, monkeypatch):
    """Raw-log row arithmetic past 2^31 rows (a full-size log gets there at ~20 M users of BASELINE config 3; round 5 found the
    three walk kernels assembling a reserved row base from a SIGNED readfirstlane: every row of a chunk beyond 2^31 fell outside
    the capacity check and was dropped, and no check ever saw it).  The test hook rg_sim_debug_set_row_base starts the raw log of
    a few thousand users at row 2^31 - 4096, so that their rows straddle the line, for every kernel that reserves or addresses
    raw rows: k_walk2 + k_walk_solo (run_walk_pipe / run_walk), k_walk, k_advance_run (rounds), k_advance (lock-step), k_tail.
    Rows against the oracle, nothing dropped."""
    from oracle import oracle as orc
    from recogym_amd.sim import Simulator, default_log_capacity
    for k in ('RECOGYM_DRAW', 'RECOGYM_WALK', 'RECOGYM_WALK_HANDOVER', 'RECOGYM_PIPE_MIN', 'RECOGYM_RUN_AHEAD', 'RECOGYM_TAIL', 'RECOGYM_PIPE'):
        monkeypatch.delenv(k, raising=False)
    sigma = 0.0
    if form in ('walk_pipe', 'walk_pipe_solo'):
        monkeypatch.setenv('RECOGYM_PIPE_MIN', '256')
        if form == 'walk_pipe_solo':
            monkeypatch.setenv('RECOGYM_WALK_HANDOVER', '64')
    elif form == 'run_walk':
        monkeypatch.setenv('RECOGYM_PIPE', '0')
    elif form == 'k_walk':
        monkeypatch.setenv('RECOGYM_WALK', '1')
    else:
        sigma = 0.1
        monkeypatch.setenv('RECOGYM_RUN_AHEAD', '32' if form == 'rounds' else '0')
        monkeypatch.setenv('RECOGYM_TAIL', '100000' if form == 'tail' else '0')
    n = 3000
    cfg = Configuration({**env_1_args, 'random_seed': 231, 'num_products': 200, 'K': 20, 'sigma_omega': sigma})
    pol = dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=9)
    base = (1 << 31) - 4096
    cap = base + default_log_capacity(cfg, n)
    sim = Simulator(cfg, n, device='cuda:0', log_capacity=cap, p_click=False, **pol)
    _abi.check(sim.lib.rg_sim_debug_set_row_base(sim._h, base), 'debug_set_row_base')
    sim.reset_users(0, n)
    sim.run()
    cnt = sim.counters()
    rows = sim.rows()
    sim.close()
    torch.cuda.empty_cache()
    assert cnt['log_dropped'] == 0 and cnt['log_rows'] > (1 << 31), cnt
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol).generate_logs(n)
    gu.assert_rows_equal(rows, {k: want[k] for k in ('u', 't', 'z', 'v', 'a', 'c', 'ps')}, ps_rtol=1e-12, what=f'row base 2^31 - 4096, {form}')
    assert (rows['phantom'] == want['phantom']).all()


def test_external_actions_out_of_range_are_counted():
    """RG_POLICY_EXTERNAL (the gym path's actions arrive from the host): an action outside [0, P) must not index beta / mu_b — the
    batched rg_sim_step evaluates it with product 0 — and is COUNTED (RG_CNT_BAD_ACTION), so that a caller bug is visible instead of
    producing plausible rows (rg_sim_step_user rejects the same input with RG_EINVAL)."""
    from recogym_amd.sim import Simulator
    cfg = Configuration({**env_1_args, 'random_seed': 3, 'num_products': 12, 'K': 4, 'sigma_omega': 0.0})
    n = 4096
    sim = Simulator(cfg, n, device='cuda:0', policy=_abi.RG_POLICY_EXTERNAL)
    sim.reset_users(0, n)
    acts = torch.zeros(n, dtype=torch.int32, device='cuda:0')
    bad = 0
    for t in range(6):
        a = torch.randint(0, 12, (n,), dtype=torch.int32, device='cuda:0')
        a[::7] = -1
        a[3::11] = 12
        st = sim.states()                                   # who is at a bandit event now: only their action is read
        is_b = torch.zeros(n, dtype=torch.bool, device='cuda:0')
        is_b[:st.numel()] = st == 1
        bad += int((is_b & ((a < 0) | (a >= 12))).sum().item())
        sim.step(a)
    c = sim.counters()
    sim.close()
    assert bad > 0 and c['bad_actions'] == bad, (bad, c['bad_actions'])
