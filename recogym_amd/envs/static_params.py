"""Host-side construction of the read-only tables and the POD config of the step loop.

These run once per `init_gym` on the host with numpy's legacy MT19937 `RandomState`, so that
Gamma / mu_organic / beta / mu_bandit are bit-identical to what the reference draws for the same
`random_seed` (reference: RecoEnv1.set_static_params, recogym/envs/reco_env_v1.py:51-75, and
generate_beta / normalize_beta, reco_env_v1.py:130-174).  O(P*K) one-off work; the per-event
hot path never touches this module.
"""
import numpy as np
from numpy.random.mtrand import RandomState

from .. import _abi


def transition_matrix(config):
    """3x3 Markov matrix over (organic, bandit, stop) — reco_env_v1.py:54-61.

    Note `prob_leave_organic` fills the stop column of BOTH rows and `prob_leave_bandit` is
    never read; reproduced on purpose.
    """
    p_ob = config.prob_organic_to_bandit
    p_bo = config.prob_bandit_to_organic
    p_stop = config.prob_leave_organic
    T = np.array([[0.0, p_ob, p_stop], [p_bo, 0.0, p_stop], [0.0, 0.0, 1.0]])
    # the reference subtracts python's left-to-right sum of the row from 1
    T[0, 0] = 1 - (T[0, 0] + T[0, 1] + T[0, 2])
    T[1, 1] = 1 - (T[1, 0] + T[1, 1] + T[1, 2])
    return T


def transition_cdf(T):
    """What RandomState.choice(3, p=T[s]) compares its uniform with: cumsum(p) / cumsum(p)[-1]."""
    out = np.empty((2, 3))
    for s in (0, 1):
        cdf = T[s].cumsum()
        cdf /= cdf[-1]
        out[s] = cdf
    return out


def flip_index(Gamma, number_of_flips):
    """Pairing used by generate_beta (reco_env_v1.py:147-168): walk product pairs from the most
    to the least correlated (Gamma Gamma^T, diagonal zeroed) and swap each pair whose two
    members are both still unpaired, until `number_of_flips` swaps were made."""
    P = Gamma.shape[0]
    cov = Gamma @ Gamma.T
    cov = cov - np.diag(np.diag(cov))
    index = np.arange(P)
    taken = np.zeros(P, dtype=bool)
    done = 0
    for flat in cov.flatten().argsort()[::-1]:
        i, j = int(flat / P), int(np.mod(flat, P))
        if taken[i] or taken[j]:
            continue
        index[i], index[j] = j, i
        taken[i] = taken[j] = True
        done += 1
        if done == number_of_flips:
            break
    return index


def draw_tables(config):
    """Gamma (P,K), mu_organic (P,), beta (P,K), mu_bandit (P,) as float64 C-contiguous arrays,
    drawn in the reference's order from RandomState(random_seed)."""
    rng = RandomState(config.random_seed)
    P, K = config.num_products, config.K
    Gamma = rng.normal(size=(P, K))
    mu_organic = rng.normal(0, config.sigma_mu_organic, size=(P, 1))
    flips = getattr(config, 'number_of_flips', 0)
    if flips == 0:
        beta, mu_bandit = Gamma, mu_organic
    else:
        idx = flip_index(Gamma, flips)
        beta, mu_bandit = Gamma[idx, :], mu_organic[idx, :]
    if getattr(config, 'normalize_beta', False):
        beta = beta / np.sqrt((beta ** 2).sum(1)[:, np.newaxis])
    c = np.ascontiguousarray
    return c(Gamma), c(mu_organic.ravel()), c(beta), c(mu_bandit.ravel())


def make_rg_config(config, seed, policy=_abi.RG_POLICY_UNIFORM_ENV, policy_seed=None,
                   ouc=None):
    """Fill struct rg_config from an env Configuration (+ optional agent parameters)."""
    cfg = _abi.RgConfig()
    cfg.num_products = int(config.num_products)
    cfg.K = int(config.K)
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.policy_seed = (int(seed) if policy_seed is None else int(policy_seed)) & 0xFFFFFFFFFFFFFFFF
    cdf = transition_cdf(transition_matrix(config))
    for s in (0, 1):
        for j in range(3):
            cfg.trans_cdf[s][j] = float(cdf[s, j])
    cfg.sigma_omega_initial = float(config.sigma_omega_initial)
    cfg.sigma_omega = float(config.sigma_omega)
    cfg.change_omega_for_bandits = int(bool(getattr(config, 'change_omega_for_bandits', False)))
    cfg.policy = int(policy)
    ouc = ouc or {}
    cfg.ouc_select_randomly = int(bool(ouc.get('select_randomly', True)))
    cfg.ouc_exploit_explore = int(bool(ouc.get('exploit_explore', True)))
    cfg.ouc_reverse_pop = int(bool(ouc.get('reverse_pop', False)))
    cfg.ouc_history_cap = int(ouc.get('history_cap', 0))
    cfg.ouc_epsilon = float(ouc.get('epsilon', 0.0))
    mode, mu, sigma = time_generator_params(config)
    cfg.time_mode, cfg.time_mu, cfg.time_sigma = mode, mu, sigma
    return cfg


def time_generator_params(config):
    """-> (time_mode, mu, sigma) of `config.time_generator` (abstract.py:72-76): 0 for the default generator (or none),
    1 for a NormalTimeGenerator — this package's or the reference's own class, recognised by name."""
    tg = getattr(config, 'time_generator', None)
    if tg is None or type(tg).__name__ == 'DefaultTimeGenerator':
        return 0, 0.0, 1.0
    if type(tg).__name__ == 'NormalTimeGenerator':
        return (1, float(getattr(tg, 'normal_time_mu', getattr(config, 'normal_time_mu', 0))),
                float(getattr(tg, 'normal_time_sigma', getattr(config, 'normal_time_sigma', 1))))
    raise NotImplementedError(f'time generator {type(tg).__name__} is not supported by the device step loop')
