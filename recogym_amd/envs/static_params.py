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


def flip_index_blocked(Gamma, number_of_flips, block_rows=None, keep=None):
    """The same pairing without the P x P matrix (80 GB at P = 10^5): Gamma Gamma^T is symmetric, so the walk only ever needs
    the pairs i < j in descending order of their value, and it stops after `number_of_flips` swaps — long before it leaves the
    top few thousand pairs.  Rounds: (1) over row blocks of Gamma Gamma^T (upper triangle), the `keep`-th largest value below
    the previous round's threshold; (2) every pair at or above it, sorted descending, walked exactly like flip_index walks the
    full argsort.  A pair (i, j) and its mirror (j, i) are the same swap, whichever the reference's unstable argsort meets
    first; exact float64 ties between DIFFERENT pairs have no defined order in the reference either (numpy's introsort) and
    probability ~1e-10 among the few thousand pairs walked.  Memory: block_rows x P doubles."""
    P, K = Gamma.shape
    if block_rows is None:
        block_rows = max(1, min(P, (256 << 20) // (8 * P)))
    if keep is None:
        keep = max(4096, 64 * int(number_of_flips))
    index = np.arange(P)
    taken = np.zeros(P, dtype=bool)
    done = 0
    upper = np.inf                      # pairs with a value >= upper were walked in an earlier round
    cols = np.arange(P)
    while done < number_of_flips:
        # round, pass 1: the keep-th largest remaining value
        vals, thr = np.empty(0), -np.inf          # the `keep` largest so far and the smallest of them (a running filter)
        for lo in range(0, P, block_rows):
            blk = Gamma[lo:lo + block_rows] @ Gamma.T
            m = (cols[None, :] > np.arange(lo, lo + blk.shape[0])[:, None]) & (blk < upper) & (blk > thr)
            vals = np.concatenate([vals, blk[m]])
            if vals.size > keep:
                vals = np.partition(vals, vals.size - keep)[vals.size - keep:]
                thr = vals.min()
        if vals.size == 0:
            break
        tau = vals.min()
        # pass 2: every remaining pair at or above it
        ii, jj, vv = [], [], []
        for lo in range(0, P, block_rows):
            blk = Gamma[lo:lo + block_rows] @ Gamma.T
            m = (cols[None, :] > np.arange(lo, lo + blk.shape[0])[:, None]) & (blk < upper) & (blk >= tau)
            r, c = np.nonzero(m)
            ii.append(r + lo); jj.append(c); vv.append(blk[r, c])
        ii, jj, vv = np.concatenate(ii), np.concatenate(jj), np.concatenate(vv)
        for k in np.argsort(-vv, kind='stable'):
            i, j = int(ii[k]), int(jj[k])
            if taken[i] or taken[j]:
                continue
            index[i], index[j] = j, i
            taken[i] = taken[j] = True
            done += 1
            if done == number_of_flips:
                break
        if tau <= 0.0:
            # the walk would now enter the zeroed diagonal and the negative half: hand the (tiny) rest to the full algorithm's
            # order — only reachable when number_of_flips approaches P / 2 on a small table
            if done < number_of_flips:
                raise ValueError('flip_index_blocked: number_of_flips too large for the blocked walk; use flip_index')
        upper = tau
    return index


# tables above this many products build the pairing in row blocks (the P x P float64 matrix of the reference's algorithm: 0.8 GB
# at 10^4 products, 80 GB at 10^5)
FLIP_BLOCKED_ABOVE = 8192


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
        idx = flip_index_blocked(Gamma, flips) if P > FLIP_BLOCKED_ABOVE else flip_index(Gamma, flips)
        beta, mu_bandit = Gamma[idx, :], mu_organic[idx, :]
    if getattr(config, 'normalize_beta', False):
        beta = beta / np.sqrt((beta ** 2).sum(1)[:, np.newaxis])
    c = np.ascontiguousarray
    return c(Gamma), c(mu_organic.ravel()), c(beta), c(mu_bandit.ravel())


def make_rg_config(config, seed, policy=_abi.RG_POLICY_UNIFORM_ENV, policy_seed=None,
                   ouc=None, env_kind=0, lr_select_randomly=False):
    """Fill struct rg_config from an env Configuration (+ optional agent parameters).  env_kind = 1: reco-gym-v0 (no K, no
    omega: the fields of the latent-factor model are left at neutral values)."""
    cfg = _abi.RgConfig()
    cfg.num_products = int(config.num_products)
    cfg.env_kind = int(env_kind)
    if env_kind:
        cfg.K = 1
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.policy_seed = (int(seed) if policy_seed is None else int(policy_seed)) & 0xFFFFFFFFFFFFFFFF
        cdf = transition_cdf(transition_matrix(config))
        for s in (0, 1):
            for j in range(3):
                cfg.trans_cdf[s][j] = float(cdf[s, j])
        cfg.policy = int(policy)
        ouc = ouc or {}
        cfg.ouc_select_randomly = int(bool(ouc.get('select_randomly', True)))
        cfg.ouc_exploit_explore = int(bool(ouc.get('exploit_explore', True)))
        cfg.ouc_reverse_pop = int(bool(ouc.get('reverse_pop', False)))
        cfg.ouc_history_cap = int(ouc.get('history_cap', 0))
        cfg.ouc_epsilon = float(ouc.get('epsilon', 0.0))
        cfg.time_sigma = 1.0
        return cfg
    cfg.K = int(config.K)
    cfg.lr_select_randomly = int(bool(lr_select_randomly))
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


def draw_env0_tables(config):
    """RecoEnv0.set_static_params (recogym/envs/reco_env_v0.py:22-47), with the same numpy / scipy calls on the same
    RandomState(random_seed) stream -> dict(phi, product_transition, click_probs, cluster_size, cdf_init, cdf_cluster):
    the last two are what `RandomState.choice(P, p=...)` compares its uniform against at reset (:52-54) and at every organic
    event (:65-67; every cluster's row of the block-diagonal matrix has the same values inside its cluster)."""
    from numpy import eye, kron, ones, sqrt
    from scipy.special import expit
    P = int(config.num_products)
    rng = RandomState(config.random_seed)
    cluster_ratio = int(P / config.num_clusters)
    ones_mat = ones((cluster_ratio, cluster_ratio))
    T = kron(eye(config.num_clusters), ones_mat)
    T = T / kron(T.sum(1), ones((P, 1))).T
    phi = rng.normal(scale=sqrt(config.phi_var), size=(P, P))
    click_probs = expit(P / 5. * (T + T.T) + phi - 5)          # abstract.py:34-37: f(mat, offset=5) = sigmoid(mat - offset)
    init = ones(P) / P
    cdf_init = init.cumsum()
    cdf_init /= cdf_init[-1]
    row = T[0, :cluster_ratio].copy()
    cdf_cluster = row.cumsum()
    cdf_cluster /= cdf_cluster[-1]
    return dict(phi=phi, product_transition=T, click_probs=np.ascontiguousarray(click_probs), cluster_size=cluster_ratio,
                cdf_init=cdf_init, cdf_cluster=cdf_cluster)
