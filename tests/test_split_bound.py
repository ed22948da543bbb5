"""The error budget the certificate grants the two-way fp16 operand split (DESIGN.md §2,
`f16_extra_delta` in recogym_amd/csrc/rg_common.hpp / rg_draw_pipelined.hip), checked numerically on the host: the device forms
l = sum_k (h1(g) h1(w) + h2(g) h1(w) + h1(g) h2(w)) with exact products and fp32 accumulation; here
the same three terms are summed exactly (float64 holds fp16 x fp16 products exactly), so what is
measured is the REPRESENTATION error the split adds on top of the accumulation budget."""
import numpy as np
import pytest


def split2(x):
    x = np.asarray(x, dtype=np.float32)
    h1 = x.astype(np.float16)
    h2 = (x - h1.astype(np.float32)).astype(np.float16)
    return h1.astype(np.float64), h2.astype(np.float64)


@pytest.mark.parametrize('scale_g, scale_w', [(1.0, 1.0), (3.0, 0.2), (0.05, 4.0), (1e-3, 1e-3), (30.0, 10.0)])
def test_fp16_split_error_is_inside_its_budget(scale_g, scale_w):
    rng = np.random.RandomState(7)
    K, n = 20, 20000
    log2e = 1.4426950408889634
    g = (rng.randn(n, K) * scale_g * log2e).astype(np.float32)     # fl32(Gamma log2 e) rows
    w = (rng.randn(n, K) * scale_w).astype(np.float32)             # fl32(omega)
    g1, g2 = split2(g)
    w1, w2 = split2(w)
    kept = (g1 * w1 + g2 * w1 + g1 * w2).sum(axis=1)               # what the MFMA sums (exactly)
    true = (g.astype(np.float64) * w.astype(np.float64)).sum(axis=1)
    err = np.abs(kept - true)                                      # log2 units
    terms = np.abs(g.astype(np.float64) * w.astype(np.float64)).sum(axis=1)
    budget = 3 * 2.0 ** -22 * terms + 2.0 ** -25 * (np.abs(g).sum(axis=1) + np.abs(w).sum(axis=1))
    assert (err <= budget).all(), float((err / budget).max())
    # and it is not a vacuous bound: the worst case uses a fair share of it
    assert (err / budget).max() > 0.02


def test_fp16_pieces_reconstruct_fp32():
    rng = np.random.RandomState(3)
    x = (rng.randn(100000) * np.exp(rng.uniform(-12, 6, 100000))).astype(np.float32)
    h1, h2 = split2(x)
    e = np.abs(x.astype(np.float64) - h1 - h2)
    assert (e <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -25)).all()


@pytest.mark.parametrize('P, K, sigma_mu, scale', [(10000, 20, 3.0, 1.0), (2000, 20, 3.0, 2.5), (3000, 64, 3.0, 1.0), (500, 5, 30.0, 0.3)])
def test_joint_logit_bound_dominates_every_partial_sum(P, K, sigma_mu, scale):
    """`ahat_of` (rg_common.hpp): the bound the certificate's accumulation budget (K + 5) 2^-24 Ahat is proportional to.
    Restated here: max_p (|mu_p| + ||Gamma_p||_2 r) read off a grid of r = (i + 1) / 4 at the grid point at or above
    ||omega||_2 (and the two older bounds).  It must dominate |mu_p + sum_{k <= j} Gamma_pk omega_k| for every product p and
    every partial sum j, and it should be visibly tighter than taking the two maxima separately."""
    rng = np.random.RandomState(P + K)
    gamma = rng.randn(P, K)
    mu = rng.randn(P) * sigma_mu
    norm = np.sqrt((gamma ** 2).sum(axis=1))
    mumax, g2max = np.abs(mu).max(), norm.max()
    gmax_k = np.abs(gamma).max(axis=0)
    grid = np.array([(np.abs(mu) + norm * ((i + 1) * 0.25)).max() for i in range(64)])
    gains = []
    for _ in range(200):
        om = rng.randn(K) * scale
        r = float(np.sqrt((om ** 2).sum())) * 1.000001
        joint = mumax + g2max * r
        gi = max(int(np.ceil(r * 4.0)), 1)
        if gi <= 64:
            joint = min(joint, grid[gi - 1])
        ahat = min(mumax + float((np.abs(om) * gmax_k).sum()), joint)
        partial = np.abs(mu[:, None] + np.cumsum(gamma * om[None, :], axis=1))
        assert partial.max() <= ahat * (1 + 1e-12), (partial.max(), ahat)
        gains.append(ahat / (mumax + min(float((np.abs(om) * gmax_k).sum()), g2max * r)))
    assert np.mean(gains) <= 1.0 and min(gains) < 0.97         # never looser than before, usually tighter


def test_no_click_threshold_is_below_every_click_boundary():
    """kNoClickBelow (rg_common.hpp): a bandit event whose uniform is below it cannot click, whatever beta[a].omega is.
    The reference's draw (reco_env_v1.py:38-41, 104-116): ctr = ff(x) = sig(5 sig(2 sig(0.3 x) - 2) - 6) and
    choice([0, 1], p=[1 - ctr, ctr]) = [u >= cdf[0]], cdf = cumsum(p) / cumsum(p)[-1].  Checked here: the constant in the
    source, the range of ff over the whole float64 line, and numpy's own boundary for the largest ctr."""
    import os, re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'recogym_amd', 'csrc', 'rg_common.hpp')).read()
    thr = float(re.search(r'constexpr double kNoClickBelow = ([0-9.]+);', src).group(1))
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    ff = lambda x: sig(5.0 * sig(2.0 * sig(0.3 * x) - 2.0) - 6.0)
    with np.errstate(over='ignore'):
        x = np.concatenate([np.linspace(-200, 200, 400001), [-1e308, -1e30, 1e30, 1e308, -np.inf, np.inf]])
        ctr = ff(x)
    assert ctr.max() <= sig(-3.5) * (1 + 1e-15) and ctr.min() >= 0.0044
    p = np.stack([1.0 - ctr, ctr], axis=1)
    cdf = np.cumsum(p, axis=1)
    cdf /= cdf[:, -1:]
    assert thr < cdf[:, 0].min() - 5e-4          # 0.97 against 0.970688: three orders above any rounding of the chain
    # and the threshold is worth having: it spares 97 % of the events
    assert thr >= 0.97


def _cert_correlated(S, A, a, b, delta, delta_c=None):
    """Host restatement of cert_correlated (rg_common.hpp): -> (num_lo, den_lo, num_hi, den_hi, valid).  delta_c: the budget of
    the recomputed in-chunk prefixes a, b where it differs from the sweep terms' (the walk behind k_sweep_xh: fp32 recompute)."""
    dp = delta * (1.0 + 2.0 * delta)
    dc = dp if delta_c is None else delta_c * (1.0 + 2.0 * delta_c)
    rho = 2.0 ** -20 * 1.001 * S
    T = S - A
    return (A * (1.0 + dp) + a * (1.0 + dc) + rho, T * (1.0 - dp) + A * (1.0 + dp),
            A * (1.0 - dp) + b * (1.0 - dc) - rho, T * (1.0 + dp) + A * (1.0 - dp), T >= 0.0 and delta < 0.25)


@pytest.mark.parametrize('delta, delta_c', [(1e-6, None), (8.6e-5, None), (1e-3, None), (1.0e-5, 7.4e-5), (1e-6, 1e-3)])
def test_correlated_certificate_is_sound_for_worst_case_term_errors(delta, delta_c):
    """The certificate on correlated errors: the prefix A at the start of the draw's chunk and the total S are
    sums of the SAME sweep terms e_p (1 + eps_p), the prefixes a, b inside the chunk are recomputed terms with errors of their
    own, all |eps| <= delta, and the stored A, S carry up to 2^-21 of fp32 rounding each.  For random softmax-like term
    vectors and the sign patterns that move a boundary furthest (prefix up / suffix down and the reverse, chunk terms against
    them, storage roundings against the decision), a certified product must be the true one — and the certified band must be
    narrower than the independent-error form's, which is the point of it."""
    rng = np.random.RandomState(5)
    P, n_cases = 2048, 300
    wins = 0
    for case in range(n_cases):
        e = np.exp(rng.standard_normal(P) * rng.choice([1.0, 3.0, 4.5]))          # true terms
        c_true = np.cumsum(e)
        S_true = c_true[-1]
        v = int(np.searchsorted(c_true, rng.random_sample() * S_true, 'right'))   # the product a draw would land in
        v = min(v, P - 1)
        c0 = (v // 32) * 32                                                       # its chunk starts here
        for sgn in (+1.0, -1.0):                                                  # which way the sweep's errors lean
            for sgn_in in (+1.0, -1.0):                                           # ... and the recomputed chunk's
                for sgn_r in (+1.0, -1.0):                                        # ... and the storage roundings
                    sweep = e.copy()
                    sweep[:c0] *= 1.0 + sgn * delta                               # terms before the chunk
                    sweep[c0:] *= 1.0 - sgn * delta                               # the chunk and everything behind it
                    A = sweep[:c0].sum() * (1.0 + sgn_r * 2.0 ** -21)
                    S = sweep.sum() * (1.0 - sgn_r * 2.0 ** -21)
                    rec = e[c0:v + 1] * (1.0 + sgn_in * (delta if delta_c is None else delta_c))
                    a, b = rec[:-1].sum(), rec.sum()
                    num_lo, den_lo, num_hi, den_hi, valid = _cert_correlated(S, A, a, b, delta, delta_c)
                    if not valid or den_lo <= 0 or den_hi <= 0:
                        continue
                    u_lo, u_hi = num_lo / den_lo, num_hi / den_hi                 # certified iff u_lo < u < u_hi
                    t_lo = (c_true[v - 1] if v else 0.0) / S_true                 # the true interval of product v
                    t_hi = c_true[v] / S_true
                    if u_lo < u_hi:
                        assert (v == 0 or u_lo >= t_lo) and (v == P - 1 or u_hi <= t_hi), (case, v, sgn, sgn_in, sgn_r)
                        # the independent form's interval: C~[v-1](1+d) / (S~(1-d)) < u < C~[v](1-d) / (S~(1+d))
                        dmax = delta if delta_c is None else max(delta, delta_c)
                        o_lo = (A + a) * (1 + dmax) / (S * (1 - dmax))
                        o_hi = (A + b) * (1 - dmax) / (S * (1 + dmax))
                        wins += (u_hi - u_lo) > max(o_hi - o_lo, 0.0)
    assert wins > n_cases            # (8 sign patterns per case: the correlated interval is the wider one in most of them)


# ---------------------------------------------------------------------------------------------------------------------------
# k_sweep_xh (recogym_amd/csrc/rg_draw_exacthi.hip): the fixed-point pieces, restated in numpy.  Checked here: every product of
# the leading group is a multiple of 2^-16 and the group's sum is EXACT in fp32 in any order; the residual group carries what
# is left of Gamma', omega and mu to ~2^-26; the error of a term against float64 stays inside xh_delta's budget.
# ---------------------------------------------------------------------------------------------------------------------------
def _f16(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float16).astype(np.float64)


def _xh_split_omega(w):
    hi = _f16(np.rint(w * 256.0) / 256.0)
    r1 = w - hi
    mid9 = _f16(r1 * 512.0)
    mid6 = _f16(mid9 * 0.125)
    lo15 = _f16((r1 - mid9 / 512.0) * 32768.0)
    return hi, mid9, lo15, mid6, r1


def _xh_table(gamma, mu):
    log2e = 1.4426950408889634074
    g = gamma * log2e
    ghi = _f16(np.rint(g * 256.0) / 256.0)
    rg = g - ghi
    glo9 = _f16(rg * 512.0)
    glo3 = _f16(rg * 8.0)
    ghi6 = _f16(ghi * 0.015625)
    m = mu * log2e
    m1 = _f16(np.rint(m * 32.0) / 32.0)
    r = m - m1
    m2s = _f16(np.rint(r * 65536.0) / 64.0)
    seed = ((r - m2s / 1024.0) * 512.0).astype(np.float32).astype(np.float64)
    return dict(g=g, ghi=ghi, ghi6=ghi6, glo9=glo9, glo3=glo3, m=m, m1=m1, m2s=m2s, seed=seed,
                err_col=np.abs(rg - glo9 / 512.0).max(axis=0), glomax=np.abs(glo9 / 512.0).max(),
                seedmax=np.abs(r - m2s / 1024.0).max())


@pytest.mark.parametrize('P, K, sigma_mu, scale', [(4000, 20, 3.0, 1.0), (1500, 20, 3.0, 2.0), (800, 7, 3.0, 1.0), (600, 13, 10.0, 0.3)])
def test_xh_pieces_are_exact_in_fp32_and_inside_the_budget(P, K, sigma_mu, scale):
    rng = np.random.RandomState(100 + P + K)
    gamma = rng.randn(P, K)
    mu = rng.randn(P) * sigma_mu
    T = _xh_table(gamma, mu)
    e24, ln2, log2e = 2.0 ** -24, np.log(2.0), 1.4426950408889634074
    NL = 5 if K > 8 else 2
    for trial in range(12):
        om = rng.randn(K) * scale
        hi, mid9, lo15, mid6, r1 = _xh_split_omega(om)
        ltrue = (T['g'] * om).sum(axis=1) + T['m']                     # float64, log2 units
        q = np.ceil(ltrue[:32].max())
        # ---- the leading group: fixed point, exact in fp32 in any order ----
        terms = np.concatenate([T['ghi'] * hi, T['m1'][:, None], (T['m2s'] * 2.0 ** -10)[:, None], np.full((P, 1), -q)], axis=1)
        assert (np.rint(terms * 65536.0) == terms * 65536.0).all()     # multiples of 2^-16
        assert np.abs(terms).sum(axis=1).max() < 255.0                  # every subset sum below 2^24 quanta
        for order in range(3):
            perm = rng.permutation(terms.shape[1])
            acc = np.zeros(P, dtype=np.float32)
            for c in perm:
                acc = (acc + terms[:, c].astype(np.float32)).astype(np.float32)
            assert (acc.astype(np.float64) == terms.sum(axis=1)).all()
        H = terms.sum(axis=1)
        # ---- the residual group, accumulated in fp32 one term at a time from its seed (scaled by 2^9) ----
        lo_terms = np.concatenate([T['ghi'] * mid9, T['ghi6'] * lo15, T['glo9'] * hi, T['glo3'] * mid6], axis=1)
        acc = T['seed'].astype(np.float32)
        for c in rng.permutation(lo_terms.shape[1]):
            acc = (acc + lo_terms[:, c].astype(np.float32)).astype(np.float32)
        x32 = (np.float32(2.0 ** -9) * acc + H.astype(np.float32)).astype(np.float32)     # the join: one rounding
        # what xh_delta (rg_common.hpp) grants this user
        absw = np.abs(om).sum()
        egam = (np.abs(om) * T['err_col']).sum()
        gmax = np.abs(gamma).max(axis=0)
        lob = (np.abs(r1) * (gmax * log2e + 2.0 ** -8)).sum() + ((np.abs(om) + np.abs(r1)) * T['glomax']).sum() + T['seedmax']
        norm = np.sqrt((gamma ** 2).sum(axis=1))
        ahat = (np.abs(mu) + norm * np.sqrt((om ** 2).sum())).max()
        assert ahat * log2e * 1.001 + abs(q) + 2.0 < 255.0             # xh_eligible
        e_lo = (16.0 * NL + 4.0) * e24 * (lob + 1.6e-5)
        e_x = e24 * (ahat * log2e + abs(q)) * 1.01
        e_drop = K * 4.0e-9
        err_log2 = np.abs(x32.astype(np.float64) - (ltrue - q))
        assert (err_log2 <= egam + e_drop + e_lo + e_x).all(), float((err_log2 / (egam + e_drop + e_lo + e_x)).max())
        # ... and the budget is not vacuous
        assert (err_log2 / (egam + e_drop + e_lo + e_x)).max() > 0.02
        delta = ln2 * (egam + e_drop + e_lo + e_x) + 1.2e-6
        if K == 20 and scale == 1.0 and sigma_mu == 3.0:
            assert delta < 1.6e-5                                       # (the two-way split's budget at this shape: ~1e-4)
