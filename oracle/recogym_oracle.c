/*
 * recogym_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C float64 restatement of the reference's reco-gym-v1 step loop, one user at a time,
 * one event at a time, exactly as /root/reference/recogym/envs/abstract.py and reco_env_v1.py
 * do it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker; nothing under recogym_amd/ imports or links it.
 *
 * Two draw sources:
 *   RGO_RNG_MT     — a sequential MT19937 stream with numpy RandomState's legacy sampling
 *                    (random_sample, legacy polar gauss with its cached second value, masked
 *                    rejection randint).  In this mode the oracle reproduces the UNMODIFIED
 *                    reference row for row; it is pinned against the reference's own golden
 *                    vectors (`Getting Started.ipynb` cells 7 and 9) and against logs generated
 *                    by importing the reference here (tests/golden/, made by
 *                    tests/make_golden.py).
 *   RGO_RNG_PHILOX — the counter RNG of include/recogym_rng.h.  In this mode it is the oracle
 *                    the HIP path must match bit-exactly on (t, u, z, v, a, c) and to 1e-12 (relative) on
 *                    ps / p_click; it is itself pinned against the reference's arithmetic by
 *                    running the unmodified reference with the same draws injected through
 *                    `env.rng` (tests/ref_harness.py -> tests/golden/philox_*.npz).
 *
 * Third-party arithmetic that is not under /root/reference: numpy.random.mtrand.RandomState
 * (README pins numpy==1.17.2; 2.2.6 installed; the legacy stream is frozen by numpy policy).
 * Its published algorithms are restated in the "numpy RandomState legacy" section below.
 *
 * Known, documented deviations from numpy at the 1e-16 level (they cannot change an index
 * unless a uniform lands within ~1e-15 of a CDF boundary): matrix-vector products are plain
 * k-ordered loops (numpy: OpenBLAS dgemv), `uprob.sum()` is a sequential sum (numpy: pairwise),
 * exp is libm's (numpy: its own SIMD exp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/recogym_hip.h"
#include "../include/recogym_rng.h"

#define RGO_RNG_PHILOX 0
#define RGO_RNG_MT 1

/* One log row in the reference's order; -1 stands for None/NA. */
typedef struct rgo_row {
    uint32_t u;
    uint32_t t;
    int32_t z;        /* 0 organic, 1 bandit */
    int32_t v;        /* product viewed, -1 on bandit rows */
    int32_t a;        /* action, -1 on organic rows */
    int32_t c;        /* click, -1 on organic rows */
    int32_t phantom;  /* 1 on the trailing undrawn bandit row (abstract.py:311-316) */
    int32_t pad;
    double ps;        /* NaN on organic rows */
    double p_click;   /* ff(beta[a].omega + mu_b[a]) on real bandit rows, NaN otherwise */
    double time;      /* the time generator's clock at the row (== t with DefaultTimeGenerator) */
} rgo_row;

/* ------------------------------------------------------------------------------------------
 * numpy RandomState legacy (numpy/random/src/mt19937/mt19937.c, legacy-distributions.c,
 * distributions.c:random_bounded_uint64 with use_masked).
 * ---------------------------------------------------------------------------------------- */
typedef struct rgo_mt {
    uint32_t key[624];
    int pos;
    int has_gauss;
    double gauss;
} rgo_mt;

static void mt_seed(rgo_mt* s, uint32_t seed) {
    /* init_genrand: RandomState(int) -> _legacy_seeding -> mt19937_seed */
    for (int i = 0; i < 624; ++i) {
        s->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)(i + 1);
    }
    s->pos = 624;
    s->has_gauss = 0;
    s->gauss = 0.0;
}

static void mt_gen(rgo_mt* s) {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;
    uint32_t y;
    int i;
    for (i = 0; i < 624 - 397; ++i) {
        y = (s->key[i] & UPPER) | (s->key[i + 1] & LOWER);
        s->key[i] = s->key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
    }
    for (; i < 623; ++i) {
        y = (s->key[i] & UPPER) | (s->key[i + 1] & LOWER);
        s->key[i] = s->key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
    }
    y = (s->key[623] & UPPER) | (s->key[0] & LOWER);
    s->key[623] = s->key[396] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
    s->pos = 0;
}

static uint32_t mt_next32(rgo_mt* s) {
    if (s->pos == 624) mt_gen(s);
    uint32_t y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static double mt_double(rgo_mt* s) {
    const uint32_t a = mt_next32(s) >> 5, b = mt_next32(s) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static double mt_gauss(rgo_mt* s) {
    if (s->has_gauss) {
        const double g = s->gauss;
        s->has_gauss = 0;
        s->gauss = 0.0;
        return g;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * mt_double(s) - 1.0;
        x2 = 2.0 * mt_double(s) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    s->gauss = f * x1;
    s->has_gauss = 1;
    return f * x2;
}

/* RandomState.randint(0, n) for n <= 2^32: masked rejection on 32-bit words. */
static uint32_t mt_randint(rgo_mt* s, uint32_t n) {
    const uint32_t rng = n - 1u;
    if (rng == 0u) return 0u;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = (mt_next32(s) & mask)) > rng) {}
    return v;
}

/* ------------------------------------------------------------------------------------------
 * The environment object: the fields of AbstractEnv / RecoEnv1 that the step loop touches.
 * ---------------------------------------------------------------------------------------- */
typedef struct rgo_env {
    rg_config cfg;
    int rng_mode;
    const double* gamma;      /* (P,K) row-major */
    const double* mu_o;       /* (P) */
    const double* beta;       /* (P,K) */
    const double* mu_b;       /* (P) */
    /* AbstractEnv state */
    int state;                /* self.state */
    int first_step;           /* self.first_step */
    uint32_t user;            /* self.current_user_id */
    uint32_t time;            /* event index of the current user == self.current_time with DefaultTimeGenerator */
    double clock;             /* self.current_time with NormalTimeGenerator (normal_time_generator.py:23-26) */
    double* omega;            /* (K) */
    double last_p_click;
    /* draw sources */
    rgo_mt env_mt;            /* self.rng in MT mode */
    rgo_mt pol_mt;            /* agent.rng / agent.model.rng in MT mode */
    /* policy state: ViewsFeaturesProvider.views (agents/abstract.py:379-395) */
    int32_t* views;           /* (P) */
    int32_t last_view;        /* BanditMFSquare.last_product_viewed (bandit_mf.py:60-65) */
    const int32_t* pol_table; /* RG_POLICY_LAST_VIEW_TABLE: action per last viewed product */
    const double* pol_ps;     /* and the ps logged with it (NULL = 1.0) */
    const double* lr_coef_t;  /* RG_POLICY_LOGREG_FROZEN: coef_ transposed [P][n_classes], */
    const double* lr_intercept; const int32_t* lr_classes; uint32_t lr_n;   /* intercept_, classes_ */
    /* reco-gym-v0 (cfg.env_kind = 1, recogym/envs/reco_env_v0.py): the cluster toy model */
    const double* e0_click_p; /* (P,P) click_probs[action][view] = f(P / 5 (T + T') + phi), reco_env_v0.py:39-44 */
    uint32_t e0_cluster;      /* products per cluster: int(P / num_clusters), reco_env_v0.py:33 */
    int32_t product_view;     /* self.product_view */
    /* scratch */
    double* buf;              /* (P) */
    double* zbuf;             /* (K) */
    double* cdfbuf;           /* (P) scratch of icdf_right */
    int64_t counters[4];      /* organic, bandit (real), clicks, phantom */
} rgo_env;

rgo_env* rgo_env_create(const rg_config* cfg, int rng_mode, const double* gamma,
                        const double* mu_o, const double* beta, const double* mu_b) {
    rgo_env* e = (rgo_env*)calloc(1, sizeof(rgo_env));
    if (!e) return NULL;
    e->cfg = *cfg;
    e->rng_mode = rng_mode;
    e->gamma = gamma; e->mu_o = mu_o; e->beta = beta; e->mu_b = mu_b;
    e->omega = (double*)calloc(cfg->K, sizeof(double));
    e->views = (int32_t*)calloc(cfg->num_products, sizeof(int32_t));
    e->buf = (double*)calloc(cfg->num_products, sizeof(double));
    e->zbuf = (double*)calloc(cfg->K, sizeof(double));
    e->cdfbuf = (double*)calloc(cfg->num_products, sizeof(double));   /* per-instance scratch: no malloc per draw */
    e->state = RG_STATE_ORGANIC;
    e->first_step = 1;
    /* init_gym: reset_random_seed (abstract.py:88); agents seed their own RandomState at
     * construction (random_agent.py:20) or first act (organic_user_count.py:42). */
    mt_seed(&e->env_mt, (uint32_t)cfg->seed);
    mt_seed(&e->pol_mt, (uint32_t)cfg->policy_seed);
    return e;
}

void rgo_env_destroy(rgo_env* e) {
    if (!e) return;
    free(e->omega); free(e->views); free(e->buf); free(e->zbuf); free(e->cdfbuf); free(e);
}

/* LogregMulticlassIpsAgent with a fitted model (agents/logreg_ips.py:60-87, select_randomly = False) */
void rgo_env_set_logreg(rgo_env* e, const double* coef_t, const double* intercept,
                        const int32_t* classes, uint32_t n_classes) {
    e->lr_coef_t = coef_t; e->lr_intercept = intercept; e->lr_classes = classes; e->lr_n = n_classes;
}

void rgo_env_set_policy_table(rgo_env* e, const int32_t* table, const double* ps) {
    e->pol_table = table;
    e->pol_ps = ps;
}

/* RecoEnv0.set_static_params results (reco_env_v0.py:22-47): the click matrix as numpy / scipy computed it, the cluster size */
void rgo_env_set_env0(rgo_env* e, const double* click_p, uint32_t cluster_size) {
    e->e0_click_p = click_p;
    e->e0_cluster = cluster_size;
}

/* AbstractEnv.reset_random_seed (abstract.py:59-62): seed is already random_seed + epoch. */
void rgo_env_reseed(rgo_env* e, uint64_t seed) {
    e->cfg.seed = seed;
    mt_seed(&e->env_mt, (uint32_t)seed);
}

void rgo_env_reseed_policy(rgo_env* e, uint64_t seed) {
    e->cfg.policy_seed = seed;
    mt_seed(&e->pol_mt, (uint32_t)seed);
}

/* K standard normals for (user, t) — the `Z(K)` of SURVEY.md Appendix A. */
static void draw_normals(rgo_env* e, uint32_t purpose, uint32_t t, double* z) {
    const uint32_t K = e->cfg.K;
    if (e->rng_mode == RGO_RNG_MT) {
        for (uint32_t k = 0; k < K; ++k) z[k] = mt_gauss(&e->env_mt);
        return;
    }
    for (uint32_t j = 0; 2 * j < K; ++j) {
        const rg_u32x4 w = rg_draw(e->cfg.seed, e->user, t, j, purpose);
        const double u1 = rg_uniform(w.w[0], w.w[1]);
        const double u2 = rg_uniform(w.w[2], w.w[3]);
        const double r = sqrt(-2.0 * log(1.0 - u1));
        const double th = RG_TWO_PI * u2;
        z[2 * j] = r * cos(th);
        if (2 * j + 1 < K) z[2 * j + 1] = r * sin(th);
    }
}

/* uniform for the event draw (which = 0) or the transition draw (which = 1) at time t */
static double draw_event_uniform(rgo_env* e, uint32_t t, int which) {
    if (e->rng_mode == RGO_RNG_MT) return mt_double(&e->env_mt);
    const rg_u32x4 w = rg_draw(e->cfg.seed, e->user, t, 0, RG_DRAW_EVENT);
    return which == 0 ? rg_uniform(w.w[0], w.w[1]) : rg_uniform(w.w[2], w.w[3]);
}

static uint32_t icdf_right(const double* p, uint32_t n, double u, double* cdf);

/* RecoEnv1.reset + AbstractEnv.reset (reco_env_v1.py:78-82, abstract.py:90-103) */
void rgo_env_reset(rgo_env* e, uint32_t user_id) {
    e->first_step = 1;
    e->state = RG_STATE_ORGANIC;
    e->time = 0;                       /* time_generator.reset(); new_time() -> 0 */
    e->clock = 0.0;
    e->user = user_id;
    memset(e->views, 0, sizeof(int32_t) * e->cfg.num_products);   /* agent.reset() */
    if (e->cfg.env_kind == 1) {
        /* RecoEnv0.reset (reco_env_v0.py:49-55): product_view = rng.choice(P, p = ones(P) / P) */
        const uint32_t P = e->cfg.num_products;
        for (uint32_t p = 0; p < P; ++p) e->buf[p] = 1.0 / (double)P;
        double u;
        if (e->rng_mode == RGO_RNG_MT) u = mt_double(&e->env_mt);
        else { const rg_u32x4 w = rg_draw(e->cfg.seed, e->user, 0, 0, RG_DRAW_RESET); u = rg_uniform(w.w[0], w.w[1]); }
        e->product_view = (int32_t)icdf_right(e->buf, P, u, e->cdfbuf);
        return;
    }
    double* z = e->zbuf;
    draw_normals(e, RG_DRAW_RESET, 0, z);
    for (uint32_t k = 0; k < e->cfg.K; ++k)
        e->omega[k] = 0.0 + e->cfg.sigma_omega_initial * z[k];
}

/* RandomState.choice(n, p) tail: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right') */
static uint32_t icdf_right(const double* p, uint32_t n, double u, double* cdf) {
    double acc = 0.0;
    for (uint32_t i = 0; i < n; ++i) { acc += p[i]; cdf[i] = acc; }
    const double last = cdf[n - 1];
    uint32_t lo = 0, hi = n;           /* first i with cdf[i]/last > u */
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (cdf[mid] / last <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* RecoEnv1.update_product_view (reco_env_v1.py:119-128) */
static int32_t update_product_view(rgo_env* e) {
    const uint32_t P = e->cfg.num_products, K = e->cfg.K;
    double* l = e->buf;
    if (e->cfg.env_kind == 1) {
        /* RecoEnv0.update_product_view (reco_env_v0.py:65-67): choice(P, p = product_transition[view, :]); the row of the
         * block-diagonal matrix (:33-37): 1 / cluster_size inside the view's cluster (T / its row sums), 0 elsewhere */
        const uint32_t cr = e->e0_cluster, c0 = ((uint32_t)e->product_view / cr) * cr;
        for (uint32_t p = 0; p < P; ++p) l[p] = (p >= c0 && p < c0 + cr) ? 1.0 / (double)cr : 0.0;
        const double u0 = draw_event_uniform(e, e->time, 0);
        e->product_view = (int32_t)icdf_right(l, P, u0, e->cdfbuf);
        return e->product_view;
    }
    double mx = -INFINITY;
    for (uint32_t p = 0; p < P; ++p) {
        double d = 0.0;
        const double* g = e->gamma + (size_t)p * K;
        for (uint32_t k = 0; k < K; ++k) d += g[k] * e->omega[k];
        l[p] = d + e->mu_o[p];
        if (l[p] > mx) mx = l[p];
    }
    double s = 0.0;
    for (uint32_t p = 0; p < P; ++p) { l[p] = exp(l[p] - mx); s += l[p]; }
    for (uint32_t p = 0; p < P; ++p) l[p] = l[p] / s;
    const double u = draw_event_uniform(e, e->time, 0);
    const uint32_t v = icdf_right(l, P, u, e->cdfbuf);
    return (int32_t)v;
}

/* RecoEnv1.update_state (reco_env_v1.py:85-100); DefaultTimeGenerator => omega_k == 1 */
static void update_state(rgo_env* e) {
    const double u = draw_event_uniform(e, e->time, 1);
    const double* cdf = e->cfg.trans_cdf[e->state];   /* state is organic or bandit here */
    int ns = 0;
    while (ns < 2 && cdf[ns] <= u) ++ns;
    const uint32_t t_event = e->time;
    e->state = ns;
    e->time += 1;
    if (e->cfg.env_kind == 1) return;   /* RecoEnv0.update_state (reco_env_v0.py:56-58): the Markov draw only — no clock, no omega
                                         * (`time` stays the event index the draws are addressed by; the reference's current_time
                                         * never moves: rows carry clock 0) */
    /* time_delta = new_time() - old_time; omega_k = 1 if time_delta == 0 else time_delta (reco_env_v1.py:89-92).
     * DefaultTimeGenerator: always 1.  NormalTimeGenerator (Philox mode only): the increment that follows event t is
     * |mu + sigma z|, z = the RG_DRAW_TIME draw of (user, t) */
    double omega_k = 1.0;
    if (e->cfg.time_mode == 1) {
        const rg_u32x4 w = rg_draw(e->cfg.seed, e->user, t_event, 0, RG_DRAW_TIME);
        const double z0 = sqrt(-2.0 * log(1.0 - rg_uniform(w.w[0], w.w[1]))) * cos(RG_TWO_PI * rg_uniform(w.w[2], w.w[3]));
        const double dt = fabs(e->cfg.time_mu + e->cfg.time_sigma * z0);
        e->clock = e->clock + dt;
        omega_k = dt == 0.0 ? 1.0 : dt;
    } else e->clock = (double)e->time;
    if (e->cfg.change_omega_for_bandits || e->state == RG_STATE_ORGANIC) {
        double* z = e->zbuf;
        /* the draws are consumed even when sigma_omega == 0 (MT mode must advance) */
        draw_normals(e, RG_DRAW_DRIFT, t_event, z);
        for (uint32_t k = 0; k < e->cfg.K; ++k)
            e->omega[k] = e->omega[k] + (e->cfg.sigma_omega * omega_k) * z[k];
    }
}

static double sig(double x) { return 1.0 / (1.0 + exp(-x)); }
/* ff, reco_env_v1.py:38-41 */
static double ff(double x) { return sig(5.0 * sig(2.0 * sig(0.3 * x) - 2.0) - 6.0); }

/* RecoEnv1.draw_click (reco_env_v1.py:104-116).  The cached_state_seed is semantically
 * transparent: ctr[a] == ff(beta[a].omega + mu_b[a]) for the current omega. */
/* numpy legacy binomial(n = 1, p) (numpy/random/src/legacy/legacy-distributions.c: legacy_random_binomial_original ->
 * random_binomial_inversion; p > 0.5 draws n - inversion(1 - p)) */
static int32_t binomial1(rgo_env* e, double p) {
    const double pe = p <= 0.5 ? p : 1.0 - p;
    const int64_t n = 1;
    const double q = 1.0 - pe;
    const double qn = exp((double)n * log(q));
    const double np = (double)n * pe;
    const double b = np + 10.0 * sqrt(np * q + 1.0);
    const int64_t bound = (int64_t)((double)n < b ? (double)n : b);
    int64_t X = 0;
    double px = qn;
    uint32_t slot = 0;
    double U = draw_event_uniform(e, e->time, 0);
    while (U > px) {
        X++;
        if (X > bound) {
            X = 0;
            px = qn;
            slot += 1;
            if (e->rng_mode == RGO_RNG_MT) U = mt_double(&e->env_mt);
            else { const rg_u32x4 w = rg_draw(e->cfg.seed, e->user, e->time, slot, RG_DRAW_EVENT); U = rg_uniform(w.w[0], w.w[1]); }
        } else {
            U -= px;
            px = ((double)(n - X + 1) * pe * px) / ((double)X * q);
        }
    }
    return (int32_t)(p <= 0.5 ? X : n - X);
}

static int32_t draw_click(rgo_env* e, int32_t a) {
    const uint32_t K = e->cfg.K;
    if (e->cfg.env_kind == 1) {
        /* RecoEnv0.draw_click (reco_env_v0.py:61-63): rng.binomial(1, click_probs[recommendation, product_view]) */
        const double p = e->e0_click_p[(size_t)a * e->cfg.num_products + (uint32_t)e->product_view];
        e->last_p_click = p;
        return binomial1(e, p);
    }
    double d = 0.0;
    const double* b = e->beta + (size_t)a * K;
    for (uint32_t k = 0; k < K; ++k) d += b[k] * e->omega[k];
    const double ctr = ff(d + e->mu_b[a]);
    e->last_p_click = ctr;
    /* choice([0,1], p=[1-ctr, ctr]) */
    const double p0 = 1.0 - ctr;
    const double c1 = p0 + ctr;
    const double u = draw_event_uniform(e, e->time, 0);
    return (p0 / c1 <= u) ? 1 : 0;
}

typedef struct rgo_session {
    rgo_row* rows;
    uint64_t n, cap;
    int overflow;
} rgo_session;

static void push_row(rgo_session* s, const rgo_row* r) {
    if (s->n < s->cap) s->rows[s->n] = *r; else s->overflow = 1;
    s->n++;
}

/* AbstractEnv.generate_organic_sessions (abstract.py:105-121): rows go straight to `out`;
 * the policy observes them at its next act (ViewsFeaturesProvider.observe). */
static void generate_organic_sessions(rgo_env* e, rgo_session* out) {
    while (e->state == RG_STATE_ORGANIC) {
        const int32_t v = update_product_view(e);
        rgo_row r;
        memset(&r, 0, sizeof(r));
        r.u = e->user; r.t = e->time; r.time = e->clock; r.z = 0; r.v = v; r.a = -1; r.c = -1;
        r.ps = NAN; r.p_click = NAN;
        push_row(out, &r);
        e->views[v] += 1;
        e->last_view = v;
        e->counters[0] += 1;
        update_state(e);
    }
}

/* AbstractEnv.step (abstract.py:123-197).  action < 0 means None.  Returns reward (-1 = None);
 * organic rows of the returned observation are appended to `out`. */
int rgo_env_step(rgo_env* e, int32_t action, rgo_row* rows, uint64_t cap, uint64_t* n_rows,
                 int32_t* done) {
    rgo_session s = {rows, 0, cap, 0};
    int32_t reward = -1;
    if (e->first_step) {
        if (action >= 0) return -100;              /* assert (action_id is None) */
        e->first_step = 0;
        generate_organic_sessions(e, &s);
    } else {
        if (action < 0) return -101;               /* assert (action_id is not None) */
        reward = draw_click(e, action);
        update_state(e);
        if (reward == 1) e->state = RG_STATE_ORGANIC;   /* abstract.py:180-181 */
        if (e->state == RG_STATE_ORGANIC) generate_organic_sessions(e, &s);
    }
    *n_rows = s.n;
    *done = (e->state == RG_STATE_STOP);
    return reward;
}

uint32_t rgo_env_time(const rgo_env* e) { return e->time; }
double rgo_env_clock(const rgo_env* e) { return e->clock; }
int rgo_env_state(const rgo_env* e) { return e->state; }
void rgo_env_omega(const rgo_env* e, double* out) {
    memcpy(out, e->omega, sizeof(double) * e->cfg.K);
}
double rgo_env_last_p_click(const rgo_env* e) { return e->last_p_click; }

/* The policy's act: returns the action, writes ps.  Covers agent=None (abstract.py:209-221),
 * RandomAgent.act (random_agent.py:22-33) and OrganicUserEventCounterModel.act
 * (organic_user_count.py:45-96) on ViewsFeaturesProvider counts (agents/abstract.py:347-358). */
int32_t rgo_env_policy_act(rgo_env* e, double* ps_out) {
    const uint32_t P = e->cfg.num_products;
    const uint32_t t = e->time;
    rg_u32x4 w = {{0, 0, 0, 0}};
    if (e->rng_mode == RGO_RNG_PHILOX)
        w = rg_draw(e->cfg.policy_seed, e->user, t, 0, RG_DRAW_POLICY);

    if (e->cfg.policy == RG_POLICY_UNIFORM_ENV || e->cfg.policy == RG_POLICY_RANDOM_AGENT) {
        uint32_t a;
        if (e->rng_mode == RGO_RNG_MT)
            a = mt_randint(e->cfg.policy == RG_POLICY_UNIFORM_ENV ? &e->env_mt : &e->pol_mt, P);
        else
            a = rg_bounded(w.w[0], w.w[1], P);
        *ps_out = 1.0 / (double)P;
        return (int32_t)a;
    }

    if (e->cfg.policy == RG_POLICY_LAST_VIEW_TABLE) {
        /* BanditMFSquare.act with frozen embeddings: argmax_a <E_p[a], E_u[lpv]> is a table */
        *ps_out = e->pol_ps ? e->pol_ps[e->last_view] : 1.0;
        return e->pol_table[e->last_view];
    }

    if (e->cfg.policy == RG_POLICY_LOGREG_FROZEN) {
        /* logreg.predict(features): decision_function = X @ coef_.T + intercept_ (sklearn
         * linear_model/_base.py), X = the 1 x P CSR row of view counts (agents/abstract.py:316-409).
         * scipy's csr_matvecs (sparsetools/csr.h) adds count * coef_t[p][:] for the stored (ascending)
         * products with a separate multiply and add; the file is built with -ffp-contract=off, so the
         * plain expression below is that arithmetic.  argmax = first maximum (numpy). */
        uint32_t best = 0;
        double best_s = 0.0;
        for (uint32_t c = 0; c < e->lr_n; ++c) {
            double sc = 0.0;
            for (uint32_t p = 0; p < P; ++p)
                if (e->views[p]) sc = sc + (double)e->views[p] * e->lr_coef_t[(size_t)p * e->lr_n + c];
            sc = sc + e->lr_intercept[c];
            if (e->cfg.lr_select_randomly && c < P) e->buf[c] = sc;
            if (c == 0 || sc > best_s) { best = c; best_s = sc; }
        }
        if (e->cfg.lr_select_randomly) {
            /* select_randomly (logreg_ips.py:61-72): action_proba = predict_proba(features) = softmax(decision_function)
             * (sklearn.utils.extmath.softmax: exp(x - max) / sum), action = rng.choice(num_products, p = action_proba) with the
             * model's own RandomState (the second policy uniform of the event here), ps = action_proba[action].
             * Needs every product as a class (lr_n == P), like the reference's choice over num_products. */
            double sum = 0.0;
            for (uint32_t c = 0; c < P; ++c) { e->buf[c] = exp(e->buf[c] - best_s); sum += e->buf[c]; }
            for (uint32_t c = 0; c < P; ++c) e->buf[c] = e->buf[c] / sum;
            double u;
            if (e->rng_mode == RGO_RNG_MT) u = mt_double(&e->pol_mt);
            else { const rg_u32x4 w = rg_draw(e->cfg.policy_seed, e->user, e->time, 0, RG_DRAW_POLICY); u = rg_uniform(w.w[2], w.w[3]); }
            const uint32_t a = icdf_right(e->buf, P, u, e->cdfbuf);
            *ps_out = e->buf[a < P ? a : P - 1];
            return (int32_t)a;
        }
        *ps_out = 1.0;
        return e->lr_classes[best];
    }

    /* OrganicUserEventCounterModel.act */
    const double eps = e->cfg.ouc_epsilon;
    double* f = e->buf;
    int is_explore = 0;
    if (e->cfg.ouc_exploit_explore) {
        /* choice([True, False], p=[eps, 1 - eps]) -> index 0 is True */
        const double u0 = (e->rng_mode == RGO_RNG_MT) ? mt_double(&e->pol_mt)
                                                     : rg_uniform(w.w[0], w.w[1]);
        const double c0 = eps, c1 = eps + (1.0 - eps);
        is_explore = !(c0 / c1 <= u0);
        double sum = 0.0;
        if (is_explore) {
            for (uint32_t p = 0; p < P; ++p) { f[p] = (e->views[p] == 0) ? 1.0 : 0.0; sum += f[p]; }
        } else {
            for (uint32_t p = 0; p < P; ++p) { f[p] = (double)e->views[p]; sum += f[p]; }
        }
        for (uint32_t p = 0; p < P; ++p) f[p] = f[p] / sum;
    } else {
        double sum = 0.0;
        for (uint32_t p = 0; p < P; ++p) { f[p] = eps + (double)e->views[p]; sum += f[p]; }
        for (uint32_t p = 0; p < P; ++p) f[p] = f[p] / sum;
        if (e->cfg.ouc_reverse_pop) {
            double s2 = 0.0;
            for (uint32_t p = 0; p < P; ++p) { f[p] = 1.0 - f[p]; s2 += f[p]; }
            for (uint32_t p = 0; p < P; ++p) f[p] = f[p] / s2;
        }
    }
    if (e->cfg.ouc_select_randomly) {
        const double u1 = (e->rng_mode == RGO_RNG_MT) ? mt_double(&e->pol_mt)
                                                     : rg_uniform(w.w[2], w.w[3]);
        const uint32_t a = icdf_right(f, P, u1, e->cdfbuf);
        if (e->cfg.ouc_exploit_explore)
            *ps_out = (is_explore ? eps : 1.0 - eps) * f[a];
        else
            *ps_out = f[a];
        return (int32_t)a;
    }
    uint32_t best = 0;
    for (uint32_t p = 1; p < P; ++p) if (f[p] > f[best]) best = p;
    *ps_out = 1.0;
    return (int32_t)best;
}

/* AbstractEnv.generate_logs (abstract.py:241-327).  User ids start at first_user_id (the
 * reference always starts at 0).  Returns the number of rows the log has (which may exceed
 * `cap`, in which case only the first cap rows were stored), or a negative error. */
int64_t rgo_env_generate_logs(rgo_env* e, uint64_t first_user_id, uint64_t n_users,
                              uint64_t n_organic_users, rgo_row* rows, uint64_t cap) {
    rgo_session s = {rows, 0, cap, 0};
    uint64_t uid = first_user_id;
    memset(e->counters, 0, sizeof(e->counters));
    for (uint64_t i = 0; i < n_organic_users; ++i) {      /* abstract.py:293-297 */
        rgo_env_reset(e, (uint32_t)uid++);
        e->first_step = 0;
        generate_organic_sessions(e, &s);
    }
    for (uint64_t i = 0; i < n_users; ++i) {               /* abstract.py:299-316 */
        rgo_env_reset(e, (uint32_t)uid++);
        e->first_step = 0;
        generate_organic_sessions(e, &s);                  /* step(None) */
        int done = (e->state == RG_STATE_STOP);
        while (!done) {
            /* step_offline: act, then step(action) */
            rgo_row r;
            memset(&r, 0, sizeof(r));
            r.u = e->user; r.t = e->time; r.time = e->clock; r.z = 1; r.v = -1;
            r.a = rgo_env_policy_act(e, &r.ps);
            r.c = draw_click(e, r.a);
            r.p_click = e->last_p_click;
            update_state(e);
            if (r.c == 1) e->state = RG_STATE_ORGANIC;
            /* the reference stores the organic rows of the NEW observation after the bandit
             * row (they are stored at the top of the next loop pass / after the loop) */
            push_row(&s, &r);
            e->counters[1] += 1;
            e->counters[2] += r.c;
            if (e->state == RG_STATE_ORGANIC) generate_organic_sessions(e, &s);
            done = (e->state == RG_STATE_STOP);
        }
        /* final step_offline(done=True): one more act, reward forced to 0 (abstract.py:223-233) */
        rgo_row r;
        memset(&r, 0, sizeof(r));
        r.u = e->user; r.t = e->time; r.time = e->clock; r.z = 1; r.v = -1; r.c = 0; r.phantom = 1;
        r.a = rgo_env_policy_act(e, &r.ps);
        r.p_click = NAN;
        push_row(&s, &r);
        e->counters[3] += 1;
    }
    return (int64_t)s.n;
}

void rgo_env_counters(const rgo_env* e, int64_t* out) {
    for (int i = 0; i < 4; ++i) out[i] = e->counters[i];
}

/* exposed for tests: the raw draw maps of include/recogym_rng.h */
void rgo_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                uint32_t* out) {
    const rg_u32x4 w = rg_philox4x32_10(c0, c1, c2, c3, k0, k1);
    for (int i = 0; i < 4; ++i) out[i] = w.w[i];
}
double rgo_uniform(uint32_t a, uint32_t b) { return rg_uniform(a, b); }
uint32_t rgo_bounded(uint32_t a, uint32_t b, uint32_t n) { return rg_bounded(a, b, n); }
double rgo_ff(double x) { return ff(x); }

/* exposed for tests: the MT legacy sampling, to be compared with numpy.random.RandomState */
void rgo_mt_probe(uint32_t seed, uint32_t n_randint, int n, double* doubles, double* gauss,
                  uint32_t* ints) {
    rgo_mt s;
    mt_seed(&s, seed);
    for (int i = 0; i < n; ++i) doubles[i] = mt_double(&s);
    for (int i = 0; i < n; ++i) gauss[i] = mt_gauss(&s);
    for (int i = 0; i < n; ++i) ints[i] = mt_randint(&s, n_randint);
}
