/*
 * recogym_hip.h — C ABI of librecogym_hip.so, the MI355X-native reco-gym-v1 step loop.
 *
 * The reference has no FFI: its boundary is the duck-typed Python class surface of
 * recogym/envs/abstract.py + reco_env_v1.py.  This header is the boundary a
 * native replacement of that path exports; every entry point cites the reference method(s)
 * whose work it takes over.  The Python mirror of the reference classes that binds these
 * symbols with ctypes lives in recogym_amd/ (see INTEGRATION.md for the stub a reference
 * maintainer would add).
 *
 * Conventions
 *   - plain C types only; device buffers are passed as raw pointers owned by the caller
 *     (PyTorch-ROCm tensors in the Python host), streams as `void*` (a hipStream_t);
 *   - every function returns 0 on success or a negative RG_E* code; rg_last_error() holds the
 *     message of the last failure on the calling thread;
 *   - nothing here allocates caller-visible memory: the caller sizes one workspace with
 *     rg_sim_workspace_bytes() and hands it to rg_sim_create();
 *   - no entry point synchronises the device except rg_sim_read_counters(), rg_sim_run() (a
 *     host loop that polls the live-user count) and rg_sim_destroy();
 *   - a handle is thread-compatible (one handle per thread / per GPU), not thread-safe;
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     RG_ENODEV.  The float64 CPU restatement used by the tests is a different library
 *     (oracle/, test infrastructure only).
 */
#ifndef RECOGYM_HIP_H
#define RECOGYM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RG_ABI_VERSION 6

/* error codes */
#define RG_OK 0
#define RG_EINVAL (-1)   /* bad argument / configuration */
#define RG_ENODEV (-2)   /* no HIP device, or a HIP runtime error */
#define RG_ENOMEM (-3)   /* workspace / log buffer too small */
#define RG_ESTATE (-4)   /* call sequence error (e.g. step before reset_users) */
#define RG_ELIMIT (-5)   /* a hard limit was hit (max steps, log overflow) */

/* Markov states — recogym/envs/abstract.py:41-43 */
#define RG_STATE_ORGANIC 0
#define RG_STATE_BANDIT 1
#define RG_STATE_STOP 2

/* policies that run on the device — the `agent` argument of generate_logs
 * (abstract.py:241-254) */
#define RG_POLICY_UNIFORM_ENV 0   /* agent=None: uniform action from the ENV stream, abstract.py:209-221 */
#define RG_POLICY_RANDOM_AGENT 1  /* RandomAgent, agents/random_agent.py:22-33 */
#define RG_POLICY_ORGANIC_USER_COUNT 2 /* OrganicUserEventCounterAgent, agents/organic_user_count.py:45-96 */
#define RG_POLICY_EXTERNAL 3      /* actions supplied by the caller per step (gym.Env.step, abstract.py:123) */
#define RG_POLICY_LAST_VIEW_TABLE 4 /* frozen policy a = table[last product viewed]: BanditMFSquare inference,
                                     agents/bandit_mf.py:52-87 (argmax_a <E_p[a], E_u[lpv]> is a P-entry table) */
#define RG_POLICY_LOGREG_FROZEN 5 /* frozen LogregMulticlassIpsAgent (select_randomly = False), agents/logreg_ips.py:60-87:
                                     a = classes[argmax_c (sum_p views[p] W[c][p] + b[c])] over the user's view counts
                                     (ViewsFeaturesProvider, agents/abstract.py:316-409), ps = 1 */

/*
 * Everything the step loop needs from `env.config` (a Configuration built from env_1_args,
 * reco_env_v1.py:18-29 + abstract.py:20-31) and from the agent's config.  POD, no pointers.
 */
typedef struct rg_config {
    uint32_t num_products;          /* P   — config.num_products */
    uint32_t K;                     /* K   — config.K */
    uint64_t seed;                  /* config.random_seed + epoch — abstract.py:59-62 */
    uint64_t policy_seed;           /* the agent's config.random_seed (env seed for UNIFORM_ENV) */
    /* Normalised cumulative transition rows, i.e. what RandomState.choice(3, p=T[s]) compares
     * its uniform against: cdf = cumsum(T[s]); cdf /= cdf[-1]   (reco_env_v1.py:54-61,87).
     * Row 0 = organic, row 1 = bandit.  Computed on the host in float64 exactly as numpy does. */
    double trans_cdf[2][3];
    double sigma_omega_initial;     /* reco_env_v1.py:80 */
    double sigma_omega;             /* reco_env_v1.py:96 */
    uint32_t change_omega_for_bandits; /* reco_env_v1.py:95 */
    uint32_t policy;                /* RG_POLICY_* */
    /* OrganicUserEventCounter parameters (organic_user_count_args, organic_user_count.py:7-27) */
    uint32_t ouc_select_randomly;
    uint32_t ouc_exploit_explore;
    uint32_t ouc_reverse_pop;
    uint32_t ouc_history_cap;       /* max organic views kept per user on the device (0 = default) */
    double ouc_epsilon;
    /* time generator (envs/features/time/): 0 = DefaultTimeGenerator (t = event index, default_time_generator.py:10-13);
     * 1 = NormalTimeGenerator (normal_time_generator.py:23-26): event n of a user happens at T_n = sum_{i<n} |mu + sigma z_i|,
     * and the drift that follows it is scaled by T_{n+1} - T_n (reco_env_v1.py:89-98).  Lock-step execution only. */
    uint32_t time_mode;
    uint32_t env_kind;              /* 0 = reco-gym-v1 (the latent-factor model); 1 = reco-gym-v0 (reco_env_v0.py: the cluster toy model,
                                     * every draw a table look-up — rg_sim_set_env0_tables instead of rg_sim_set_tables; K is ignored,
                                     * pass 1).  Lock-step execution only. */
    double time_mu;                 /* config.normal_time_mu (default 0) */
    double time_sigma;              /* config.normal_time_sigma (default 1) */
    /* RG_POLICY_LOGREG_FROZEN with select_randomly = True (logreg_ips.py:61-72): the action is SAMPLED from predict_proba —
     * softmax of the decision function, rng.choice(num_products, p = proba), ps = proba[action] — with the second policy
     * uniform of the event (words 2,3).  Needs every product as a class (classes = 0 .. P-1) and P <= 1024; lock-step only. */
    uint32_t lr_select_randomly;
    uint32_t reserved1;
} rg_config;

/*
 * One emitted log row, 16 bytes — the device-side form of one row of the DataFrame that
 * generate_logs builds (abstract.py:256-290,318-327).
 *   code bit 31 : z  (0 = organic, 1 = bandit)
 *   code bit 30 : c  (click; bandit rows only)
 *   code bit 29 : phantom (the trailing never-drawn bandit row, abstract.py:311-316)
 *   code bits 0..28 : v (organic) or a (bandit)
 */
typedef struct rg_event {
    uint32_t u;      /* user id */
    uint32_t t;      /* per-user event index == DefaultTimeGenerator time */
    uint32_t code;
    float ps;        /* propensity of the logged action (NaN on organic rows) */
} rg_event;

#define RG_EV_BANDIT 0x80000000u
#define RG_EV_CLICK 0x40000000u
#define RG_EV_PHANTOM 0x20000000u
#define RG_EV_INDEX_MASK 0x1FFFFFFFu

/* counters returned by rg_sim_read_counters */
#define RG_CNT_ORGANIC 0        /* organic rows */
#define RG_CNT_BANDIT 1         /* real bandit rows (phantom excluded) */
#define RG_CNT_CLICKS 2         /* sum of c over bandit rows — bench_agents.py:203-206 */
#define RG_CNT_PHANTOM 3        /* phantom rows (one per finished non-organic-only user) */
#define RG_CNT_LIVE 4           /* users not yet in state stop */
#define RG_CNT_STEP 5           /* Markov transitions performed per user so far (== current t) */
#define RG_CNT_LOG_ROWS 6       /* rows written to the log buffer */
#define RG_CNT_LOG_DROPPED 7    /* rows that did not fit (capacity exceeded) */
#define RG_CNT_EXACT_DRAWS 8    /* organic draws resolved by the float64 path */
#define RG_CNT_HIST_OVERFLOW 9  /* OrganicUserEventCounter views that did not fit ouc_history_cap */
#define RG_CNT_EXACT_SWEEPS 10  /* float64 product sweeps those draws needed (== EXACT_DRAWS unless sigma_omega = 0,
                                   where a user's float64 sums are taken once and reused) */
#define RG_CNT_EXACT_OVERFLOW 11 /* uncertified draws beyond what the float64 resolve scratch covers in a step (25 % of the
                                   live users): the run is incomplete and must be reported — never reached by the
                                   reference's parameter ranges */
#define RG_CNT_LR_ACTS 12       /* RG_POLICY_LOGREG_FROZEN: acts computed (one per change of a user's view history that an event needed) */
#define RG_CNT_LR_ROWS 13       /* ... and the coef^T rows (viewed products) those acts read */
#define RG_CNT_LR_EXACT 14      /* ... acts the fp32 scores could not certify (decided by float64 scores) */
#define RG_CNT_MEMO_HITS 15     /* sigma_omega = 0, user-major walk: organic draws answered by the user's memo of certified draws */
#define RG_CNT_ANCHORED 24      /* sigma_omega = 0 walk: the part of RG_CNT_EXACT_DRAWS that the float64-ANCHORED certificate resolved
                                   (the user's float64 prefix at the start of the draw's 64-product chunk + fp32 inside it) instead
                                   of a float64 walk of the chunk's products */
#define RG_CNT_BAD_ACTION 25   /* RG_POLICY_EXTERNAL: events whose action was outside [0, num_products): rg_sim_step evaluates them with
                                 * product 0 (it must not index beta / mu_b with them) and LOGS a = 0 — a caller bug made visible here
                                 * (rg_sim_step_user rejects the same input with RG_EINVAL before it reaches the device) */
#define RG_CNT_N 32             /* out[] of rg_sim_read_counters; slots 16..23 are internal */

typedef struct rg_sim rg_sim;

const char* rg_last_error(void);
int rg_abi_version(void);

/* number of visible HIP devices (0 on a CPU-only box; never an error) */
int rg_device_count(void);

/* Bytes of device workspace a simulator over `n_users` concurrent users needs
 * (omega, state lists, per-step counters, fp32 table copies, policy history). 0 on bad config. */
size_t rg_sim_workspace_bytes(const rg_config* cfg, uint64_t n_users);

/* AbstractEnv.init_gym (abstract.py:64-88) minus the table draws: binds a configuration and a
 * caller-owned device workspace to a handle.  Host-only; performs no device work. */
int rg_sim_create(rg_sim** out, const rg_config* cfg, uint64_t n_users, void* d_workspace,
                  size_t workspace_bytes);
int rg_sim_destroy(rg_sim* sim);

/* Run-path tuning knobs by name (none changes a result or the workspace layout; the defaults are the measured optima, DESIGN.md
 * §4 / §9).  rg_sim_create takes their initial values from the RECOGYM_* environment variables of the same meaning (the A/B
 * tests' way in); after that the library never reads the environment on the run path.  Names: walk_bias, walk_refill,
 * walk_handover, walk_click_batch, walk_search_batch, walk_helpers (0 .. 7), walk_click_join, walk_line64, pipe_groups, pipe_mode, pipe_occ1, pipe_occ2, pipe_xblocks,
 * pipe_min_users, exact_mix, exact_tile, resident_grid, slices (-1 = by population), sweep_prefix_off, tail_below,
 * repack_every, run_ahead (events a round of a run to the end may take a user through, 0 = an event per launch), lr_part_cap (acts
 * of a step the frozen-LogReg fp16 screen takes; can only be lowered), sweep_lds (1: the unsliced sweep of a run whose draws are
 * not cached keeps its tile prefixes in LDS and searches them there, k_draw_tp; 0: k_draw_bf16p's scratch + search), debug.
 * Read-only: sweep_lds_kernel (1 where k_draw_tp serves the configuration).
 * RG_EINVAL for an unknown name or a value out of range. */
int rg_sim_set_option(rg_sim* sim, const char* name, int64_t value);
int rg_sim_get_option(rg_sim* sim, const char* name, int64_t* value);

/* env_kind = 1 — RecoEnv0.set_static_params (reco_env_v0.py:22-47), as the tables its draws compare against, float64 device
 * arrays computed on the host exactly as numpy / the C library do (the device only compares):
 *   cdf_init    [P]               cumsum(ones(P) / P) / last          — reset: choice(P, p = initial_product_probs), :52-54
 *   cdf_cluster [cluster_size]    cumsum of a row of the block-diagonal product_transition inside its cluster, / last
 *                                 (every cluster's row has the same values) — update_product_view: choice(P, p = T[view]), :65-67
 *   click_p     [P][P]            click_probs[action][view] = f(P / 5 (T + T') + phi), :39-44 (exported as p_click)
 *   click_qn    [P][P]            exp(log(1 - p)): the threshold legacy binomial(1, p) compares its uniform with (numpy
 *                                 random_binomial_inversion; 1 - p where p > 0.5: the draw is then 1 - inversion(1 - p))
 *   click_px1   [P][P]            (p qn) / q of that algorithm's second step (a restart of the inversion, ~1e-16 of the draws)
 * cluster_size = P / num_clusters.  The library keeps the pointers. */
/* Host helper (no device, no handle): from click_probs p[n] the two thresholds of numpy's legacy binomial(1, p) —
 * qn[i] = exp(log(1 - p')) and px1[i] = (p' qn) / (1 - p'), p' = p or 1 - p where p > 0.5 — with the C library's exp / log, the ones
 * numpy's legacy-distributions.c calls: the device compares against exactly these doubles. */
int rg_env0_click_thresholds(const double* p, uint64_t n, double* qn, double* px1);
int rg_sim_set_env0_tables(rg_sim* sim, const double* d_cdf_init, const double* d_cdf_cluster, uint32_t cluster_size,
                           const double* d_click_p, const double* d_click_qn, const double* d_click_px1, void* stream);

/* RecoEnv1.set_static_params / generate_beta results (reco_env_v1.py:51-75,133-174): row-major
 * float64 device arrays Gamma (P,K), mu_organic (P), beta (P,K), mu_bandit (P), drawn on the
 * host from RandomState(seed) so they are bit-identical to the reference's.  The library keeps
 * the pointers (caller keeps them alive) and builds its fp32 tile copies in the workspace. */
int rg_sim_set_tables(rg_sim* sim, const double* d_gamma, const double* d_mu_organic,
                      const double* d_beta, const double* d_mu_bandit, void* stream);

/* RG_POLICY_LAST_VIEW_TABLE: per-product device tables, kept by pointer (caller keeps them alive):
 * d_action[p] = action taken when the user's last organic view was p, d_ps[p] = the `ps` value
 * logged with it (NULL = 1.0; BanditMFSquare logs its logit there, bandit_mf.py:84). */
int rg_sim_set_policy_table(rg_sim* sim, const int32_t* d_action, const float* d_ps);

/* RG_POLICY_LOGREG_FROZEN: the fitted model's arrays on the device, kept by pointer: d_coef_t =
 * sklearn's coef_ TRANSPOSED, row-major [num_products][n_classes] float64; d_intercept [n_classes];
 * d_classes [n_classes] = classes_ (the action of every class).  A two-class sklearn model (coef_ of
 * one row) is passed as two classes with a zero first row/intercept.  Scores are accumulated exactly
 * as scipy's CSR x dense product does (viewed products ascending, multiply then add, intercept
 * last), so the argmax is sklearn's predict() bit for bit. */
int rg_sim_set_logreg(rg_sim* sim, const double* d_coef_t, const double* d_intercept,
                      const int32_t* d_classes, uint32_t n_classes);

/* Optional fast path of RG_POLICY_LOGREG_FROZEN (after rg_sim_set_logreg): fp32 copies of coef^T [num_products][n_classes]
 * and intercept [n_classes] (each value rounded to nearest), d_wmax[p] >= max_c |coef_t[p][c]| and bmax >= max_c |intercept[c]|.
 * The policy's act is computed when a user's view history has changed (not once per event); with these arrays the class
 * scores are first taken in fp32 and accepted when the best one leads by more than twice the rounding bound
 * (views + 3) 2^-24 (bmax + sum_p views_p wmax[p]); everything else goes through the float64 walk of rg_sim_set_logreg's
 * arrays, so the logged action is sklearn's predict() bit for bit either way.  All NULL / 0 = float64 only. */
int rg_sim_set_logreg_fp32(rg_sim* sim, const float* d_coef32_t, const float* d_intercept32, const float* d_wmax, float bmax);

/* Optional screening pass on top of rg_sim_set_logreg_fp32 (its intercept32 / wmax / bmax are used): d_coef16_t = coef^T
 * [num_products][n_classes] rounded to nearest IEEE half (every |value| <= 65504; n_classes % 8 == 0).  An act then takes
 * all class scores from the half table (half the bytes of the fp32 one: at 10^4 classes the act is bound by streaming the
 * rows of the viewed products), keeps the classes whose score is within twice the rounding bound
 * sum_p views_p (2^-11 wmax[p] + 2^-25) + (views + 3) 2^-24 (bmax + sum_p views_p wmax[p]) of the best one, and decides among
 * them by float64 scores in scipy's order (rg_sim_set_logreg's arrays): sklearn's predict() bit for bit.  NULL = off. */
int rg_sim_set_logreg_fp16(rg_sim* sim, const uint16_t* d_coef16_t);

/* Optional: the screening pass from an 8-bit copy of coef^T instead of the fp16 one (after rg_sim_set_logreg_fp16; ABI v6): the rows
 * the pass streams are a QUARTER of the float32 bytes.  d_coef8_t [num_products][n_classes] unsigned bytes q + 128 with
 * q = rint(coef^T[p][c] / d_scale8[p]) in [-127, 127], d_scale8[p] = wmax[p] / 127 (fp32, rounded up): a weight is off by at most
 * d_scale8[p] / 2, so the pass's bound is sum_p views_p wmax[p] / 254 (+ the fp32 accumulation terms) where the fp16 pass has
 * sum_p views_p wmax[p] 2^-11 — more classes survive it; they are scored once more from the fp16 rows (that pass's bound), float64
 * scores still decide among what is left.  The pass reads 20 bytes per lane and row: d_coef8_t must be readable 16 bytes past its
 * last row, and n_classes % 4 == 0.  NULL = the fp16 pass. */
int rg_sim_set_logreg_int8(rg_sim* sim, const uint8_t* d_coef8_t, const float* d_scale8);

/* Where rows go.  d_log == NULL (or capacity 0) disables logging: only counters are kept. */
int rg_sim_set_log(rg_sim* sim, rg_event* d_log, uint64_t capacity);

/* Optional float64 side arrays of the log, `capacity` (of rg_sim_set_log) doubles each, caller-owned device
 * memory; either may be NULL.  d_ps[row] receives the propensity of bandit row `row` in float64 — the dtype of
 * the reference's `ps` column (abstract.py:283-290,318-327; IPS estimators divide by it) — and d_p_click[row]
 * the click probability ff(beta[a].omega + mu_bandit[a]) the click of that row was drawn with
 * (reco_env_v1.py:104-116).  Entries of organic rows are not written.  rg_sim_set_log detaches them. */
int rg_sim_set_log_aux(rg_sim* sim, double* d_ps, double* d_p_click);

/* NormalTimeGenerator only: d_time[row] receives the (float64) time of raw-log row `row`, every row (the 16-byte row's
 * `t` stays the per-user event index, which orders the log).  `capacity` doubles, like the log.  rg_sim_set_log detaches it. */
int rg_sim_set_log_time(rg_sim* sim, double* d_time);

/* RecoEnv1.reset + AbstractEnv.reset (reco_env_v1.py:78-82, abstract.py:90-103) for `n` users
 * with ids first_user_id .. first_user_id+n-1 at once: state <- organic, t <- 0,
 * omega <- sigma_omega_initial * Z(K).  Users with id < organic_only_below are the
 * `num_organic_offline_users` warm-up users of generate_logs (abstract.py:293-297): they emit
 * only their first organic session.  Re-seeds nothing; `seed`/`policy_seed` may be changed
 * between runs with rg_sim_reseed (reset_random_seed(epoch), abstract.py:59-62). */
int rg_sim_reset_users(rg_sim* sim, uint64_t first_user_id, uint64_t n,
                       uint64_t organic_only_below, void* stream);
int rg_sim_reseed(rg_sim* sim, uint64_t seed, uint64_t policy_seed);

/* One Markov transition for every live user (one emitted row per live user, plus the phantom
 * row of users that stop) — the batched form of AbstractEnv.step / step_offline /
 * generate_organic_sessions (abstract.py:105-239) and RecoEnv1.update_product_view /
 * draw_click / update_state (reco_env_v1.py:85-128).  With RG_POLICY_EXTERNAL,
 * d_actions[i] is the action for the i-th user of the reset range (read only for users in the
 * bandit state; a user that stops then gets no phantom row — the caller's agent owns it). */
int rg_sim_step(rg_sim* sim, const int32_t* d_actions, void* stream);

/* AbstractEnv.step for ONE user (the gym.Env episode API: reset / step, abstract.py:123-197) with a single read-back: needs
 * RG_POLICY_EXTERNAL and a one-user reset range.  `action` is the agent's action for a user in the bandit state (ignored in
 * the organic state).  One Markov transition; then *out (host memory) receives, through one pinned-memory copy and ONE stream
 * synchronisation, the row the step emitted, the user's state and clock after it, and the float64 side values of the row. */
typedef struct rg_step_result {
    rg_event row;        /* the emitted row (has_row = 0: none, e.g. no log attached) */
    int32_t state;       /* RG_STATE_* after the transition */
    int32_t has_row;
    double time;         /* the user's clock after the step (event index + 1 with the default time generator) */
    double ps;           /* float64 propensity of the row where the side array is attached, else the row's float32 value */
    double p_click;      /* click probability of a bandit row where that side array is attached, else 0 */
} rg_step_result;
int rg_sim_step_user(rg_sim* sim, int32_t action, rg_step_result* out, void* stream);

/* generate_logs' user loop (abstract.py:299-316) for all users at once: steps until every user
 * reached `stop` or max_steps transitions were made.  Synchronises `stream`.  With max_steps >=
 * 65536 ("to the end") the last users of the run (<= 4096 alive at a 16-step poll, fewer for tables larger than 10^4 x 20; RECOGYM_TAIL overrides) are walked
 * to their end one user per workgroup instead of step by step; rows, counters and the sorted
 * log are the same, RG_CNT_STEP then reports the longest trajectory.  Fails with RG_ELIMIT when the run is incomplete
 * (RG_CNT_EXACT_OVERFLOW != 0, or a user reached the step limit).  The sigma_omega = 0 user-major walk cannot overflow the
 * float64 scratch (it holds a row per user there), so only the step limit applies to it. */
int rg_sim_run(rg_sim* sim, uint32_t max_steps, void* stream);

/* Synchronises `stream` and copies RG_CNT_N counters to the host. */
int rg_sim_read_counters(rg_sim* sim, int64_t* out, void* stream);

/* Measurement aid for bench.py's roofline line: with profiling on, every step records HIP
 * events on the stream the kernels are launched on.  rg_sim_get_profile returns
 * out[0..3] = total milliseconds spent in the MFMA organic-draw kernel, in its search kernel
 * (sliced mode only), in the float64 resolve kernels and in the advance kernel; out[4] =
 * profiled steps; out[5] = milliseconds in the tail kernel (rg_sim_run finishes the last users
 * of a run user by user instead of step by step, or — sigma_omega = 0 — the whole run user-major: k_walk);
 * out[6], out[7] = milliseconds in round 1 / the later rounds of k_walk; out[8] = milliseconds in the frozen-LogReg act
 * kernels (k_logreg_select + k_logreg_acts; not part of out[3]); out[9] reserved.  `out` must hold 10 doubles.  Off by default. */
int rg_sim_set_profiling(rg_sim* sim, int on);
int rg_sim_get_profile(rg_sim* sim, double* out);

/* Current Markov state per user of the reset range (int8, RG_STATE_*), for the gym.Env
 * compatibility path.  d_state has n entries. */
int rg_sim_export_state(rg_sim* sim, int8_t* d_state, void* stream);
/* Copy omega (float64, user-major (n,K)) out, for debugging views (`env.omega`). */
int rg_sim_export_omega(rg_sim* sim, double* d_omega, void* stream);

/* The row order of generate_logs' DataFrame (abstract.py:299-316): the
 * step-major device log is scattered so that each user's rows are contiguous, ordered by t,
 * the phantom row last, users in id order.  Caller-owned device buffers:
 *   d_row_offsets  n+1 int64 — filled with the exclusive prefix of rows per user (last = total)
 *   d_scratch      n + ceil(n/256) int64
 *   d_sorted       sorted_capacity rows
 * Synchronises `stream` once (to read the emitted-row count); fails with RG_ELIMIT when the
 * log buffer overflowed. */
int rg_sim_sort_log(rg_sim* sim, int64_t* d_row_offsets, int64_t* d_scratch, rg_event* d_sorted,
                    uint64_t sorted_capacity, void* stream);

/* The side arrays of rg_sim_set_log_aux in the row order rg_sim_sort_log produced (`d_row_offsets` as filled
 * by it): NaN on organic rows, and for p_click on the phantom row (never drawn).  Either output may be NULL. */
int rg_sim_sort_log_aux(rg_sim* sim, const int64_t* d_row_offsets, double* d_sorted_ps,
                        double* d_sorted_p_click, uint64_t sorted_capacity, void* stream);

/* The time column in the row order of rg_sim_sort_log (phantom rows: the time their act would have had). */
int rg_sim_sort_log_time(rg_sim* sim, const int64_t* d_row_offsets, double* d_sorted_time, uint64_t sorted_capacity,
                         void* stream);
/* Current time of every user of the reset range (n float64; = its event index with the default generator). */
int rg_sim_export_time(rg_sim* sim, double* d_time, void* stream);

/* ---- test hooks (the parity suite's adversarial certificate test; not part of the reference surface) ----
 * rg_sim_debug_set_omega: overwrite omega of the reset range, (n, K) float64 user-major, right after
 * rg_sim_reset_users.  rg_sim_debug_set_uniforms: d_u[i] replaces the uniform of user index i's next organic
 * product draws (NULL restores the addressed draws).  rg_sim_debug_uncertified: d_flags[i] = 1 iff user
 * index i's organic draw of the LAST step was not certified by the matrix-core kernel and went to the
 * float64 resolve (n bytes). */
int rg_sim_debug_set_omega(rg_sim* sim, const double* d_omega, void* stream);
int rg_sim_debug_set_uniforms(rg_sim* sim, const double* d_u);
/* rg_sim_debug_set_row_base: the raw log of every later reset range starts at row `rows` instead of 0 (rg_sim_reset_users marks
 * the entries below it unused, so the attached log must hold them): the regression test of raw-row arithmetic beyond 2^31 rows
 * (a full-size log reaches that line at ~20 M users of BASELINE config 3) without simulating 20 M users.  0 restores the default. */
int rg_sim_debug_set_row_base(rg_sim* sim, uint64_t rows);
int rg_sim_debug_uncertified(rg_sim* sim, uint8_t* d_flags, void* stream);

/* ---- test hooks of the two "decide cheaply, float64 inside a band" paths of the user-major walk ----
 * rg_sim_debug_click_decisions: for user index i of the reset range, the click decision of a bandit event with action
 * d_actions[i] and uniform d_u[i] on the user's current omega (reco_env_v1.py:104-116), by the walk's fp32 form and in
 * float64: d_out[i] bit 0 = the fp32 form decided (outside its error margin), bit 1 = its decision, bit 2 = the
 * float64 decision.  Needs sigma_omega == 0 (where the walk runs).
 * rg_sim_debug_set_history: overwrite the view histories (ViewsFeaturesProvider, agents/abstract.py:347-358) of the reset
 * range right after rg_sim_reset_users: user index i has d_nd[i] distinct products d_products[i * stride + j]
 * (ascending) with d_counts[i * stride + j] views each.
 * rg_sim_debug_ouc_acts: OrganicUserEventCounterModel.act (organic_user_count.py:45-96) of every user index on its
 * current history with d_u1[i] as the uniform of the action draw: d_action / d_ps (float64) as logged, d_flags[i] = 1
 * iff the action was decided by the integer prefix walk (outside its 2^-36 band), 0 = float64 cdf walk. */
/* rg_sim_debug_walk_fate: after a sigma_omega == 0 run "to the end" (the user-major walk), d_flags[i] (n bytes) = bit 0: user
 * index i met a draw the fast certificate rejected (or was handed over by a draining wave) and went through the float64
 * batch and round 2; bit 1: its last events were walked by the last round (a wave per user).  The sampled-oracle parity
 * check picks users of every kind with it. */
int rg_sim_debug_walk_fate(rg_sim* sim, uint8_t* d_flags, void* stream);
int rg_sim_debug_click_decisions(rg_sim* sim, const int32_t* d_actions, const double* d_u, uint8_t* d_out, void* stream);
int rg_sim_debug_set_history(rg_sim* sim, const uint32_t* d_nd, const uint32_t* d_products, const uint32_t* d_counts,
                             uint32_t stride, void* stream);
int rg_sim_debug_ouc_acts(rg_sim* sim, const double* d_u1, int32_t* d_action, double* d_ps, uint8_t* d_flags, void* stream);

#ifdef __cplusplus
}
#endif

#endif /* RECOGYM_HIP_H */
