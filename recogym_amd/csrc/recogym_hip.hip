// recogym_hip.hip — librecogym_hip.so: the reco-gym-v1 step loop as batched CDNA4 (gfx950) kernels.
//
// What runs here (reference file:line each kernel takes over; see DESIGN.md for the data layout
// and the roofline of each kernel):
//
//   k_reset_users     RecoEnv1.reset / AbstractEnv.reset          reco_env_v1.py:78-82, abstract.py:90-103
//   k_draw_bf16p      RecoEnv1.update_product_view                reco_env_v1.py:119-128
//                     the default: logits on the 16-bit matrix pipe as a two-way fp16 (or three-way
//                     bf16) split of the fp32 operands, pipelined pairs of chunks, every index
//                     certified against float64 (search_and_emit); k_draw_bf16 = its
//                     one-accumulator form, k_draw_mfma = fp32 MFMA (K classes without a 16-bit
//                     instantiation), k_draw_search = the search of the product-sliced form
//   k_exact_sums_m / k_exact_pick (k_exact_sums_h: matrix and vector-ALU forms side by side; k_exact_sums: K > 64)
//                     the same draw in float64 for the draws the fast path cannot certify (dot products on the
//                     float64 matrix cores)
//   k_cache_finalize, k_walk2 / k_walk / k_walk_solo (sigma_omega == 0)
//                     the whole run user-major from a per-user cache of exp-sums: draw, policy act, click,
//                     transition and row of every event of a user on one lane (run_walk_pipe: every list length stays
//                     on the device; k_walk2's view-history line in LDS is compact and in prefix form)
//   k_advance         AbstractEnv.step / step_offline, RecoEnv1.draw_click / update_state, the
//                     policy's act (policy_act / logreg_act_wave) and the log rows of generate_logs
//                                                                 abstract.py:123-239,267-316
//                                                                 reco_env_v1.py:85-116
//   k_tail            all of the above for the last users of a run, one user per workgroup
//   k_repack_*        no reference counterpart: restores the locality of the per-user state
//   k_rows_per_user, k_scan_*, k_scatter_*
//                     row order of generate_logs' DataFrame       abstract.py:299-327
//
// Lock-step structure: every live user advances exactly one Markov transition per step, so the
// step index IS the per-user event time t (DefaultTimeGenerator).  Users that are in the
// organic state at step t sit in list_o[t&1], users in the bandit state in list_b[t&1]; a step
// reads those lists and appends survivors to the lists of step t+1.  All randomness is
// addressed by (seed, user, t, purpose) (include/recogym_rng.h), so results do not depend on
// list order, grid shape or the number of GPUs the users are sharded over.
//
// gfx950 only.  No CPU fallback: every compute entry point fails with RG_ENODEV without a device.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "../../include/recogym_hip.h"
#include "../../include/recogym_rng.h"

// Translation-unit parts.  The file compiles as ONE translation unit (RG_PART undefined) or, in parallel, as seven
// (-DRG_PART=1..7, linked into one library by __graft_entry__.build()): part 1 holds the host code and the small
// kernels; every large kernel family lives in a part of its own (2 float64 resolve, 3 fp32 / lean bf16 sweeps, 4 the
// pipelined sweep + per-user cache kernels, 5 the wide-K sweep, 6 advance / tail / frozen LogReg, 7 the user-major
// walk) and hands its kernels to the host code through the *_kernel_for functions.  Types and device helpers are
// shared by all parts (namespace rgk, identical in every unit).
#ifndef RG_PART
#define RG_PART 0
#endif
#define RG_HAS(p) (RG_PART == 0 || RG_PART == (p))

namespace rgk {

constexpr uint32_t kMaxSteps = 1u << 16;       // P(a user survives that long) ~ exp(-650)
constexpr int kBlock = 256;                    // 4 waves of 64
constexpr int kMaxGrid = 4096;
constexpr uint32_t kDefaultHistoryCap = 256;
// runs smaller than this keep slot == user index throughout (RECOGYM_REPACK_MIN overrides: tests)
inline uint64_t repack_min_users() {
    const char* e = getenv("RECOGYM_REPACK_MIN");
    return e ? static_cast<uint64_t>(strtoull(e, nullptr, 10)) : (1ull << 18);
}

inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(RG_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_));        \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Everything a kernel needs, passed by value.
struct DevSim {
    // configuration
    uint32_t P, K;
    uint64_t seed, policy_seed;
    double cdf_o0, cdf_o1, cdf_b0, cdf_b1;   // normalised cumulative transition rows
    double sigma0, sigma_omega;
    uint32_t change_omega_for_bandits, policy;
    uint32_t ouc_select_randomly, ouc_exploit_explore, ouc_reverse_pop, hist_cap;
    double ouc_epsilon;
    // user range
    uint64_t first_user;
    uint32_t n_users;         // users of the current reset range
    uint32_t n_cap, n_pad;    // users the workspace was carved for (list stride), padded to 64
    uint64_t organic_only_below;
    // tables (caller-owned float64) and fp32 copies (workspace)
    const double* gamma; const double* mu_o; const double* beta; const double* mu_b;
    float* gamma32; float* mu32;   // [P_pad][KS] (k >= K zero, rows >= P zero) / [P_pad] (-inf pad)
    uint32_t has_g32t;        // gamma32t is there (gamma32t_wanted)
    float* gamma32t;          // [n_chunks][2 KH][32]: the same values chunk by chunk, k-major inside a chunk — a lane per
                              // product reads one k of its chunk as one coalesced 128-byte run (k_draw_cached)
    double* gammaT;           // [K][PT] float64 transpose of Gamma, PT = P rounded up to 64 (coalesced f64 draw)
    uint32_t PT;
    double* gamma_rm;         // [PT][4 XKB + 4] row-major float64 Gamma, k zero-padded to 4 XKB, then mu_o (-inf beyond P):
    uint32_t XKB;             // one row = what one product costs the user-per-lane float64 kernel in scalar loads; 0 = K > 64
    float* exact_ref;         // [n_users] log2-scaled reference of a draw handed to the float64 kernel
    double* exact_sums;       // [exact_rows][PT/64] float64 exp-sum of every 64-product chunk
    uint32_t exact_rows;      // rows of exact_sums: n_cap where they are per-user constants (sigma_omega = 0 cache) or every
                              // draw goes through float64; else max(4096, n_cap / 8) — a step's uncertified draws (a few percent
                              // of its organic users) are resolved in batches of that many list entries
    uint32_t walk_handover;   // k_walk: live lanes at which a wave whose queue is empty passes its users to the next round (0: never)
    uint32_t walk_refill;     // k_walk: free lanes of a wave at which it takes new users from the queue
    uint32_t walk_bias;       // k_walk: 0 = both event kinds every iteration; else one kind, organic when n_org * walk_bias >= n_bandit * 4
    uint32_t walk_click_batch;   // k_walk2: lanes waiting for ctr (kWClick) at which the wave takes them (0: in the bandit iteration itself)
    uint32_t walk_search_batch;  // k_walk2: lanes that missed the memo at which the wave runs the search (its chunk passes take 8 users each)
    uint32_t walk_line64;        // k_walk2 (host side: which instantiation): round 3's 64-bit history line of 15 products (RECOGYM_WALK_HIST=1)
    uint32_t exact_base;      // first exact_list entry of the batch being resolved
    uint32_t exact_last;      // this is the last batch launched for the step
    float2* sc_scratch;       // [kMaxGrid*4 waves][kMaxSC][32] {sum, reference} of the MFMA draw kernel
    float* chunk_scratch;     // [kMaxGrid*4 waves][n_chunks][32] exp-sum of every 32-product chunk
    float* stats;             // [2*KH] max_p |Gamma[p][k]|, then max_p ||Gamma[p]||_2, max_p |mu_o[p]|
    // geometry of the MFMA draw kernel
    uint32_t KH;              // MFMA k-steps per chunk (each 32x32x2 step consumes 2 k); 0 = no MFMA path
    uint32_t KS;              // row stride of gamma32 / the LDS tile, floats (== 2 mod 4: conflict-free b64)
    uint32_t TP;              // products per LDS tile (multiple of 32)
    uint32_t P_pad;           // rows of gamma32 / mu32
    uint32_t n_chunks;        // ceil(P / 32)
    uint32_t sc_chunks;       // chunks per stored partial sum ("super-chunk")
    uint32_t n_sc;            // super-chunks (<= kMaxSC)
    uint32_t use_mfma;        // 0 = float64 only, 1 = fp32 MFMA kernel, 2 = split-bf16 MFMA kernel
    // split-bf16 kernel geometry: A row = [G1|G2|G3] (3K bf16, zero padded to 16*N1), row stride RS bytes
    uint32_t N1, N2, N3;      // k-steps of the three MFMA groups (B = w1 / w2 / w3)
    uint32_t f16;             // 1: gsplit holds the two-way fp16 split [G1|G2|G1|0..|1] (one group of N1 k-steps)
    uint32_t wide;            // 1: served by k_draw_f16w (21 < K <= 64: 512-thread blocks, 256 users per table pass)
    uint32_t RS;              // row stride of gsplit / its LDS tile, bytes ((RS/16) odd: conflict-free b128)
    uint32_t TPB;             // products per LDS tile of the bf16 kernel
    unsigned short* gsplit;   // [P_pad][RS/2] bf16 three-way split of fl32(Gamma log2 e), then 1,1,1 in the last 3 columns of 16*N1
    float* mu32s;             // [P_pad] fl32(mu_o log2 e), -inf beyond P
    uint32_t ablate;          // timing experiments only (RECOGYM_ABLATE); results are wrong when non-zero
    // sigma_omega == 0: a user's omega — hence its softmax — never changes after the reset, so the exp-sums of its
    // first product sweep (step 0: every user starts organic) are kept PER USER (index = user index, never moved by
    // the repack; row n_cap is a dummy that inactive lanes write) and every later draw of that user is only the
    // search over them (k_draw_search), with the same certificate and the same float64 resolve
    uint32_t use_cache;
    float2* cache_rec;        // [n_cap + 1][kMaxSC] {sum, reference} of every super-chunk
    float* cache_chunk;       // [n_cap + 1][n_chunks] exp-sum of every 32-product chunk
    float* beta32;            // [P][KB4] fp32 copy of beta (rows padded with zeros to KB4 = K rounded up to 4): k_walk's click fast path
    uint32_t KB4;
    uint8_t* cache_resc;      // [n_cap + 1] re-references of the sweep (certificate budget)
    // what every later draw of a user starts from, one contiguous row per user (k_cache_finalize builds it from the
    // records above right after step 0): [0,32) super-chunk sums scaled to the common reference | 32: that reference,
    // 33: the certificate's delta (rounded up), 34-35: - | [36,44) 32 int8: reference offset of every super-chunk
    // (scale of its chunk sums) | [44, 44 + 2 KH) omega32.  256 bytes at K <= 20: two lines instead of six
    float* cache_row; uint32_t cache_row_f;   // row stride in floats (multiple of 32)
    // k_walk2: [n_cap + 1][32] hot row {S, delta, Q, n_hot | 9 x {product, u_lo, u_hi}} and [n_cap + 1][32] fp32 prefix at the end of
    // every super-chunk (cache_chunk holds the chunk-level prefixes once k_cache_prefix ran)
    float* walk_hot; float* walk_scp;
    // user-major walk of the sigma_omega == 0 mode (k_walk): users parked at their first uncertified draw
    uint32_t fin_in_sweep;    // run_walk_pipe: the prefix-form sweep also leaves what k_cache_finalize + k_cache_prefix would (the
                              // user's Q, delta, omega32, empty memo) for every user whose reference never moved; those two
                              // kernels then only visit the (rare) users it did move for (cache_resc != 0)
    uint32_t sweep_only;      // the step-0 sweep only fills the cache (no search, no rows): k_walk draws t = 0 too;
                              // 2: ... and k_draw_bf16p stores the chunk sums as running PREFIXES on the reference in force (and the
                              // prefix at every super-chunk end in walk_scp): k_walk2's form, no conversion pass
    uint32_t* park_list;      // [n_cap + 64] user indices, reserved in chunks of 64 (0xFFFFFFFF = unused entry)
    uint32_t* park_t;         // [n_cap] time of the parked draw
    uint8_t* f64_valid;       // [n_cap] exact_sums / exact_ref rows (indexed by user index in this mode) are valid
    // The walked run as a pipeline over user groups (run_walk_pipe): every launch works on the user-index range
    // [grp_lo, grp_lo + grp_n) and on work queues of its own, so that the launches of different groups can be in flight at once
    // on different streams.  Outside the pipeline: the whole reset range and the two counters[] slots.
    uint32_t grp_lo, grp_n;
    uint32_t list_in;         // first park_list entry of the list k_exact_sums_h / k_exact_prefix read
    unsigned long long* q_ticket;        // ticket counter of the launch's work queue
    unsigned long long* q_park;          // entries reserved so far in the list the launch appends to (blocks of 64)
    const unsigned long long* q_count;   // non-null: the length of the list the launch reads is *q_count, known on the device
                                         // only (the argument is then an upper bound used for nothing but launch shapes)
    unsigned long long* walk_ctl;        // [kWalkCtlWords] the queues' counters (workspace)
    unsigned long long* step1_buf;       // [16] rg_sim_step_user: word 0 = the action, words 8.. = the packed result
    uint32_t* exact_cnt_b;    // [kMaxSteps+2] draws to resolve whose float64 sums are already there: they sit at the
                              // BACK of exact_list (entry n_cap - 1 - i); those that need the sums at the front
    // state (workspace)
    double* omega;            // [n_pad][OMS] user-major (OMS = K rounded up to 2): a user's vector is contiguous,
                              // so the scrambled order of the live lists costs at most one extra cache line per user
    uint32_t OMS;
    uint32_t* list;           // [2 parity][2 state][n_users]
    uint32_t* step_cnt;       // [kMaxSteps+2][2]: users in organic / bandit state at step t
    uint64_t* log_base;       // [kMaxSteps+2]: first log row of step t
    uint32_t* exact_list;     // [n_users] organic users whose draw needs the float64 path
    uint32_t* exact_cnt;      // [kMaxSteps+2]
    uint32_t* n_events;       // [n_users] rows the user emitted (set when it leaves); these three are indexed by
    rg_event* phantom;        // [n_users] trailing undrawn bandit row                  USER INDEX (uid), not by slot
    uint8_t* has_phantom;     // [n_users]
    // per-user view history (OUC / frozen LogReg policies), user-major rows of hist_cap 64-bit entries:
    //   entry 0        header: (views so far << 32) | distinct products viewed (nd)
    //   entries 1..nd  (product << 32) | view count, ascending by product (== ascending as integers)
    // one 128-byte line holds the header and the first 15 products: most users' whole history
    unsigned long long* hist;
    uint32_t* lpv;            // [n_users] last product viewed (RG_POLICY_LAST_VIEW_TABLE)
    uint32_t* uid;            // [n_users] slot -> user index (user id = first_user + uid[slot]); identity until a repack
    // second copy of the slot-indexed state: k_repack_copy moves the live users' state into it, densely
    // and in list order, and the host swaps the pointers (restores the locality the lists lose over time)
    double* omega_alt; unsigned long long* hist_alt; uint32_t* lpv_alt; uint32_t* uid_alt;
    const int32_t* pol_table; const float* pol_ps;   // caller-owned per-product tables of that policy
    const double* lr_coef_t; const double* lr_intercept; const int32_t* lr_classes; uint32_t lr_n;   // RG_POLICY_LOGREG_FROZEN
    // the policy's act depends on the view history only: it is computed when the history has changed since the last act
    // (lr_dirty, set by history_add) and kept per user; k_logreg_select / k_logreg_acts run before k_advance
    const float* lr_coef32_t; const float* lr_intercept32; const float* lr_wmax; float lr_bmax;   // fp32 copies + max_c |coef[p][c]|, max |b|
    const unsigned short* lr_coef16_t;   // fp16 copy of coef^T (screening pass of k_logreg_acts16), or null
    uint32_t* lr_action;      // [n_cap] by user index: action of the user's current history
    uint8_t* lr_dirty;        // [n_cap] by user index
    uint32_t* lr_list;        // [n_cap] slots whose act is to be computed this step
    uint32_t* lr_cnt;         // [kMaxSteps + 2]
    uint32_t* lr_part;        // [n_cap][kLrSplit][kLrPartWords]: the screen's result per listed act and class range
    // omega drift of a lock-step step (sigma_omega > 0): k_advance lists the users whose transition drifts omega, k_drift applies
    // the K normals a lane per (user, Box-Muller pair) — ~2 600 float64 instructions per drifting user that only ~22 % of
    // k_advance's lanes would execute (the others idle through them)
    uint32_t* drift_list;     // [n_cap] slots
    double* drift_sig;        // [n_cap] sigma_omega x time delta of the entry (NormalTimeGenerator only; else sigma_omega)
    uint32_t* drift_cnt;      // [kMaxSteps + 2]
    unsigned long long* counters;   // [RG_CNT_N]
    // log
    rg_event* log; uint64_t log_cap;
    // optional float64 side arrays, one entry per log row (same raw position): the propensity `ps` as the
    // reference logs it (float64, abstract.py:318-327) and the click probability of the row (reco_env_v1.py:104-116)
    double* aux_ps; double* aux_pclick;
    double* phantom_ps;       // [n_users] float64 propensity of the phantom row
    // test hooks (rg_sim_debug_*): per-user-index uniforms replacing the organic draw's u at the next step
    const double* u_override;
    // NormalTimeGenerator (time_mode = 1, normal_time_generator.py:23-26; lock-step only)
    uint32_t time_mode;
    double time_mu, time_sigma;
    double* utime;            // [n_cap] current time of every user (index = user index)
    double* phantom_time;     // [n_cap] time of the phantom row
    double* aux_time;         // optional side array of the log: time of every raw row
};

}  // namespace rgk
using namespace rgk;

// Run-path options: every switch the launch code consults, read ONCE (rg_sim_create, from the RECOGYM_* environment: the A/B
// tests' way in) and settable through rg_sim_set_option — no getenv on the run path.
struct RunOpts {
    int exact_tile;          // RECOGYM_EXACT_TILE: the K > 64 tile kernel for every float64 resolve
    int exact_mix;           // RECOGYM_EXACT_MIX: groups of every 8 of the walk's float64 batch in the matrix form (8 = all)
    int resident_grid;       // RECOGYM_RESIDENT_GRID: sweep grid = the resident blocks
    int slices;              // RECOGYM_SLICES: product slices of the lock-step sweep (-1 = by population)
    int sweep_prefix_off;    // RECOGYM_SWEEP_PREFIX_OFF: the sweep stores sums, k_cache_prefix converts them
    int debug;               // RECOGYM_DEBUG
    unsigned long long repack_min;   // RECOGYM_REPACK_MIN: users below which slot == user index throughout
};

struct rg_sim {
    rg_config cfg;
    DevSim d;
    RunOpts opt;
    void* workspace;
    size_t workspace_bytes;
    uint32_t t;               // next step to run
    uint32_t live_upper;      // upper bound of live users (for grid sizing)
    bool tables_set, users_reset;
    bool repacked;            // slots no longer equal user indices (since the last reset)
    bool walk;                // rg_sim_run "to the end" walks the run user-major (k_walk) instead of step-major
    int walk_occ;             // blocks per CU the walk kernel is compiled for (k_walk: 3; k_walk2: 3 at K <= 20, 2 at K <= 32)
    bool walk2;               // the walk is k_walk2 (prefix sums + memo; RECOGYM_WALK=1 keeps k_walk)
    bool walk_solo;           // its last round is k_walk_solo (RECOGYM_WALK_SOLO=0: k_walk2's)
    int n_cus;                // compute units of the device (grid of the persistent walk kernel)
    double prof_walk_ms[2];   // round 1 / round 2 of k_walk
    // the walked run as a pipeline over user groups on two or three streams (run_walk_pipe)
    int pipe_groups;          // user groups (1 = one group: the serial chain without host read-backs); 0 = run_walk (host-side counts)
    int pipe_mode;            // 0: every launch on the caller's stream; 1: float64 batch + round 2 of a group on a second stream;
                              // 2: ... and the sweeps on a third
    int pipe_occ1, pipe_occ2; // blocks per CU of the round-1 / round-2 grids (<= what the kernel is compiled for)
    int pipe_xblocks;         // blocks of the float64 batch's grid
    bool fin_in_sweep;        // the sweep of run_walk_pipe leaves the finalize / prefix kernels' output itself (RECOGYM_FIN_IN_SWEEP=0: A/B)
    uint32_t pipe_min_users;  // users of a group (and of a pipelined run) at least: an unsliced sweep's 1024 user tiles (RECOGYM_PIPE_MIN: tests)
    hipStream_t pipe_streams[2];
    std::vector<hipEvent_t> pipe_events;   // ordering events (no timing), created once
    double prof_pipe_ms;      // profiling: wall time of the pipelined runs (its kernels' own times overlap)
    // rg_sim_debug_walk_fate: where the last walked run left the list of its last round (null: there was none)
    uint32_t fate_base; const unsigned long long* fate_count;
    uint32_t repack_every;    // steps between repacks (RECOGYM_REPACK, 0 = never)
    uint32_t tail_below;      // rg_sim_run hands the run to k_tail once at most this many users live (RECOGYM_TAIL, 0 = never)
    double prof_tail_ms;
    uint32_t* h_pinned;       // 4 x u32 staging for the live-count readback
    char* h_step;             // 128 pinned bytes of rg_sim_step_user: the action going down, the packed result coming back
    size_t mfma_smem, bf16_smem;
    void (*bf16_kernel)(DevSim, uint32_t, uint32_t);
    uint32_t draw_threads, draw_users;   // block size of that kernel and the users one block sweeps for (256 / 128; wide K: 512 / 256)
    bool profiling;
    std::vector<hipEvent_t> prof_events;   // 6 per profiled step: before draw, after mfma, after search, after exact, after the frozen LogReg acts, after advance
    size_t prof_used;
    double prof_ms[5];                     // sweep, search, float64 resolve, LogReg acts, advance
    uint64_t prof_launches;
};

namespace rgk {

// kernels of the other parts, as the host code (part 1) gets them
typedef void (*exact_h_kernel_t)(DevSim, uint32_t, uint32_t);
typedef void (*exact_m_kernel_t)(DevSim, uint32_t, int, int, uint32_t);
typedef void (*exact_pick_kernel_t)(DevSim, uint32_t, int, uint32_t);
typedef void (*finalize_kernel_t)(DevSim);
typedef void (*cached_kernel_t)(DevSim, uint32_t);
typedef void (*draw_kernel_t)(DevSim, uint32_t, uint32_t);
typedef void (*search_kernel_t)(DevSim, uint32_t);
typedef void (*mfma_kernel_t)(DevSim, uint32_t);
typedef void (*advance_kernel_t)(DevSim, uint32_t, const int32_t*);
typedef void (*walk_kernel_t)(DevSim, uint32_t, int, uint32_t, uint32_t, uint32_t);
exact_h_kernel_t exact_h_kernel_for(uint32_t kb);          // part 2
exact_m_kernel_t exact_m_kernel_for(uint32_t kb);
exact_m_kernel_t exact_tile_kernel();                      // k_exact_sums
exact_h_kernel_t exact_ref_kernel();                       // k_exact_ref
exact_pick_kernel_t exact_pick_kernel();                   // k_exact_pick
search_kernel_t search_kernel_for(const DevSim& d);        // part 3
draw_kernel_t bf16_kernel_for(const DevSim& d);
mfma_kernel_t mfma_kernel_for(uint32_t KH);
finalize_kernel_t finalize_kernel_for(const DevSim& d);    // part 4
cached_kernel_t cached_kernel_for(const DevSim& d);
draw_kernel_t bf16p_kernel_for(const DevSim& d);
draw_kernel_t f16w_kernel_for(const DevSim& d);            // part 5
search_kernel_t drift_kernel();                            // part 6
search_kernel_t logreg_select_kernel();
search_kernel_t logreg_acts_kernel();
search_kernel_t logreg_screen_kernel();
search_kernel_t logreg_decide_kernel();
advance_kernel_t advance_kernel();
search_kernel_t tail_kernel();
walk_kernel_t walk_kernel_for(const DevSim& d, int occ);   // part 7
walk_kernel_t walk2_kernel_for(const DevSim& d, int occ);  // (nullptr: this configuration keeps k_walk)
typedef void (*solo_kernel_t)(DevSim, uint32_t, uint32_t, uint32_t);
solo_kernel_t solo_kernel_for(const DevSim& d);            // (nullptr: the last round is k_walk2's too)
void (*cache_prefix_kernel())(DevSim, int);
void (*exact_prefix_kernel())(DevSim, uint32_t);

// ------------------------------------------------------------------------------------------
// workspace carving (host)
// ------------------------------------------------------------------------------------------
struct Carve {
    size_t off = 0;
    char* base;
    explicit Carve(void* b) : base(static_cast<char*>(b)) {}
    template <class T> T* take(size_t n) {
        off = align_up(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

constexpr uint32_t kMaxSC = 32;           // stored partial sums per user in the MFMA draw kernel
constexpr uint32_t kAhatGrid = 64;        // stats[2 KH + 2 + i] = max_p (|mu_p| + ||Gamma_p||_2 (i + 1) / 4): the logit bound, jointly over p
constexpr uint32_t kHoleCode = 0xFFFFFFFFu;   // rg_event.code of an unused raw-log entry (no real row has every bit set: P < 2^29)
// timing experiments of the walk (RECOGYM_ABLATE bits 16-22: see DESIGN.md) exist in -DRG_WALK_TIMING builds only
#ifdef RG_WALK_TIMING
#define RG_WALK_ABL(bit) (d.ablate & (1u << (bit)))
#else
#define RG_WALK_ABL(bit) (false)
#endif

constexpr int kCntTailRows = 16, kCntTailOrganic = 17, kCntTailBandit = 18, kCntTailMaxT = 19, kCntTailTicket = 20,
              kCntTailLimit = 21, kCntWalkTicket = 22, kCntParkCnt = 23;   // internal slots of counters[] (RG_CNT_N = 24)
constexpr int kCntWalkHits = RG_CNT_MEMO_HITS;
// the walked run as a pipeline over user groups (run_walk_pipe): at most kMaxWalkGroups groups; a park_list region per group
// holds its users plus the 64-entry blocks its waves leave part-used (<= 2 per wave: parked users, hand-over); walk_ctl =
// 8 counters per group {round-1 ticket, round-1 list length, float64 batch ticket, round-2 ticket, -...} and, in block
// kMaxWalkGroups, {last round's list length, last round's ticket}.  The last round's list: kParkSlack entries per group
// (a wave of a round 2 hands over once: <= 2 blocks).  Walk grids are capped at kMaxWalkWaves waves.
constexpr uint32_t kMaxWalkGroups = 16;
constexpr uint32_t kMaxWalkWaves = 4096;
constexpr uint32_t kParkSlack = 2u * 64u * kMaxWalkWaves;
constexpr uint32_t kWalkCtlWords = 8u * (kMaxWalkGroups + 1u);

struct Geom { uint32_t KH, KS, TP, P_pad, n_chunks, sc_chunks, n_sc, N1, N2, N3, RS, TPB, F16; };

inline Geom geom_of(const rg_config& c) {
    Geom g{};
    const uint32_t need = (c.K + 1) / 2;
    const uint32_t opts[] = {4, 10, 16, 32, 64};
    for (uint32_t o : opts) if (!g.KH && need <= o) g.KH = o;
    if (!g.KH) return g;                            // K > 128: float64 kernel only
    g.KS = 2 * g.KH;
    while (g.KS % 4 != 2) ++g.KS;
    g.TP = 256;
    // (never below 64 products: k_draw_mfma consumes the tile in PAIRS of 32-product chunks.  KH = 64 takes 2 x 33 KB
    // of tile + 64 KB of omega stage = 133 KB of the CU's 160 KB)
    while (g.TP > 64 && static_cast<size_t>(g.TP) * g.KS * 4 > 24 * 1024) g.TP /= 2;
    g.P_pad = static_cast<uint32_t>(align_up(c.num_products, 256)) + 256;
    g.n_chunks = (c.num_products + 31) / 32;
    g.n_chunks = (g.n_chunks + 3) & ~3u;            // chunks are processed in pairs of pairs
    {   // split-bf16 classes (N1,N2,N3): smallest class with 3K <= 16 N1, 2K <= 16 N2, K <= 16 N3
        const uint32_t cls[][3] = {{1, 1, 1}, {2, 1, 1}, {3, 2, 1}, {4, 3, 2}, {6, 4, 2}, {12, 8, 4}};
        for (const auto& c3 : cls)
            if (!g.N1 && 3 * c.K + 3 <= 16 * c3[0] && 2 * c.K <= 16 * c3[1] && c.K <= 16 * c3[2]) {
                g.N1 = c3[0]; g.N2 = c3[1]; g.N3 = c3[2];
            }
        // two-way fp16 split (one MFMA group: A = [G1|G2|G1|..|1], B = [w1|w1|w2|..|-q]) where 3K + 1 columns fit
        // 64 and a kernel exists for (KH, N1); RECOGYM_DRAW=bf16 / RECOGYM_BF16=lean keep the three-way bf16 split
        const char* e_draw = getenv("RECOGYM_DRAW");
        const char* e_lean = getenv("RECOGYM_BF16");
        const bool want_f16 = !(e_draw && !strcmp(e_draw, "bf16")) && !(e_lean && !strcmp(e_lean, "lean"));
        if (want_f16 && 3 * c.K + 1 <= 64 && g.KH <= 16) {
            g.F16 = 1;
            g.N1 = (3 * c.K + 1 + 15) / 16; g.N2 = 0; g.N3 = 0;
        }
        // wide embeddings (21 < K <= 64): the same two-way fp16 split, k_draw_f16w (N1 classes 7 / 10 / 13 k-steps,
        // tiles of one pair of chunks); RECOGYM_F16W=0 keeps the older choice (bf16 classes / fp32 MFMA)
        const char* e_w = getenv("RECOGYM_F16W");
        if (want_f16 && !g.F16 && c.K > 21 && c.K <= 64 && (g.KH == 16 || g.KH == 32) && !(e_w && e_w[0] == '0')) {
            g.F16 = 2;
            g.N1 = 3 * c.K + 1 <= 112 ? 7 : (3 * c.K + 1 <= 160 ? 10 : 13); g.N2 = 0; g.N3 = 0;
        }
        if (g.N1) {
            g.RS = 32 * g.N1 + 16;
            g.TPB = g.F16 == 2 ? 64 : 128;          // 4 chunks per tile: the kernel walks pairs of pairs (wide: one pair)
        }
    }
    g.sc_chunks = (g.n_chunks + kMaxSC - 1) / kMaxSC;
    g.sc_chunks = (g.sc_chunks + 3) & ~3u;
    g.n_sc = (g.n_chunks + g.sc_chunks - 1) / g.sc_chunks;
    return g;
}

// the per-user sum cache exists where omega cannot change (sigma_omega == 0) and a 16-bit MFMA kernel class serves K
// (RECOGYM_CACHE=0: A/B tests)
inline bool cache_wanted(const rg_config& c, const Geom& g) {
    const char* e = getenv("RECOGYM_CACHE");
    return c.sigma_omega == 0.0 && g.N1 != 0 && !(e && e[0] == '0');
}

// The chunk-major fp32 copy of Gamma (gamma32t): the recompute of a draw's chunk reads it as one 128-byte run per k and user
// (eight lanes per user); the row-major gather it replaces was address-rate-bound.  The walk's and the cached draw's searches
// need it, and the lock-step search of K <= 32 uses it too.
inline bool gamma32t_wanted(const rg_config& c, const Geom& g) { return g.KH != 0 && (cache_wanted(c, g) || g.KH <= 16); }

// rows of the float64 chunk-sum scratch (see DevSim::exact_rows)
inline size_t exact_rows_of(const rg_config& c, const Geom& g, uint64_t n) {
    const char* e = getenv("RECOGYM_DRAW");
    const char* f = getenv("RECOGYM_FORCE_EXACT");
    const bool all_f64 = !g.KH || (e && !strcmp(e, "f64")) || (f && f[0] == '1');
    if (cache_wanted(c, g) || all_f64) return n;
    const uint64_t r = n / 8;
    return r < 4096 ? (n < 4096 ? n : 4096) : r;
}

inline size_t bf16_smem_bytes(const Geom& g, uint32_t K, uint32_t buffers) {
    // split tiles + mu tiles (2 buffers: lean kernel, 3: pipelined kernel) + the per-wave omega32 stage [4][32][K]
    return buffers * (static_cast<size_t>(g.TPB) * g.RS + g.TPB * 4) + 4 * 32 * static_cast<size_t>(K) * 4 + 256;
}

inline size_t mfma_smem_bytes(const Geom& g) {
    return sizeof(float) * (2 * (static_cast<size_t>(g.TP) * g.KS + g.TP) + 64 + 4 * 32 * 2 * g.KH);   // tiles + omega stage
}

// K classes of the user-per-lane float64 kernel (omega lives in 8 XKB registers per lane)
inline uint32_t exact_kb_of(uint32_t K) {
    const uint32_t opts[] = {1, 2, 3, 4, 5, 6, 8, 12, 16};
    for (uint32_t o : opts) if (K <= 4 * o) return o;
    return 0;
}

inline uint32_t hist_cap_of(const rg_config& c) {
    if (c.policy != RG_POLICY_ORGANIC_USER_COUNT && c.policy != RG_POLICY_LOGREG_FROZEN) return 0;
    // entries per row: the header + the distinct products kept, rounded up to whole 128-byte lines of 16 entries (what
    // the register paths load at a time)
    return ((c.ouc_history_cap ? c.ouc_history_cap : kDefaultHistoryCap - 1u) + 1u + 15u) & ~15u;
}

inline size_t carve_all(const rg_config& c, uint64_t n, void* base, DevSim* d) {
    Carve w(base);
    const size_t n_pad = align_up(n, 64);
    const size_t P = c.num_products, K = c.K;
    const Geom g = geom_of(c);
    (void)P; (void)K;
    float* gamma32 = w.take<float>(static_cast<size_t>(g.P_pad) * (g.KS ? g.KS : 1));
    float* mu32 = w.take<float>(g.P_pad ? g.P_pad : 1);
    float* gamma32t = w.take<float>(gamma32t_wanted(c, g) ? static_cast<size_t>(g.n_chunks) * 2 * g.KH * 32 : 1);
    float* stats = w.take<float>(2 * g.KH + 2 + kAhatGrid);
    const size_t PT = align_up(P, 64);
    double* gammaT = w.take<double>(K * PT);
    const uint32_t xkb = exact_kb_of(c.K);
    double* gamma_rm = w.take<double>(xkb ? PT * (4 * static_cast<size_t>(xkb) + 4) : 1);
    float* exact_ref = w.take<float>(n);
    const size_t exact_rows = exact_rows_of(c, g, n);
    double* exact_sums = w.take<double>(exact_rows * (PT / 64));
    unsigned short* gsplit = w.take<unsigned short>(g.N1 ? static_cast<size_t>(g.P_pad) * (g.RS / 2) : 1);
    float* mu32s = w.take<float>(g.N1 ? g.P_pad : 1);
    float2* sc_scratch = w.take<float2>(g.KH ? static_cast<size_t>(kMaxGrid) * 4 * kMaxSC * 32 : 1);
    float* chunk_scratch = w.take<float>(g.KH ? static_cast<size_t>(kMaxGrid) * 4 * g.n_chunks * 32 : 1);
    double* omega = w.take<double>(((K + 1) & ~static_cast<size_t>(1)) * n_pad);
    uint32_t* list = w.take<uint32_t>(4 * n);
    uint32_t* step_cnt = w.take<uint32_t>(2 * (kMaxSteps + 2));
    uint64_t* log_base = w.take<uint64_t>(kMaxSteps + 2);
    uint32_t* exact_list = w.take<uint32_t>(n);
    uint32_t* exact_cnt = w.take<uint32_t>(kMaxSteps + 2);
    uint32_t* n_events = w.take<uint32_t>(n);
    rg_event* phantom = w.take<rg_event>(n);
    uint8_t* has_phantom = w.take<uint8_t>(n);
    const size_t hc = hist_cap_of(c);
    unsigned long long* hist = w.take<unsigned long long>(hc * n_pad);
    uint32_t* lpv = w.take<uint32_t>(c.policy == RG_POLICY_LAST_VIEW_TABLE ? n : 1);
    unsigned long long* counters = w.take<unsigned long long>(RG_CNT_N);
    const bool drifts = c.sigma_omega != 0.0;
    uint32_t* drift_list = w.take<uint32_t>(drifts ? n : 1);
    double* drift_sig = w.take<double>(drifts && c.time_mode ? n : 1);
    uint32_t* drift_cnt = w.take<uint32_t>(drifts ? kMaxSteps + 2 : 1);
    uint32_t* uid = w.take<uint32_t>(n);
    double* phantom_ps = w.take<double>(n);
    double* utime = w.take<double>(c.time_mode ? n : 1);
    double* phantom_time = w.take<double>(c.time_mode ? n : 1);
    const bool cache = cache_wanted(c, g);
    float2* cache_rec = w.take<float2>(cache ? (n + 1) * kMaxSC : 1);
    float* cache_chunk = w.take<float>(cache ? (n + 1) * static_cast<size_t>(g.n_chunks) : 1);
    uint8_t* cache_resc = w.take<uint8_t>(cache ? n + 1 : 1);
    const size_t KB4 = (K + 3) & ~static_cast<size_t>(3);
    float* beta32 = w.take<float>(cache ? P * KB4 : 4);
    const uint32_t cache_row_f = (44u + 2u * g.KH + 31u) & ~31u;
    float* cache_row = w.take<float>(cache ? (n + 1) * static_cast<size_t>(cache_row_f) : 1);
    float* walk_hot = w.take<float>(cache ? (n + 1) * 32 : 1);
    float* walk_scp = w.take<float>(cache ? (n + 1) * static_cast<size_t>(kMaxSC) : 1);
    uint8_t* f64_valid = w.take<uint8_t>(cache ? n : 1);
    uint32_t* exact_cnt_b = w.take<uint32_t>(kMaxSteps + 2);
    const bool lr = c.policy == RG_POLICY_LOGREG_FROZEN;
    uint32_t* lr_action = w.take<uint32_t>(lr ? n : 1);
    uint8_t* lr_dirty = w.take<uint8_t>(lr ? n : 1);
    uint32_t* lr_list = w.take<uint32_t>(lr ? n : 1);
    uint32_t* lr_cnt = w.take<uint32_t>(lr ? kMaxSteps + 2 : 1);
    uint32_t* lr_part = w.take<uint32_t>(lr ? n * static_cast<size_t>(8 * (4 + 2 * 8)) : 1);       // kLrSplit x kLrPartWords
    // round 1's list, then round 2's hand-overs, 64-entry blocks per wave; the pipeline: a region per user group (its users +
    // kParkSlack for the blocks its waves leave part-used) and one for the last round's list
    uint32_t* park_list = w.take<uint32_t>(cache ? n + 128 + static_cast<size_t>(2 * kMaxWalkGroups) * kParkSlack : 1);
    unsigned long long* walk_ctl = w.take<unsigned long long>(kWalkCtlWords);
    unsigned long long* step1_buf = w.take<unsigned long long>(16);      // rg_sim_step_user: {action | result}
    uint32_t* park_t = w.take<uint32_t>(cache ? n : 1);
    const bool rp = n >= repack_min_users();      // small runs never repack: no second copy
    double* omega_alt = w.take<double>(rp ? ((K + 1) & ~static_cast<size_t>(1)) * n_pad : 1);
    unsigned long long* hist_alt = w.take<unsigned long long>(rp ? hc * n_pad : 1);
    uint32_t* lpv_alt = w.take<uint32_t>(rp && c.policy == RG_POLICY_LAST_VIEW_TABLE ? n : 1);
    uint32_t* uid_alt = w.take<uint32_t>(rp ? n : 1);
    if (d) {
        d->phantom_ps = phantom_ps; d->utime = utime; d->phantom_time = phantom_time;
        d->drift_list = drift_list; d->drift_sig = drift_sig; d->drift_cnt = drift_cnt;
        d->use_cache = cache ? 1u : 0u; d->cache_rec = cache_rec; d->cache_chunk = cache_chunk; d->cache_resc = cache_resc;
        d->beta32 = cache ? beta32 : nullptr; d->KB4 = static_cast<uint32_t>(KB4);
        d->f64_valid = f64_valid; d->exact_cnt_b = exact_cnt_b; d->cache_row = cache_row; d->cache_row_f = cache_row_f;
        d->park_list = park_list; d->park_t = park_t; d->sweep_only = 0;
        d->walk_ctl = walk_ctl; d->step1_buf = step1_buf;
        d->walk_hot = cache ? walk_hot : nullptr; d->walk_scp = walk_scp;
        d->lr_action = lr_action; d->lr_dirty = lr ? lr_dirty : nullptr; d->lr_list = lr_list; d->lr_cnt = lr_cnt; d->lr_part = lr_part;
        d->uid = uid; d->omega_alt = omega_alt; d->hist_alt = hist_alt;
        d->lpv_alt = lpv_alt; d->uid_alt = uid_alt;
        d->gamma32 = gamma32; d->mu32 = mu32; d->gamma32t = gamma32t; d->has_g32t = gamma32t_wanted(c, g) ? 1u : 0u; d->stats = stats; d->omega = omega; d->list = list;
        d->gamma_rm = gamma_rm; d->XKB = xkb;
        d->exact_rows = static_cast<uint32_t>(exact_rows); d->exact_base = 0;
        d->gammaT = gammaT; d->PT = static_cast<uint32_t>(PT); d->exact_ref = exact_ref; d->exact_sums = exact_sums; d->sc_scratch = sc_scratch; d->chunk_scratch = chunk_scratch;
        d->gsplit = gsplit; d->mu32s = mu32s; d->N1 = g.N1; d->N2 = g.N2; d->N3 = g.N3; d->RS = g.RS; d->TPB = g.TPB;
        d->f16 = g.F16 ? 1u : 0u; d->wide = g.F16 == 2 ? 1u : 0u;
        d->KH = g.KH; d->KS = g.KS; d->TP = g.TP; d->P_pad = g.P_pad; d->n_chunks = g.n_chunks;
        d->sc_chunks = g.sc_chunks; d->n_sc = g.n_sc; d->use_mfma = g.KH ? 1u : 0u;
        d->step_cnt = step_cnt; d->log_base = log_base; d->exact_list = exact_list;
        d->exact_cnt = exact_cnt; d->n_events = n_events; d->phantom = phantom;
        d->has_phantom = has_phantom; d->hist = hist;
        d->counters = counters; d->lpv = (c.policy == RG_POLICY_LAST_VIEW_TABLE) ? lpv : nullptr;
        d->n_pad = static_cast<uint32_t>(n_pad);
        d->OMS = static_cast<uint32_t>((K + 1) & ~static_cast<size_t>(1));
        d->hist_cap = static_cast<uint32_t>(hc);
    }
    return align_up(w.off, 256);
}

inline int validate(const rg_config* c, uint64_t n) {
    if (!c) return fail(RG_EINVAL, "config is NULL");
    if (c->num_products == 0 || c->num_products > RG_EV_INDEX_MASK)
        return fail(RG_EINVAL, "num_products %u out of range [1, 2^29)", c->num_products);
    if (c->K == 0 || c->K > 1024) return fail(RG_EINVAL, "K %u out of range [1, 1024]", c->K);
    if (sizeof(double) * (static_cast<size_t>(c->K) * 64 + 64 + 16 * c->K) > 64 * 1024)
        return fail(RG_EINVAL, "K %u exceeds the float64 draw kernel's LDS budget", c->K);
    if (n == 0 || n >= (1ull << 31)) return fail(RG_EINVAL, "n_users %llu out of range", (unsigned long long)n);
    if (c->policy > RG_POLICY_LOGREG_FROZEN) return fail(RG_EINVAL, "unknown policy %u", c->policy);
    if (c->time_mode > 1) return fail(RG_EINVAL, "unknown time_mode %u", c->time_mode);
    if (c->time_mode == 1 && !(c->time_sigma >= 0.0)) return fail(RG_EINVAL, "normal_time_sigma must be >= 0");
    for (int s = 0; s < 2; ++s)
        if (!(c->trans_cdf[s][0] >= 0.0 && c->trans_cdf[s][0] <= c->trans_cdf[s][1] &&
              c->trans_cdf[s][1] <= 1.0))
            return fail(RG_EINVAL, "transition cdf row %d is not monotone in [0,1]", s);
    return RG_OK;
}

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t prefix_in_mask(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

__device__ __forceinline__ double sigmoid64(double x) { return 1.0 / (1.0 + exp(-x)); }
// ff(): reco_env_v1.py:38-41
__device__ __forceinline__ double ff64(double x) {
    return sigmoid64(5.0 * sigmoid64(2.0 * sigmoid64(0.3 * x) - 2.0) - 6.0);
}

// No click below this uniform, whatever the action and omega: ff() is three nested sigmoids, sigmoid(0.3 x) in [0, 1] ->
// 2 s - 2 in [-2, 0] -> sigmoid in [0.119, 0.5] -> 5 s - 6 in [-5.40, -3.5] -> ctr in [0.004478, 0.0293123] (SURVEY.md
// appendix A.8), and numpy's choice([0, 1], p = [1 - ctr, ctr]) clicks iff u >= (1 - ctr) / ((1 - ctr) + ctr) >= 0.97068.
// 97 % of the bandit events need neither beta[a] nor omega: their click is 0 (the float64 path is taken when ctr itself
// is exported, `aux_pclick`).
constexpr double kNoClickBelow = 0.97;

// The click of a bandit event, click = [u >= 1 - ff(beta[a].omega + mu_b[a])] (reco_env_v1.py:104-116), decided in fp32
// wherever that is provably the float64 decision.  `b_row` = beta32[a] (KB4 floats, zero padded), om_at(k) =
// float(omega_k), mb = float(mu_b[a]).  Returns 1 / 0 = click / no click, -1 = undecided (the caller evaluates float64).
// Error budget (derivation): with e = 2^-24, x~ = fl32 dot of the rounded operands + fl32(mu_b),
//   |x~ - x| <= (K + 3) e (sum_k |beta_k omega_k| + |mu_b|)         (operand rounding 2e, K fma roundings, one add)
//   |ff'| <= 0.0285 * 5 * 0.25 * 2 * 0.25 * 0.3 = 5.4e-3             (range of the three nested sigmoids)
// so the dot contributes <= 7.5e-9 (ax + |mu_b|) at K = 20; the three v_exp_f32 / v_rcp_f32 sigmoids (1 ulp each) add
// <= 6e-8 to ctr, 1 - ctr and float(u) another 2^-24 + 2^-25: < 2e-7 in all.  The margin taken is 100x that:
// 2e-5 + 1e-6 (ax + |mu_b|); ~4e-5 of the acts land inside it.  Tested adversarially through
// rg_sim_debug_click_decisions (uniforms placed at 1 - ctr +- eps).
// `om_at(k)` = float(omega_k) (k < KMAX compile-time unrolled: LDS, memory or a register array), KMAX >= KB4 a multiple of 4.
template <int KMAX, class OmAt>
__device__ __forceinline__ int click_decide32(const float* b_row, OmAt om_at, uint32_t K, uint32_t KB4, float mb, double u) {
    const float4* b4 = reinterpret_cast<const float4*>(b_row);
    float x = 0.0f, ax = 0.0f;
#pragma unroll
    for (int k4 = 0; k4 < KMAX / 4; ++k4) {
        if (static_cast<uint32_t>(4 * k4) < KB4) {
            const float4 v = b4[k4];
            const float bb[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (static_cast<uint32_t>(4 * k4 + i) < K) {
                    const float wk = om_at(4 * k4 + i);
                    x = fmaf(bb[i], wk, x);
                    ax = fmaf(fabsf(bb[i]), fabsf(wk), ax);
                }
        }
    }
    auto sig32 = [](float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * z)); };
    const float ctr32 = sig32(5.0f * sig32(2.0f * sig32(0.3f * (x + mb)) - 2.0f) - 6.0f);
    const float margin = 2.0e-5f + 1.0e-6f * (ax + fabsf(mb));
    const float p0 = 1.0f - ctr32;
    const float uf = static_cast<float>(u);
    if (p0 < uf - margin) return 1;
    if (p0 > uf + margin) return 0;
    return -1;
}

// Box-Muller pair j of the K normals addressed by (user, t, purpose)
__device__ __forceinline__ void normal_pair(uint64_t seed, uint32_t user, uint32_t t, uint32_t j,
                                            uint32_t purpose, double* z0, double* z1) {
    const rg_u32x4 w = rg_draw(seed, user, t, j, purpose);
    const double u1 = rg_uniform(w.w[0], w.w[1]);
    const double u2 = rg_uniform(w.w[2], w.w[3]);
    const double r = sqrt(-2.0 * log(1.0 - u1));
    double s, c;
    sincos(RG_TWO_PI * u2, &s, &c);
    *z0 = r * c;
    *z1 = r * s;
}

// the uniform of a user's organic product draw at step t (word pair 0 of the event draw); the test hook
// rg_sim_debug_set_uniforms replaces it by a caller-chosen value per user index
__device__ __forceinline__ double organic_uniform(const DevSim& d, uint32_t uidx, uint32_t user, uint32_t t) {
    if (d.u_override) return d.u_override[uidx];
    const rg_u32x4 rw = rg_draw(d.seed, user, t, 0, RG_DRAW_EVENT);
    return rg_uniform(rw.w[0], rw.w[1]);
}

__device__ __forceinline__ uint32_t* list_ptr(const DevSim& d, uint32_t parity, uint32_t state) {
    return d.list + (static_cast<size_t>(parity) * 2 + state) * d.n_cap;
}

// ------------------------------------------------------------------------------------------
// k_reset_users
// ------------------------------------------------------------------------------------------
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_reset_users(DevSim d) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.step_cnt[0] = d.n_users;   // everyone starts organic (abstract.py:93)
        d.step_cnt[1] = 0;
        d.log_base[0] = 0;
    }
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        const uint32_t user = static_cast<uint32_t>(d.first_user + i);
        for (uint32_t j = 0; 2 * j < d.K; ++j) {
            double z0, z1;
            normal_pair(d.seed, user, 0u, j, RG_DRAW_RESET, &z0, &z1);
            d.omega[static_cast<size_t>(i) * d.OMS + 2 * j] = 0.0 + d.sigma0 * z0;
            if (2 * j + 1 < d.K) d.omega[static_cast<size_t>(i) * d.OMS + 2 * j + 1] = 0.0 + d.sigma0 * z1;
        }
        list_ptr(d, 0, RG_STATE_ORGANIC)[i] = i;
        d.uid[i] = i;
        d.n_events[i] = 0;
        d.has_phantom[i] = 0;
        if (d.time_mode) d.utime[i] = 0.0;
        if (d.lr_dirty) d.lr_dirty[i] = 1;
        if (d.hist_cap) d.hist[static_cast<size_t>(i) * d.hist_cap] = 0ull;
        if (d.use_cache) { d.f64_valid[i] = 0; d.cache_resc[i] = 0; }
    }
}
#endif

// fp32 copies of Gamma / mu_organic for the MFMA path: gamma32 [P_pad][KS] (columns >= K and rows
// >= P are zero), mu32 [P_pad] (-inf beyond P, so padded products get probability exactly 0).
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_make_fp32_tables(DevSim d) {
    const size_t n = static_cast<size_t>(d.P_pad) * d.KS;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t p = i / d.KS, k = i % d.KS;
        d.gamma32[i] = (p < d.P && k < d.K) ? static_cast<float>(d.gamma[p * d.K + k]) : 0.0f;
        if (i < d.P_pad) d.mu32[i] = i < d.P ? static_cast<float>(d.mu_o[i]) : -INFINITY;
    }
    if (d.has_g32t) {
        const size_t K2 = 2 * d.KH, nt = static_cast<size_t>(d.n_chunks) * K2 * 32;
        for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < nt;
             i += static_cast<size_t>(gridDim.x) * kBlock) {
            const size_t c = i / (K2 * 32), k = (i / 32) % K2, p = c * 32 + (i & 31);
            d.gamma32t[i] = (p < d.P && k < d.K) ? static_cast<float>(d.gamma[p * d.K + k]) : 0.0f;
        }
    }
}
#endif

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return static_cast<unsigned short>(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short hbits) {
    return __builtin_bit_cast(float, static_cast<unsigned>(hbits) << 16);
}
// x = h[0] + h[1] up to max(2^-22 |x|, 2^-25): two fp16 pieces, 11 significant bits each (the
// second piece turns subnormal below 2^-14: absolute granularity 2^-24)
__device__ __forceinline__ void f16_split2(float x, unsigned short* sp) {
    const _Float16 h1 = static_cast<_Float16>(x);
    const _Float16 h2 = static_cast<_Float16>(x - static_cast<float>(h1));
    sp[0] = __builtin_bit_cast(unsigned short, h1);
    sp[1] = __builtin_bit_cast(unsigned short, h2);
}
// x = s[0] + s[1] + s[2] up to ~2^-25 |x|: three bf16 pieces, 8 significant bits each
__device__ __forceinline__ void bf16_split3(float x, unsigned short* sp) {
    sp[0] = bf16_rne(x);
    float r = x - bf16_to_f32(sp[0]);
    sp[1] = bf16_rne(r);
    r -= bf16_to_f32(sp[1]);
    sp[2] = bf16_rne(r);
}

// gsplit[p] = [G1(K) | G2(K) | G3(K) | 0 ... 0 | 1 1 1] (bf16), the A operand rows of the split-bf16
// kernel, G = fl32(Gamma log2 e): the MFMA then yields logits in log2 units, and the three ones
// multiply the three bf16 pieces of -reference that sit in the user's B row.
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_make_split_table(DevSim d) {
    const size_t rs2 = d.RS / 2;
    const size_t n = static_cast<size_t>(d.P_pad) * rs2;
    const double log2e = 1.4426950408889634074;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t p = i / rs2, ke = i % rs2;
        unsigned short v = 0;
        if (d.f16) {
            if (p < d.P && ke < 3 * static_cast<size_t>(d.K)) {
                unsigned short sp[2];
                f16_split2(static_cast<float>(d.gamma[p * d.K + ke % d.K] * log2e), sp);
                v = sp[ke / d.K == 1 ? 1 : 0];                              // [G1 | G2 | G1]
            } else if (ke == 16u * d.N1 - 1) v = 0x3C00;                   // fp16(1.0): the reference column
        } else if (p < d.P && ke < 3 * static_cast<size_t>(d.K)) {
            unsigned short sp[3];
            bf16_split3(static_cast<float>(d.gamma[p * d.K + ke % d.K] * log2e), sp);
            v = sp[ke / d.K];
        } else if (ke >= 16u * d.N1 - 3 && ke < 16u * d.N1) v = 0x3F80;   // bf16(1.0)
        d.gsplit[i] = v;
        if (i < d.P_pad) d.mu32s[i] = i < d.P ? static_cast<float>(d.mu_o[i] * log2e) : -INFINITY;
    }
}
#endif

// float64 transpose of Gamma for the float64 draw kernel: lane-per-product reads coalesce
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_make_gammaT(DevSim d) {
    const size_t n = static_cast<size_t>(d.K) * d.PT;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t k = i / d.PT, p = i % d.PT;
        d.gammaT[i] = p < d.P ? d.gamma[p * d.K + k] : 0.0;
    }
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_make_beta32(DevSim d) {
    const size_t n = static_cast<size_t>(d.P) * d.KB4;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t p = i / d.KB4, k = i % d.KB4;
        d.beta32[i] = k < d.K ? static_cast<float>(d.beta[p * d.K + k]) : 0.0f;
    }
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_make_gamma_rm(DevSim d) {
    const uint32_t rs = 4 * d.XKB + 4;
    const size_t n = static_cast<size_t>(d.PT) * rs;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t p = i / rs, c = i % rs;
        double v = 0.0;
        if (c < d.K) v = p < d.P ? d.gamma[p * d.K + c] : 0.0;
        else if (c == 4 * d.XKB) v = p < d.P ? d.mu_o[p] : -INFINITY;
        d.gamma_rm[i] = v;
    }
}
#endif

// Table statistics for the logit error bound of the MFMA path (one block per statistic):
//   block k < 2KH : max_p |Gamma[p][k]|      block 2KH : max_p ||Gamma[p]||_2
//   block 2KH+1   : max_p |mu_o[p]|
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_table_stats(DevSim d) {
    __shared__ double red[kBlock];
    const uint32_t which = blockIdx.x;
    double m = 0.0;
    for (uint32_t p = threadIdx.x; p < d.P; p += kBlock) {
        double x;
        if (which < 2 * d.KH) x = which < d.K ? fabs(d.gamma[static_cast<size_t>(p) * d.K + which]) : 0.0;
        else if (which == 2 * d.KH || which >= 2 * d.KH + 2) {
            double q = 0.0;
            for (uint32_t k = 0; k < d.K; ++k) { const double g = d.gamma[static_cast<size_t>(p) * d.K + k]; q += g * g; }
            x = sqrt(q);
            // the grid of the joint bound: |mu_p| + ||Gamma_p||_2 r at r = (i + 1) / 4 (ahat_of)
            if (which >= 2 * d.KH + 2) x = fabs(d.mu_o[p]) + x * (static_cast<double>(which - (2 * d.KH + 2) + 1) * 0.25);
        } else x = fabs(d.mu_o[p]);
        m = fmax(m, x);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = kBlock / 2; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s2]);
        __syncthreads();
    }
    // round up: the bound must dominate the float64 value
    if (threadIdx.x == 0) d.stats[which] = static_cast<float>(red[0] * (1.0 + 1e-6));
}
#endif

typedef unsigned long long hent_t;
__device__ __forceinline__ hent_t* hist_row(const DevSim& d, uint32_t slot) { return d.hist + static_cast<size_t>(slot) * d.hist_cap; }
__device__ __forceinline__ uint32_t h_prod(hent_t e) { return static_cast<uint32_t>(e >> 32); }
__device__ __forceinline__ uint32_t h_cnt(hent_t e) { return static_cast<uint32_t>(e); }
constexpr int kHistRegs = 16;   // header + 15 products: one 128-byte line, held in registers

// the first line of a history row: 8 independent 16-byte loads (one latency instead of a dependent walk)
__device__ __forceinline__ void hist_load_line(const hent_t* row, hent_t e[kHistRegs]) {
#pragma unroll
    for (int i = 0; i < kHistRegs / 2; ++i) {
        const ulonglong2 x = reinterpret_cast<const ulonglong2*>(row)[i];
        e[2 * i] = x.x; e[2 * i + 1] = x.y;
    }
}

// count / sum, correctly rounded, from y = RN(1 / sum) (one true division per act instead of one per viewed product):
// q = RN(c y); r = c - sum q (exact in one fma); RN(q + r y) is the correctly rounded quotient whenever y is the
// correctly rounded reciprocal and the significand of `sum` is not all ones (Markstein 1990; Cornea, Harrison & Tang,
// "Scientific Computing on Itanium", Thm 8.5) — `sum` is an integer below 2^32 here, so it never is.  Checked
// exhaustively / on random operands against exact rational arithmetic in tests/test_host_logic.py.
__device__ __forceinline__ double div_by_reciprocal(double c, double sum, double y) {
    const double q = c * y;
    const double r = fma(-sum, q, c);
    return fma(r, y, q);
}

// !(acc / last <= u) exactly as float64 evaluates it, without the division where the answer is clear:
// acc < fl(u last)(1 - 2^-50) implies fl(acc / last) <= u, acc > fl(u last)(1 + 2^-50) implies fl(acc / last) > u
__device__ __forceinline__ bool cdf_exceeds(double acc, double last, double u) {
    const double tl = u * last;
    if (acc < tl * 0x1.ffffffffffff8p-1) return false;
    if (acc > tl * 0x1.0000000000004p+0) return true;
    return !(acc / last <= u);
}

// ------------------------------------------------------------------------------------------
// The policy's act on the device.  Returns the action; writes the propensity.
//   agent=None       abstract.py:209-221        uniform over P from the ENV stream
//   RandomAgent      random_agent.py:22-33      uniform over P from the agent's stream
//   OrganicUserEventCounter  organic_user_count.py:45-96 on the user's own view counts
// ------------------------------------------------------------------------------------------
// DENSE = false leaves out the O(P) forms of the OrganicUserEventCounter policy (explore flip, epsilon smoothing,
// reverse_pop: BASELINE configs use epsilon = 0) — ~40 % of this function's code, which the walk kernel would
// otherwise carry through its instruction cache on every step; the host picks the instantiation.
// HOOK = true (rg_sim_debug_ouc_acts only): the OrganicUserEventCounter draw takes `u1_hook` for its second uniform
// and *flag_hook tells whether the act was decided by the integer prefix walk (1) or by the float64 cdf walk (0).
template <bool DENSE = true, bool HOOK = false>
__device__ uint32_t policy_act(const DevSim& d, uint32_t slot, uint32_t user, uint32_t t,
                               double* ps_out, double u1_hook = 0.0, int* flag_hook = nullptr) {
    if (d.policy == RG_POLICY_LAST_VIEW_TABLE) {
        const uint32_t p = d.lpv[slot];
        *ps_out = d.pol_ps ? static_cast<double>(d.pol_ps[p]) : 1.0;
        return static_cast<uint32_t>(d.pol_table[p]);
    }
    if (d.policy == RG_POLICY_LOGREG_FROZEN) {
        // sklearn predict(): decision_function = X @ coef_.T + intercept_ with X the 1 x P CSR row of view
        // counts.  scipy's csr_matvecs adds count * coef_t[p][:] for the viewed products in ascending
        // order with a separate multiply and add (no FMA), then the intercept is added: reproduced
        // exactly, so ties and near-ties break like the reference's argmax (first maximum).
        const hent_t* hr = hist_row(d, slot);
        const uint32_t nd = h_cnt(hr[0]);
        uint32_t best = 0;
        double best_s = 0.0;
        for (uint32_t c = 0; c < d.lr_n; ++c) {
            double sc = 0.0;
            for (uint32_t i = 1; i <= nd; ++i)
                sc = __dadd_rn(sc, __dmul_rn(static_cast<double>(h_cnt(hr[i])), d.lr_coef_t[static_cast<size_t>(h_prod(hr[i])) * d.lr_n + c]));
            sc = __dadd_rn(sc, d.lr_intercept[c]);
            if (c == 0 || sc > best_s) { best = c; best_s = sc; }
        }
        *ps_out = 1.0;
        return static_cast<uint32_t>(d.lr_classes[best]);
    }
    const rg_u32x4 w = rg_draw(d.policy_seed, user, t, 0, RG_DRAW_POLICY);
    if (d.policy != RG_POLICY_ORGANIC_USER_COUNT) {
        *ps_out = 1.0 / static_cast<double>(d.P);
        return rg_bounded(w.w[0], w.w[1], d.P);
    }
    // --- OrganicUserEventCounterModel.act over the user's sorted (product, count) history ---
    const hent_t* hr = hist_row(d, slot);
    const double eps = d.ouc_epsilon;
    bool explore = false;
    if (d.ouc_exploit_explore && eps != 0.0) {            // (eps == 0: 0 / 1 <= u0 for every u0 — never explores)
        const double u0 = rg_uniform(w.w[0], w.w[1]);
        const double c0 = eps, c1 = eps + (1.0 - eps);
        explore = !(c0 / c1 <= u0);
    }
    const double u1 = HOOK ? u1_hook : rg_uniform(w.w[2], w.w[3]);
    if (HOOK) *flag_hook = 0;
    if (d.ouc_exploit_explore && !explore) {
        // p_i = count_i / sum(counts): zero entries add exactly 0.0 to the running cdf, so the
        // sequential float64 cumsum over all P products equals the one over the viewed ones.
        // sum(counts) = the views so far (integers: exact in float64 in any order) sits in the header.
        hent_t e[kHistRegs];
        hist_load_line(hr, e);
        const uint32_t nd = h_cnt(e[0]);
        const double sum = static_cast<double>(h_prod(e[0]));
        if (d.ouc_select_randomly) {
            // The float64 walk below compares RN(acc_i / last) with u1, where acc_i is the running sum of the
            // correctly rounded count_j / sum and last their total: it equals the exact ratio C_i / sum
            // (C_i = count_1 + .. + count_i, integers) up to (4 nd + 2) roundings — < 1e-12 relative for any history
            // that fits a row.  So wherever C_i and u1 * sum are further apart than 2^-36 relative the answer is decided
            // by integers — one add, one conversion and two compares per viewed product, a line of 16 entries at a
            // time, instead of a division and a float64 sum per product — and the walk in float64 is only taken by a
            // lane that lands inside that band (~1e-10 of the acts).
            const double T = u1 * sum;
            // C integer: C > T_hi <=> C > floor(T_hi), !(C < T_lo) <=> C >= ceil(T_lo) — the loop compares integers
            const uint32_t Thi = static_cast<uint32_t>(fmin(floor(T * (1.0 + 0x1p-36)), 4294967295.0));
            const uint32_t Tlo = static_cast<uint32_t>(fmin(ceil(T * (1.0 - 0x1p-36)), 4294967295.0));
            uint32_t C = 0, a_f = 0, c_f = 0;
            bool found = false, amb = false;
            hent_t f[kHistRegs];
#pragma unroll
            for (int i = 0; i < kHistRegs; ++i) f[i] = e[i];
            for (uint32_t base = 0; base <= nd && !found; base += kHistRegs) {
                if (base && RG_WALK_ABL(22)) { found = true; a_f = 0; c_f = 1; break; }   // timing experiment: first line only
                if (base) hist_load_line(hr + base, f);            // (rows are whole 16-entry lines)
#pragma unroll
                for (int i = 0; i < kHistRegs; ++i) {
                    const uint32_t idx = base + i;
                    if (idx >= 1 && idx <= nd && !found) {
                        C += h_cnt(f[i]);
                        if (C > Thi) { found = true; a_f = h_prod(f[i]); c_f = h_cnt(f[i]); }
                        else if (C >= Tlo) amb = true;
                    }
                }
            }
            if (found && !amb) {
                *ps_out = (1.0 - eps) * (static_cast<double>(c_f) / sum);
                if (HOOK) *flag_hook = 1;
                return a_f;
            }
        }
        if (nd < kHistRegs) {
            // the whole history is in registers: p_i once, then the cdf walk without touching memory again
            double pr[kHistRegs - 1];
            double last = 0.0;
            const double y = 1.0 / sum;
#pragma unroll
            for (int i = 1; i < kHistRegs; ++i) {
                pr[i - 1] = 0.0;
                if (static_cast<uint32_t>(i) <= nd) { pr[i - 1] = div_by_reciprocal(static_cast<double>(h_cnt(e[i])), sum, y); last += pr[i - 1]; }
            }
            if (d.ouc_select_randomly) {
                double acc = 0.0, pa = 0.0;
                uint32_t a = d.P - 1;     // searchsorted(..., 'right') on a cdf ending at 1.0
                bool found = false;
#pragma unroll
                for (int i = 1; i < kHistRegs; ++i)
                    if (static_cast<uint32_t>(i) <= nd && !found) {
                        acc += pr[i - 1];
                        if (cdf_exceeds(acc, last, u1)) { a = h_prod(e[i]); pa = pr[i - 1]; found = true; }
                    }
                *ps_out = (1.0 - eps) * pa;
                return a;
            }
            uint32_t best = 0; double bestp = -1.0;
#pragma unroll
            for (int i = 1; i < kHistRegs; ++i)
                if (static_cast<uint32_t>(i) <= nd && pr[i - 1] > bestp) { bestp = pr[i - 1]; best = h_prod(e[i]); }
            *ps_out = 1.0;
            return best;
        }
        if (d.ouc_select_randomly) {
            double last = 0.0;
            for (uint32_t i = 1; i <= nd; ++i) last += static_cast<double>(h_cnt(hr[i])) / sum;
            double acc = 0.0;
            uint32_t a = d.P - 1;
            double pa = 0.0;
            bool found = false;
            for (uint32_t i = 1; i <= nd && !found; ++i) {
                const hent_t x = hr[i];
                const double p = static_cast<double>(h_cnt(x)) / sum;
                acc += p;
                if (!(acc / last <= u1)) { a = h_prod(x); pa = p; found = true; }
            }
            *ps_out = (1.0 - eps) * pa;
            return a;
        }
        uint32_t best = 0; double bestp = -1.0;
        for (uint32_t i = 1; i <= nd; ++i) {
            const hent_t x = hr[i];
            const double p = static_cast<double>(h_cnt(x)) / sum;
            if (p > bestp) { bestp = p; best = h_prod(x); }
        }
        *ps_out = 1.0;
        return best;
    }
    if (!DENSE) { *ps_out = 1.0; return 0u; }       // (not reached: the host selects DENSE = true for these configurations)
    const uint32_t nd = h_cnt(hr[0]);
    // Dense cases (explore flip, epsilon smoothing, reverse_pop): every product has mass, the
    // float64 running sums are order-dependent, so walk all P products like numpy does.
    // O(P) per act; used by parity tests and small P only (BASELINE configs use epsilon = 0).
    auto count_of = [&](uint32_t p, uint32_t* cursor) -> double {
        // history is sorted by product id; cursor walks it once
        while (*cursor < nd && h_prod(hr[1 + *cursor]) < p) ++*cursor;
        return (*cursor < nd && h_prod(hr[1 + *cursor]) == p)
                   ? static_cast<double>(h_cnt(hr[1 + *cursor])) : 0.0;
    };
    auto feature = [&](double cnt) -> double {
        if (d.ouc_exploit_explore) return cnt == 0.0 ? 1.0 : 0.0;   // explore: unseen products
        return eps + cnt;
    };
    double sum = 0.0;
    uint32_t cur = 0;
    for (uint32_t p = 0; p < d.P; ++p) sum += feature(count_of(p, &cur));
    double sum2 = 0.0;
    if (!d.ouc_exploit_explore && d.ouc_reverse_pop) {
        cur = 0;
        for (uint32_t p = 0; p < d.P; ++p) sum2 += 1.0 - feature(count_of(p, &cur)) / sum;
    }
    auto prob = [&](double cnt) -> double {
        double pr = feature(cnt) / sum;
        if (!d.ouc_exploit_explore && d.ouc_reverse_pop) pr = (1.0 - pr) / sum2;
        return pr;
    };
    if (d.ouc_select_randomly) {
        double last = 0.0;
        cur = 0;
        for (uint32_t p = 0; p < d.P; ++p) last += prob(count_of(p, &cur));
        double acc = 0.0, pa = 0.0;
        uint32_t a = d.P - 1;
        bool found = false;
        cur = 0;
        for (uint32_t p = 0; p < d.P; ++p) {
            const double pr = prob(count_of(p, &cur));
            acc += pr;
            if (!found && !(acc / last <= u1)) { a = p; pa = pr; found = true; }
        }
        *ps_out = d.ouc_exploit_explore ? eps * pa : pa;
        return a;
    }
    uint32_t best = 0; double bestp = -1.0;
    cur = 0;
    for (uint32_t p = 0; p < d.P; ++p) {
        const double pr = prob(count_of(p, &cur));
        if (pr > bestp) { bestp = pr; best = p; }
    }
    *ps_out = 1.0;
    return best;
}

// ViewsFeaturesProvider.observe (agents/abstract.py:347-358): count one organic view, keeping the
// user's (product, count) history sorted by product id.
__device__ void history_add(const DevSim& d, uint32_t slot, uint32_t v) {
    if (d.lr_dirty) d.lr_dirty[d.uid[slot]] = 1;           // the frozen LogReg policy's cached act is stale now
    hent_t* hr = hist_row(d, slot);
    hent_t e[kHistRegs];
    hist_load_line(hr, e);
    const uint32_t nd = h_cnt(e[0]);
    const hent_t key = static_cast<hent_t>(v) << 32;
    if (nd < kHistRegs) {
        // header + every product in registers: position by comparison, the shifted tail written back
        // as whole 16-byte pairs (entries beyond nd + 1 of the line are don't-care)
        uint32_t pos = 1;                       // first entry with product >= v (nd + 1 if none)
        bool hit = false;
#pragma unroll
        for (int i = 1; i < kHistRegs; ++i)
            if (static_cast<uint32_t>(i) <= nd) {
                pos += e[i] < key ? 1u : 0u;
                hit = hit || h_prod(e[i]) == v;
            }
        if (hit) {
#pragma unroll
            for (int i = 1; i < kHistRegs; ++i)
                if (static_cast<uint32_t>(i) == pos) hr[i] = e[i] + 1ull;
            hr[0] = e[0] + (1ull << 32);
            return;
        }
        if (nd + 1 >= d.hist_cap) { atomicAdd(&d.counters[RG_CNT_HIST_OVERFLOW], 1ull); return; }
        hent_t f[kHistRegs + 2];                 // the row after the insertion
        f[0] = e[0] + (1ull << 32) + 1ull;
#pragma unroll
        for (int i = 1; i < kHistRegs + 1; ++i)
            f[i] = static_cast<uint32_t>(i) < pos ? e[i < kHistRegs ? i : 0] : (static_cast<uint32_t>(i) == pos ? (key | 1ull) : e[i - 1]);
        f[kHistRegs + 1] = 0ull;
        hr[0] = f[0];
#pragma unroll
        for (int i = 0; i < (kHistRegs + 2) / 2; ++i)
            if (static_cast<uint32_t>(2 * i + 1) >= pos && static_cast<uint32_t>(2 * i) <= nd + 1)
                reinterpret_cast<ulonglong2*>(hr)[i] = make_ulonglong2(i == 0 ? f[0] : f[2 * i], f[2 * i + 1]);
        return;
    }
    // longer histories: the position a line of 16 entries at a time (8 independent loads and 16 compares instead of a
    // dependent load per entry), the shift four entries at a time from the top
    if RG_WALK_ABL(22) return;          // timing experiment: histories stop growing at one line
    uint32_t pos = 1;                           // first entry with product >= v (nd + 1 if none)
    hent_t at = 0ull;                           // the entry there
    bool past = false;
    for (uint32_t base = 0; base <= nd && !past; base += kHistRegs) {
        hent_t f[kHistRegs];
        if (base) hist_load_line(hr + base, f);
        else {
#pragma unroll
            for (int i = 0; i < kHistRegs; ++i) f[i] = e[i];
        }
#pragma unroll
        for (int i = 0; i < kHistRegs; ++i) {
            const uint32_t idx = base + i;
            if (idx >= 1 && idx <= nd && !past) {
                if (f[i] < key) pos = idx + 1;
                else { past = true; at = f[i]; }
            }
        }
    }
    if (past && h_prod(at) == v) {
        hr[pos] = at + 1ull;
        hr[0] = e[0] + (1ull << 32);
        return;
    }
    if (nd + 1 >= d.hist_cap) { atomicAdd(&d.counters[RG_CNT_HIST_OVERFLOW], 1ull); return; }
    uint32_t j = nd + 1;                        // entries [pos, j) move up by one, highest first
    while (j > pos) {
        if (j >= pos + 4) {
            const hent_t a0 = hr[j - 4], a1 = hr[j - 3], a2 = hr[j - 2], a3 = hr[j - 1];
            hr[j - 3] = a0; hr[j - 2] = a1; hr[j - 1] = a2; hr[j] = a3;
            j -= 4;
        } else { hr[j] = hr[j - 1]; --j; }
    }
    hr[pos] = key | 1ull;
    hr[0] = e[0] + (1ull << 32) + 1ull;
}

// The same for a product BEHIND the first line of a longer history (k_walk2: the header and the 15 smallest products live in
// LDS, entries 16 .. nd of the row in memory are current and all larger than the line's last product).  Touches only entries
// >= 16 of the row; the header stays with the caller.  Returns 1 if the product is new (the caller's distinct count), 0 if
// its count was raised; `nd` = distinct products before the view (>= 15, nd + 1 < hist_cap checked by the caller).
__device__ __forceinline__ uint32_t history_tail_add(hent_t* hr, uint32_t nd, uint32_t v, uint32_t first = 16u) {
    const hent_t key = static_cast<hent_t>(v) << 32;
    uint32_t pos = first;                       // first entry >= `first` (the caller's line: 16, compact 32) with product >= v (nd + 1 if none)
    hent_t at = 0ull;
    bool past = false;
    for (uint32_t base = first; base <= nd && !past; base += kHistRegs) {
        hent_t f[kHistRegs];
        hist_load_line(hr + base, f);
#pragma unroll
        for (int i = 0; i < kHistRegs; ++i) {
            const uint32_t idx = base + i;
            if (idx <= nd && !past) {
                if (f[i] < key) pos = idx + 1;
                else { past = true; at = f[i]; }
            }
        }
    }
    if (past && h_prod(at) == v) { hr[pos] = at + 1ull; return 0u; }
    uint32_t j = nd + 1;                        // entries [pos, j) move up by one, highest first
    while (j > pos) {
        if (j >= pos + 4) {
            const hent_t a0 = hr[j - 4], a1 = hr[j - 3], a2 = hr[j - 2], a3 = hr[j - 1];
            hr[j - 3] = a0; hr[j - 2] = a1; hr[j - 1] = a2; hr[j] = a3;
            j -= 4;
        } else { hr[j] = hr[j - 1]; --j; }
    }
    hr[pos] = key | 1ull;
    return 1u;
}

// ------------------------------------------------------------------------------------------
// k_draw_exact — the organic product draw in float64, one wave per user.
//   l = Gamma omega + mu_o ; p = softmax(l) ; v = first index with cumsum(p)/cumsum(p)[-1] > u
// Pass 1: per-lane online (max, sum exp) over products lane, lane+64, ...; wave combine.
// Pass 2: recompute exp(l - max) in product order, wave-wide inclusive scan per 64 products,
//         first lane whose running prefix exceeds u * total wins.
// With from_list == 0 it serves every organic user of the step (correctness-first path);
// with from_list == 1 only the users the fp32 MFMA kernel could not certify.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double x) {
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ double wave_sum(double x) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

__device__ __forceinline__ void write_organic_row(const DevSim& d, uint32_t t, uint32_t pos, uint32_t slot,
                                                  uint32_t user, uint32_t v) {
    const uint64_t row = d.log_base[t] + pos;
    if (d.log && row < d.log_cap) {
        rg_event e;
        e.u = user; e.t = t; e.code = v; e.ps = __builtin_nanf("");
        d.log[row] = e;
        if (d.aux_time) d.aux_time[row] = d.utime[d.uid[slot]];     // the draw kernels run before k_advance moves the clock
    }
    if (d.lpv) d.lpv[slot] = v;   // BanditMFSquare.update_lpv, bandit_mf.py:60-65
}

// exp(x) in float64 for x <= ~700 (0 for x <= -750, incl. -inf): Cody-Waite reduction by ln 2 and a
// degree-13 Taylor polynomial on |r| <= 0.3466 (remainder 4e-18), ~20 instructions instead of the
// device library's ~55.  Accuracy ~1 ulp; the float64 path only has to agree with the oracle's
// libm exp to ~1e-15 relative (deviations at that level cannot move an index).
__device__ __forceinline__ double exp64(double x) {
    x = fmax(x, -750.0);
    const double kf = rint(x * 1.4426950408889634074);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;            // 1/13!
    p = fma(p, r, 2.08767569878681e-09);          // 1/12!
    p = fma(p, r, 2.505210838544172e-08);         // 1/11!
    p = fma(p, r, 2.755731922398589e-07);         // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);        // 1/9!
    p = fma(p, r, 2.48015873015873e-05);          // 1/8!
    p = fma(p, r, 1.984126984126984e-04);         // 1/7!
    p = fma(p, r, 1.388888888888889e-03);         // 1/6!
    p = fma(p, r, 8.333333333333333e-03);         // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);        // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, static_cast<int>(kf));
}

// Table variant for the kernel that spends its time in exp: exp(x) = 2^e * T[j] * exp(r) with
// n = rint(x * 32/ln 2) = 32 e + j and |r| <= ln 2 / 64, so a degree-6 polynomial is enough
// (remainder r^7/5040 < 4e-18) — ~15 float64 instructions instead of ~35.  T[j] = 2^(j/32),
// correctly rounded; `tab` is the block's LDS copy (32 doubles, one bank pair each: conflict-free).
static __device__ const double kExp2Tab32[32] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0};

__device__ __forceinline__ double exp64t(double x, const double* tab) {
    x = fmax(x, -750.0);                                   // e^-750 underflows to exactly 0 (also takes -inf)
    const double nf = rint(x * 0x1.71547652b82fep+5);      // 32 / ln 2
    double r = fma(nf, -0x1.62e42fe000000p-6, x);          // ln 2 / 32, high part (29 bits: nf * hi is exact)
    r = fma(nf, -0x1.f473de6af278fp-35, r);                // low part
    const int n = static_cast<int>(nf);
    double p = 1.3888888888888889e-03;                     // 1/6!
    p = fma(p, r, 8.3333333333333332e-03);                 // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);                 // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);                 // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(tab[n & 31] * p, n >> 5);
}

// inclusive scan of x over the 64 lanes of the wave
__device__ __forceinline__ double wave_scan(double x, int lane) {
    for (int o = 1; o < 64; o <<= 1) {
        const double y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    return x;
}

__device__ __forceinline__ double logit64(const DevSim& d, const double* om, uint32_t p) {
    // same association as the oracle / numpy: (sum_k Gamma[p][k] omega[k]) + mu[p], k ascending
    const double* g = d.gammaT + p;
    double l = 0.0;
#pragma unroll 4
    for (uint32_t k = 0; k < d.K; ++k) l += g[static_cast<size_t>(k) * d.PT] * om[k];
    return l + d.mu_o[p];
}

// four products per lane (p, p+64, p+128, p+192): four independent FMA chains keep 4x the loads
// in flight — the float64 kernel is latency-bound otherwise.  Products >= P give -inf.
__device__ __forceinline__ void logit64x4(const DevSim& d, const double* om, uint32_t p, double out[4]) {
    uint32_t idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { idx[u] = min(p + 64u * u, d.PT - 1); out[u] = 0.0; }
#pragma unroll 2
    for (uint32_t k = 0; k < d.K; ++k) {
        const double w = om[k];
        const double* g = d.gammaT + static_cast<size_t>(k) * d.PT;
#pragma unroll
        for (int u = 0; u < 4; ++u) out[u] += g[idx[u]] * w;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) out[u] = (p + 64u * u < d.P) ? out[u] + d.mu_o[p + 64u * u] : -INFINITY;
}

// The float64 draw is split in two kernels so that a step with FEW users to resolve (the long
// tail of the lock-step loop: ~1 400 of the ~1 800 steps of a 10 M-user run) is parallel over
// PRODUCTS instead of serial over them:
//   k_exact_sums  block = 16 users (4 per wave) x one slice of the 64-product chunks; the users
//                 share float64 Gamma^T tiles staged in LDS (unshared, the kernel was
//                 L2-bandwidth-bound: the table is P*K*8 bytes per user); writes exp-sums (or
//                 maxima, mode 0) per (user, chunk) to scratch.
//   k_exact_ref   (pure float64 mode only) reference = max logit per user.
//   k_exact_pick  wave per user: prefix over the chunk sums, u * total located by ballot, that
//                 chunk recomputed from the table, the row written.
constexpr int kUPW = 4;                      // users per wave
constexpr int kExactUsers = 4 * kUPW;        // users per block

#if RG_HAS(2)
__global__ void __launch_bounds__(kBlock) k_exact_sums(DevSim d, uint32_t t, int from_list, int mode, uint32_t S) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t n_chunks = d.PT / 64;
    const uint32_t n_cc = (n_chunks + 7) / 8;                  // coarse chunks of 8 x 64 products
    // LDS: Gamma^T tile [K][64] doubles, mu tile [64], omega [16 users][K]
    double* g_tile = reinterpret_cast<double*>(smem_raw);
    double* mu_tile = g_tile + static_cast<size_t>(d.K) * 64;
    double* om_all = mu_tile + 64;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const bool batched = from_list == 1 && !d.use_cache;
    const uint32_t base = batched ? d.exact_base : 0u;
    uint32_t n = from_list ? d.exact_cnt[t] : n_o;
    if (batched) n = min(n, base + d.exact_rows);
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const uint32_t n_groups = n > base ? (n - base + kExactUsers - 1) / kExactUsers : 0u;
    const uint32_t cps = ((n_cc + S - 1) / S) * 8;             // chunks per slice (whole coarse chunks)
    const uint32_t n_work = n_groups * S;

    for (uint32_t wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
        const uint32_t grp = wk / S, slice = wk % S;
        const uint32_t c0 = slice * cps, c1 = min(c0 + cps, n_chunks);
        if (c0 >= c1) continue;
        uint32_t w_idx[kUPW], srow[kUPW];
        bool act[kUPW];
        double M[kUPW], part[kUPW];
#pragma unroll
        for (int u = 0; u < kUPW; ++u) part[u] = 0.0;
        __syncthreads();      // previous work item's LDS is free
        double* om = om_all + static_cast<size_t>(wave * kUPW) * d.K;
#pragma unroll
        for (int u = 0; u < kUPW; ++u) {
            w_idx[u] = base + grp * kExactUsers + wave * kUPW + u;
            act[u] = w_idx[u] < n;
            const uint32_t pos = act[u] ? (from_list ? d.exact_list[w_idx[u]] : w_idx[u]) : 0u;
            const uint32_t slot = act[u] ? cur[pos] : 0u;
            srow[u] = w_idx[u] - base;                                         // row of this batch's scratch
            if (from_list && d.use_cache && act[u]) { w_idx[u] = d.uid[slot]; srow[u] = w_idx[u]; }   // per-user rows in this mode
            // any shift gives the same float64 decision up to 1e-16: a draw handed over by the
            // MFMA kernel reuses that kernel's reference, pure float64 mode uses k_exact_ref's
            M[u] = (mode == 1 && act[u]) ? static_cast<double>(d.exact_ref[w_idx[u]]) * 0.69314718055994530942 : 0.0;
            for (uint32_t k = lane; k < d.K; k += 64)
                om[k * kUPW + u] = act[u] ? d.omega[static_cast<size_t>(slot) * d.OMS + k] : 0.0;   // [k][user]
        }
        // The next chunk's Gamma^T tile is fetched into registers while the current one is being
        // used (the tile is K*64 doubles = K/4 per thread; staged through registers for K <= 32),
        // so the L2/HBM latency of the staging is off the per-chunk critical path.
        constexpr int kPF = 8;
        const bool prefetch = d.K * 64 <= kPF * kBlock;
        double pf[kPF];
        double pf_mu = 0.0;
        auto fetch = [&](uint32_t c) {
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                const uint32_t idx = threadIdx.x + i * kBlock;
                if (idx < d.K * 64) pf[i] = d.gammaT[static_cast<size_t>(idx >> 6) * d.PT + c * 64 + (idx & 63)];
            }
            if (threadIdx.x < 64) { const uint32_t p = c * 64 + threadIdx.x; pf_mu = p < d.P ? d.mu_o[p] : -INFINITY; }
        };
        if (prefetch) fetch(c0);
        for (uint32_t c = c0; c < c1; ++c) {
            __syncthreads();
            // stage Gamma^T[:, c*64 .. c*64+63] and mu (coalesced: 64 consecutive doubles per k)
            if (prefetch) {
#pragma unroll
                for (int i = 0; i < kPF; ++i) {
                    const uint32_t idx = threadIdx.x + i * kBlock;
                    if (idx < d.K * 64) g_tile[idx] = pf[i];
                }
                if (threadIdx.x < 64) mu_tile[threadIdx.x] = pf_mu;
            } else {
                for (uint32_t i = threadIdx.x; i < d.K * 64; i += kBlock) {
                    const uint32_t k = i >> 6, pp = i & 63;
                    g_tile[i] = d.gammaT[static_cast<size_t>(k) * d.PT + c * 64 + pp];
                }
                if (threadIdx.x < 64) {
                    const uint32_t p = c * 64 + threadIdx.x;
                    mu_tile[threadIdx.x] = p < d.P ? d.mu_o[p] : -INFINITY;
                }
            }
            __syncthreads();
            if (prefetch && c + 1 < c1) fetch(c + 1);
            // same association as the oracle / numpy: (sum_k Gamma[p][k] omega[k]) + mu[p]
            double l[kUPW];
#pragma unroll
            for (int u = 0; u < kUPW; ++u) l[u] = 0.0;
#pragma unroll 4
            for (uint32_t k = 0; k < d.K; ++k) {
                const double g = g_tile[k * 64 + lane];
                const double4 o4 = *reinterpret_cast<const double4*>(om + k * kUPW);   // 2 broadcast ds_read_b128
                l[0] += g * o4.x; l[1] += g * o4.y; l[2] += g * o4.z; l[3] += g * o4.w;
            }
            const double mu = mu_tile[lane];        // -inf for products >= P: exp() gives exactly 0
            // lane-local accumulation; one cross-lane reduction per coarse chunk (8 x 64 products) —
            // float64 cross-lane ops go through the LDS crossbar and dominated this kernel
#pragma unroll
            for (int u = 0; u < kUPW; ++u) {
                l[u] += mu;
                part[u] = mode == 0 ? fmax(part[u] == 0.0 && (c & 7) == 0 ? -INFINITY : part[u], l[u])
                                    : part[u] + exp64(l[u] - M[u]);
            }
            if ((c & 7) == 7 || c + 1 == c1) {
#pragma unroll
                for (int u = 0; u < kUPW; ++u) {
                    const double r = mode == 0 ? wave_max(part[u]) : wave_sum(part[u]);
                    if (lane == 0 && act[u]) d.exact_sums[static_cast<size_t>(srow[u]) * n_cc + (c >> 3)] = r;
                    part[u] = 0.0;
                }
            }
        }
    }
}
#endif


// ------------------------------------------------------------------------------------------
// k_exact_sums_m — the float64 chunk sums on the float64 MATRIX cores (v_mfma_f64_16x16x4_f64).
//
// Same job and output as k_exact_sums_u (exp-sum, or maximum in mode 0, of every 64-product chunk, per user).  There
// a lane owns a user and every product costs K dependent v_fma_f64 fed by scalar loads of the Gamma row plus ~18
// VALU instructions of exp — all on the vector ALU (44-47 % of its float64 peak at K = 20, 16 % at K = 64 where omega
// alone is 128 registers).  Here the dot products move to the matrix pipe, which runs beside the VALU:
//   D[product i][user j] += Gamma[i][4s..4s+3] . omega_j[4s..4s+3]        16 products x 16 users x 4 k per MFMA
// A = the Gamma rows of a 64-product chunk staged in LDS by the block (a straight copy of gamma_rm, mu in the last
// column), B = omega of 16 users (register resident for the work item), 4 (K <= 32) or 2 groups of 16 users per wave
// so that every A fragment read from LDS feeds 4 / 2 MFMAs; the VALU only adds mu, subtracts the reference and takes
// the exp of the 4 logits a lane gets per group and tile.  The matrix unit's accumulation order differs from the
// k-ascending chain (as the oracle's differs from OpenBLAS'): a 1e-16-level difference, decisions unchanged.
// C/D layout of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 * reg.
// ------------------------------------------------------------------------------------------
using f64x4 = __attribute__((ext_vector_type(4))) double;
__host__ __device__ constexpr int exact_m_groups(uint32_t kb) { return kb <= 8 ? 4 : 2; }
__host__ __device__ constexpr uint32_t exact_m_lds(uint32_t kb) { return (2u * 64u * (4u * kb + 4u) + 32u) * 8u; }

#if RG_HAS(2)
template <int KB>
__global__ void __launch_bounds__(kBlock) k_exact_sums_m(DevSim d, uint32_t t, int from_list, int mode, uint32_t S) {
    constexpr int G = exact_m_groups(KB);
    constexpr uint32_t UPW = 16 * G, UPB = (kBlock / 64) * UPW;      // users per wave / per block
    constexpr uint32_t RSd = 4 * KB + 4, TILE = 64 * RSd;            // doubles per staged chunk
    constexpr int NLD = (TILE / 2 + kBlock - 1) / kBlock;            // 16-byte pieces of a chunk per thread
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* tiles = reinterpret_cast<double*>(smem_raw);             // [2][TILE]
    double* exp_tab = tiles + 2 * TILE;
    if (threadIdx.x < 32) exp_tab[threadIdx.x] = kExp2Tab32[threadIdx.x];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int q = lane >> 4, jl = lane & 15;
    const uint32_t n_cc = d.PT / 64;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const bool batched = from_list == 1 && !d.use_cache;
    const uint32_t base = batched ? d.exact_base : 0u;
    uint32_t n = from_list == 2 ? t : (from_list ? d.exact_cnt[t] : n_o);
    if (batched) n = min(n, base + d.exact_rows);
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const uint32_t n_groups = n > base ? (n - base + UPB - 1) / UPB : 0u;
    const uint32_t ccps = (n_cc + S - 1) / S;                        // chunks per slice
    const uint32_t n_work = n_groups * S;
    for (uint32_t wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
        const uint32_t grp = wk / S, slice = wk % S;
        const uint32_t cc0 = slice * ccps, cc1 = min(cc0 + ccps, n_cc);
        if (cc0 >= cc1) continue;
        // ---- this lane's users: group g, column jl (the four lane quarters hold the same users, other rows) ----
        uint32_t row[G];
        bool act[G];
        double b[G][KB], M[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t w_idx = base + grp * UPB + wave * UPW + g * 16 + jl;
            act[g] = w_idx < n;
            uint32_t slot;
            if (from_list == 2) {
                slot = act[g] ? d.park_list[w_idx] : 0xFFFFFFFFu;
                act[g] = slot != 0xFFFFFFFFu;
                if (!act[g]) slot = 0u;
                w_idx = slot;
            } else {
                const uint32_t pos = act[g] ? (from_list ? d.exact_list[w_idx] : w_idx) : 0u;
                slot = act[g] ? cur[pos] : 0u;
                if (from_list && d.use_cache && act[g]) w_idx = d.uid[slot];
            }
            row[g] = w_idx - (batched ? base : 0u);
#pragma unroll
            for (int s2 = 0; s2 < KB; ++s2) {
                const uint32_t k = 4 * s2 + q;
                b[g][s2] = (act[g] && k < d.K) ? d.omega[static_cast<size_t>(slot) * d.OMS + k] : 0.0;
            }
            M[g] = (mode == 1 && act[g]) ? static_cast<double>(d.exact_ref[w_idx]) * 0.69314718055994530942 : 0.0;
        }
        // ---- chunks of the slice: the next one is fetched into registers while this one is used ----
        double2 pf[NLD];
        auto fetch = [&](uint32_t cc) {
            const double2* src = reinterpret_cast<const double2*>(d.gamma_rm + static_cast<size_t>(cc) * TILE);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const uint32_t idx = threadIdx.x + i * kBlock;
                if (idx < TILE / 2) pf[i] = src[idx];
            }
        };
        auto stash = [&](uint32_t buf) {
            double2* dst = reinterpret_cast<double2*>(tiles + buf * TILE);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const uint32_t idx = threadIdx.x + i * kBlock;
                if (idx < TILE / 2) dst[idx] = pf[i];
            }
        };
        __syncthreads();                       // the previous work item is done with both buffers
        fetch(cc0);
        stash(0);
        for (uint32_t cc = cc0; cc < cc1; ++cc) {
            __syncthreads();                   // chunk cc is in its buffer; the other one is free
            const bool more = cc + 1 < cc1;
            if (more) fetch(cc + 1);
            const double* A = tiles + ((cc - cc0) & 1u) * TILE;
            double sum[G];
#pragma unroll
            for (int g = 0; g < G; ++g) sum[g] = mode == 0 ? -INFINITY : 0.0;
#pragma unroll 1
            for (int tt = 0; tt < 4; ++tt) {
                f64x4 acc[G];
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = f64x4{0.0, 0.0, 0.0, 0.0};
                const double* arow = A + static_cast<size_t>(tt * 16 + jl) * RSd + q;
#pragma unroll
                for (int s2 = 0; s2 < KB; ++s2) {
                    const double a = arow[4 * s2];
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[g][s2], acc[g], 0, 0, 0);
                }
                double mu[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) mu[r] = A[static_cast<size_t>(tt * 16 + q + 4 * r) * RSd + 4 * KB];   // -inf for products >= P
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double l = acc[g][r] + mu[r];
                        sum[g] = mode == 0 ? fmax(sum[g], l) : sum[g] + exp64t(l - M[g], exp_tab);
                    }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                double x = sum[g];
                const double y = __shfl_xor(x, 16);
                x = mode == 0 ? fmax(x, y) : x + y;
                const double z = __shfl_xor(x, 32);
                x = mode == 0 ? fmax(x, z) : x + z;
                if (q == (g & 3) && act[g]) d.exact_sums[static_cast<size_t>(row[g]) * n_cc + cc] = x;
            }
            if (more) stash(((cc - cc0) & 1u) ^ 1u);
        }
    }
}
#endif

typedef const __attribute__((address_space(4))) double kdouble;   // constant address space: uniform loads become s_load

// ------------------------------------------------------------------------------------------
// k_exact_sums_h — the parked users' batch of k_walk on BOTH float64 pipes at once.
//
// At K <= 20 the matrix form (k_exact_sums_m: the MFMA pipe binds, the VALU is half idle) and the vector form
// (k_exact_sums_u: the VALU binds, the matrix pipe idles) take the same time.  Here a block takes the next group of 256
// listed users from a ticket counter and runs `mfma_of_8` groups of every 8 in the matrix form, the others in the
// vector form (a lane per user, Gamma rows through the scalar cache), so that the waves resident on a SIMD are a mix
// of both and the two pipes work side by side.  Exp-sums only (mode 1), whole table per user (no product slices).
// ------------------------------------------------------------------------------------------
// (compiled for four waves per SIMD — 127 registers instead of 102 + 32 — the batch takes the same time, as it does with 4 or 6
// of 8 groups in the matrix form: profiles/r4/ab_call7_exact_occupancy.jsonl)
#if RG_HAS(2)
template <int KB>
__global__ void __launch_bounds__(kBlock) k_exact_sums_h(DevSim d, uint32_t n, uint32_t mfma_of_8) {
    constexpr int G = exact_m_groups(KB);
    constexpr uint32_t UPW = 16 * G, UPB = (kBlock / 64) * UPW;      // 256 users per group at K <= 32
    constexpr uint32_t RSd = 4 * KB + 4, TILE = 64 * RSd;
    constexpr int NLD = (TILE / 2 + kBlock - 1) / kBlock;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* tiles = reinterpret_cast<double*>(smem_raw);             // [2][TILE]
    double* exp_tab = tiles + 2 * TILE;
    __shared__ uint32_t s_grp;
    if (threadIdx.x < 32) exp_tab[threadIdx.x] = kExp2Tab32[threadIdx.x];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int q = lane >> 4, jl = lane & 15;
    const uint32_t n_cc = d.PT / 64;
    if (d.q_count) n = static_cast<uint32_t>(*d.q_count);      // the list's length as the kernel before this one left it
    const uint32_t* plist = d.park_list + d.list_in;
    const uint32_t n_groups = (n + UPB - 1) / UPB;
    // Few groups per resident block (a rank's share of a strongly scaled run; the last round of blocks of any run): cut every
    // group's pass over the table into S product slices, so that the work items are >= 16 per launched block and the last
    // round of blocks is a slice, not a table, long (C3, 1.25 M users: 1 290 groups over 768 resident blocks = 2 rounds for 1.7)
    uint32_t S = 1;
    if (n_groups && n_groups < 16u * gridDim.x) S = min(8u, (16u * gridDim.x + n_groups - 1) / n_groups);
    S = min(S, n_cc);
    const uint32_t n_items = n_groups * S;
    for (;;) {
        __syncthreads();                       // s_grp and the LDS tiles of the previous group are free
        if (threadIdx.x == 0) s_grp = static_cast<uint32_t>(atomicAdd(d.q_ticket, 1ull));
        __syncthreads();
        if (s_grp >= n_items) break;
        const uint32_t grp = s_grp / S, slice = s_grp % S;
        const uint32_t cc_lo = slice * n_cc / S, cc_hi = (slice + 1u) * n_cc / S;      // this item's 64-product chunks
        if ((grp & 7u) < mfma_of_8) {
            // ================= matrix form (k_exact_sums_m's body, from_list == 2, one slice) =================
            uint32_t row[G];
            bool act[G];
            double b[G][KB], M[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t w_idx = grp * UPB + wave * UPW + g * 16 + jl;
                uint32_t slot = w_idx < n ? plist[w_idx] : 0xFFFFFFFFu;
                act[g] = slot != 0xFFFFFFFFu;
                if (!act[g]) slot = 0u;
                row[g] = slot;
#pragma unroll
                for (int s2 = 0; s2 < KB; ++s2) {
                    const uint32_t k = 4 * s2 + q;
                    b[g][s2] = (act[g] && k < d.K) ? d.omega[static_cast<size_t>(slot) * d.OMS + k] : 0.0;
                }
                M[g] = act[g] ? static_cast<double>(d.exact_ref[slot]) * 0.69314718055994530942 : 0.0;
            }
            double2 pf[NLD];
            auto fetch = [&](uint32_t cc) {
                const double2* src = reinterpret_cast<const double2*>(d.gamma_rm + static_cast<size_t>(cc) * TILE);
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const uint32_t idx = threadIdx.x + i * kBlock;
                    if (idx < TILE / 2) pf[i] = src[idx];
                }
            };
            auto stash = [&](uint32_t buf) {
                double2* dst = reinterpret_cast<double2*>(tiles + buf * TILE);
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const uint32_t idx = threadIdx.x + i * kBlock;
                    if (idx < TILE / 2) dst[idx] = pf[i];
                }
            };
            fetch(cc_lo);
            stash(cc_lo & 1u);
            for (uint32_t cc = cc_lo; cc < cc_hi; ++cc) {
                __syncthreads();
                const bool more = cc + 1 < cc_hi;
                if (more) fetch(cc + 1);
                const double* A = tiles + (cc & 1u) * TILE;
                double sum[G];
#pragma unroll
                for (int g = 0; g < G; ++g) sum[g] = 0.0;
#pragma unroll 1
                for (int tt = 0; tt < 4; ++tt) {
                    f64x4 acc[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g] = f64x4{0.0, 0.0, 0.0, 0.0};
                    const double* arow = A + static_cast<size_t>(tt * 16 + jl) * RSd + q;
#pragma unroll
                    for (int s2 = 0; s2 < KB; ++s2) {
                        const double a = arow[4 * s2];
#pragma unroll
                        for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[g][s2], acc[g], 0, 0, 0);
                    }
                    double mu[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) mu[r] = A[static_cast<size_t>(tt * 16 + q + 4 * r) * RSd + 4 * KB];
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sum[g] += exp64t(acc[g][r] + mu[r] - M[g], exp_tab);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    double x = sum[g];
                    x += __shfl_xor(x, 16);
                    x += __shfl_xor(x, 32);
                    if (q == (g & 3) && act[g]) d.exact_sums[static_cast<size_t>(row[g]) * n_cc + cc] = x;
                }
                if (more) stash((cc & 1u) ^ 1u);
            }
        } else {
            // ================= vector form (k_exact_sums_u's body): wave = 64 users of the group =================
            constexpr int UPL = UPB / (kBlock / 64) / 64;          // users per lane: 1 (256-user groups)
            uint32_t w_row[UPL];
            bool act[UPL];
            double om[UPL][4 * KB], M[UPL];
#pragma unroll
            for (int j = 0; j < UPL; ++j) {
                const uint32_t w_idx = grp * UPB + wave * 64 * UPL + j * 64 + lane;
                uint32_t slot = w_idx < n ? plist[w_idx] : 0xFFFFFFFFu;
                act[j] = slot != 0xFFFFFFFFu;
                if (!act[j]) slot = 0u;
                w_row[j] = slot;
#pragma unroll
                for (int k = 0; k < 4 * KB; ++k)
                    om[j][k] = (act[j] && static_cast<uint32_t>(k) < d.K) ? d.omega[static_cast<size_t>(slot) * d.OMS + k] : 0.0;
                M[j] = act[j] ? static_cast<double>(d.exact_ref[slot]) * 0.69314718055994530942 : 0.0;
            }
            for (uint32_t cc = cc_lo; cc < cc_hi; ++cc) {
                double acc[UPL];
#pragma unroll
                for (int j = 0; j < UPL; ++j) acc[j] = 0.0;
                const uint32_t p1 = cc * 64 + 64;
#pragma unroll 2
                for (uint32_t p = cc * 64; p < p1; ++p) {
                    kdouble* row = (kdouble*)(d.gamma_rm) + static_cast<size_t>(p) * RSd;
                    double l[UPL];
#pragma unroll
                    for (int j = 0; j < UPL; ++j) l[j] = 0.0;
#pragma unroll
                    for (int k = 0; k < 4 * KB; ++k) {
                        const double g = row[k];
#pragma unroll
                        for (int j = 0; j < UPL; ++j) l[j] += g * om[j][k];
                    }
#pragma unroll
                    for (int j = 0; j < UPL; ++j) acc[j] += exp64t(l[j] + row[4 * KB] - M[j], exp_tab);
                }
#pragma unroll
                for (int j = 0; j < UPL; ++j)
                    if (act[j]) d.exact_sums[static_cast<size_t>(w_row[j]) * n_cc + cc] = acc[j];
            }
        }
    }
}
#endif

#if RG_HAS(2)
exact_h_kernel_t exact_h_kernel_for(uint32_t kb) {
    switch (kb) {                               // K <= 32: 256-user groups in both forms
        case 1: return k_exact_sums_h<1>;   case 2: return k_exact_sums_h<2>;   case 3: return k_exact_sums_h<3>;
        case 4: return k_exact_sums_h<4>;   case 5: return k_exact_sums_h<5>;   case 6: return k_exact_sums_h<6>;
        case 8: return k_exact_sums_h<8>;
        default: return nullptr;
    }
}

exact_m_kernel_t exact_m_kernel_for(uint32_t kb) {
    switch (kb) {
        case 1: return k_exact_sums_m<1>;   case 2: return k_exact_sums_m<2>;   case 3: return k_exact_sums_m<3>;
        case 4: return k_exact_sums_m<4>;   case 5: return k_exact_sums_m<5>;   case 6: return k_exact_sums_m<6>;
        case 8: return k_exact_sums_m<8>;   case 12: return k_exact_sums_m<12>; case 16: return k_exact_sums_m<16>;
        default: return nullptr;
    }
}
#endif
// launch shape of k_exact_sums_m for `est` users: blocks of 64 / 128 / 256 users x S product slices
inline void launch_exact_m(exact_m_kernel_t km, const DevSim& d, uint32_t t, int from_list, int mode, uint64_t est, hipStream_t st) {
    const uint32_t upb = (kBlock / 64) * 16 * exact_m_groups(d.XKB);
    const uint64_t groups = (est + upb - 1) / upb;
    const uint32_t n_chunks = d.PT / 64;
    uint32_t S = static_cast<uint32_t>(4096 / (groups ? groups : 1));      // ~16 work items per CU when users are few
    if (S > n_chunks) S = n_chunks;
    if (S < 1) S = 1;
    uint64_t grid = groups * S;
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    const size_t smem = exact_m_lds(d.XKB);
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(km), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
    hipLaunchKernelGGL(km, dim3(static_cast<uint32_t>(grid)), dim3(kBlock), smem, st, d, t, from_list, mode, S);
}


// pure float64 mode: the reference of every user = its max logit (reco_env_v1.py:121)
#if RG_HAS(2)
__global__ void __launch_bounds__(kBlock) k_exact_ref(DevSim d, uint32_t t, uint32_t G) {
    const int lane = lane_id();
    const uint32_t n_cc = (d.PT / 64 + G - 1) / G;
    const uint32_t n = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t waves_total = gridDim.x * (kBlock / 64);
    for (uint32_t w = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w < n; w += waves_total) {
        double m = -INFINITY;
        for (uint32_t c = lane; c < n_cc; c += 64) m = fmax(m, d.exact_sums[static_cast<size_t>(w) * n_cc + c]);
        m = wave_max(m);
        if (lane == 0) d.exact_ref[w] = static_cast<float>(m * 1.4426950408889634074);
    }
}
#endif

// The float64 pick of one user, by a whole wave (every argument wave-uniform): prefix over the stored chunk sums ->
// the chunk that holds u * total -> its products walked in product order.  `om` = the user's omega in LDS.
// G = 64-product chunks per stored sum (8: tile kernel's coarse chunks, 1: user-per-lane kernel)
// om[k * om_stride]: the user's float64 omega (contiguous in k_exact_pick, one column of the wave's [K][64] LDS block
// in k_walk)
__device__ __forceinline__ uint32_t exact_pick_wave(const DevSim& d, const double* sums, const double* om, double M,
                                                    double u, uint32_t G, int lane, uint32_t om_stride = 1) {
    const uint32_t n_chunks = d.PT / 64;
    const uint32_t n_cc = (n_chunks + G - 1) / G;
    // One scan per block of 64 stored sums, kept in registers (up to 4 blocks = 256 sums = P <= 16 384 at G = 1; a
    // second pass over memory otherwise): the total and the search use the same partial sums — the same association.
    constexpr int RB = 4;
    double x[RB], incl[RB];
    const bool in_regs = n_cc <= 64u * RB;
    double total = 0.0;
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const uint32_t c = 64u * r + lane;
            x[r] = c < n_cc ? sums[c] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (64u * r < n_cc) { incl[r] = wave_scan(x[r], lane); total += __shfl(incl[r], 63); }
            else incl[r] = 0.0;
        }
    } else {
        for (uint32_t c0 = 0; c0 < n_cc; c0 += 64) {
            const uint32_t c = c0 + lane;
            total += __shfl(wave_scan(c < n_cc ? sums[c] : 0.0, lane), 63);
        }
    }
    // The reference normalises p = e / sum(e) before its cumsum and divides by cdf[-1];
    // dividing every term by the same positive constants moves the decision only at the
    // 1e-16 level, so the running sum of e is compared with u * total directly.
    const double target = u * total;
    // first coarse chunk whose inclusive running sum exceeds the target, and the sum before it
    uint32_t ccstar = n_cc - 1;
    double before = 0.0, run = 0.0;
    bool found = false;
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (64u * r < n_cc && !found) {
                const uint32_t c = 64u * r + lane;
                const unsigned long long hit = __ballot(c < n_cc && run + incl[r] > target);
                if (hit) {
                    const int L = __builtin_ctzll(hit);
                    ccstar = 64u * r + L;
                    before = run + __shfl(incl[r] - x[r], L);
                    found = true;
                } else run += __shfl(incl[r], 63);
            }
        }
    } else {
        for (uint32_t c0 = 0; c0 < n_cc && !found; c0 += 64) {
            const uint32_t c = c0 + lane;
            const double xv = c < n_cc ? sums[c] : 0.0;
            const double inc = wave_scan(xv, lane);
            const unsigned long long hit = __ballot(c < n_cc && run + inc > target);
            if (hit) {
                const int L = __builtin_ctzll(hit);
                ccstar = c0 + L;
                before = run + __shfl(inc - xv, L);
                found = true;
            } else run += __shfl(inc, 63);
        }
    }
    if (!found) before = run - sums[n_cc - 1];          // u * total rounded up to total
    __builtin_amdgcn_wave_barrier();
    // walk the G x 64 products of that coarse chunk in product order
    uint32_t v = min(ccstar * G * 64 + G * 64 - 1, d.P - 1);   // if rounding leaves no hit: its last product
    double acc = before;
    for (uint32_t i = 0; i < G; ++i) {
        const uint32_t p = (ccstar * G + i) * 64 + lane;
        if (ccstar * G + i >= n_chunks) break;
        const double* g = d.gammaT + p;                  // PT columns: always in range
        double lg = 0.0;
        // same association as the oracle (k ascending); the loads of eight k are issued together
        for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {
            double gv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = g[static_cast<size_t>(min(k0 + j, d.K - 1)) * d.PT];
#pragma unroll
            for (int j = 0; j < 8; ++j) if (k0 + j < d.K) lg += gv[j] * om[(k0 + j) * om_stride];
        }
        lg = p < d.P ? lg + d.mu_o[p] : -INFINITY;
        const double inc = wave_scan(exp64(lg - M), lane);
        const unsigned long long hit = __ballot(p < d.P && acc + inc > target);
        if (hit) { v = (ccstar * G + i) * 64 + static_cast<uint32_t>(__builtin_ctzll(hit)); break; }
        acc += __shfl(inc, 63);
    }
    return v;
}

#if RG_HAS(2)
__global__ void __launch_bounds__(kBlock) k_exact_pick(DevSim d, uint32_t t, int from_list, uint32_t G) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    double* om = reinterpret_cast<double*>(smem_raw) + static_cast<size_t>(wave) * d.K;
    const uint32_t n_cc = (d.PT / 64 + G - 1) / G;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const bool cached = from_list && d.use_cache;
    const bool batched = from_list == 1 && !d.use_cache;
    const uint32_t base = batched ? d.exact_base : 0u;
    const uint32_t n_all = from_list ? d.exact_cnt[t] : n_o;
    const uint32_t n_a = batched ? min(n_all, base + d.exact_rows) : n_all;   // draws whose sums the previous kernel took
    const uint32_t n = n_a + (cached ? d.exact_cnt_b[t] : 0u);          // + draws of users whose sums were there already
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const uint32_t waves_total = gridDim.x * (kBlock / 64);
    // more uncertified draws than the batches cover: reported, never silently dropped
    if (batched && d.exact_last && n_all > n_a && blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(&d.counters[RG_CNT_EXACT_OVERFLOW], static_cast<unsigned long long>(n_all - n_a));
    for (uint32_t w = base + blockIdx.x * (kBlock / 64) + wave; w < n; w += waves_total) {
        const uint32_t pos = from_list ? d.exact_list[w < n_a ? w : d.n_cap - 1u - (w - n_a)] : w;
        const uint32_t slot = cur[pos];
        const uint32_t uidx = d.uid[slot];
        const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
        const uint32_t row = cached ? uidx : w;
        const double M = static_cast<double>(d.exact_ref[row]) * 0.69314718055994530942;
        const double* sums = d.exact_sums + static_cast<size_t>(row - base) * n_cc;
        for (uint32_t k = lane; k < d.K; k += 64) om[k] = d.omega[static_cast<size_t>(slot) * d.OMS + k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t v = exact_pick_wave(d, sums, om, M, organic_uniform(d, uidx, user, t), G, lane);
        if (lane == 0) {
            write_organic_row(d, t, pos, slot, user, v);
            if (d.hist_cap) history_add(d, slot, v);
            if (cached && w < n_a) d.f64_valid[uidx] = 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (from_list == 1 && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&d.counters[RG_CNT_EXACT_DRAWS], static_cast<unsigned long long>(n > base ? n - base : 0u));
        atomicAdd(&d.counters[RG_CNT_EXACT_SWEEPS], static_cast<unsigned long long>(n_a > base ? n_a - base : 0u));
    }
}
exact_m_kernel_t exact_tile_kernel() { return k_exact_sums; }
exact_h_kernel_t exact_ref_kernel() { return k_exact_ref; }
exact_pick_kernel_t exact_pick_kernel() { return k_exact_pick; }
#endif

// ------------------------------------------------------------------------------------------
// k_draw_mfma — the organic product draw on the fp32 matrix cores, with a certified margin.
//
// One wave = 32 organic users (MFMA columns) x all P products in chunks of 32 (MFMA rows):
//     D[product i][user j] = mu[i] + sum_k Gamma32[i][k] * omega32[j][k]      (v_mfma_f32_32x32x2_f32)
// "products as rows" puts the 32 logits of one user into two lanes (16 registers each), so
// max / exp / sum over products is register-local; the two lanes of a user combine once per
// super-chunk.  Gamma32 tiles ([TP][KS] floats, KS == 2 mod 4 -> conflict-free ds_read_b64)
// and the mu tile are staged in LDS and shared by the block's 4 waves (128 users).
//
// Sampling v = first index with cumsum(p)/cumsum(p)[-1] > u needs the total before the prefix
// search.  Pass 1 (MFMA) keeps, per user, the sum of exp(l - ref) of each of <= 32 super-chunks
// (in LDS).  The search then picks the super-chunk from those sums in float64 and recomputes
// only that super-chunk (1/32 of P) on the vector ALU in product order to find the index.
//
// The result is only ACCEPTED if it is provably the float64 answer: with delta bounding the
// relative error of every fp32 prefix sum, v is certified iff
//     C~[v-1] (1+delta) < u S~ (1-delta)   and   u S~ (1+delta) < C~[v] (1-delta).
// Users that fail the test are appended to exact_list and resolved by k_draw_exact (float64).
// ------------------------------------------------------------------------------------------
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kRescaleGap = 57.0f;        // re-reference when a logit exceeds the reference by > ~40 nats
constexpr double kDeltaFixed = 3.0e-5;      // exp / summation / constant-rounding budget
constexpr double kDeltaPerRescale = 6.0e-6;
// split-bf16 kernel: references are integers (exact in bf16 pieces, exact exp2 of differences) and
// the logit leaves the MFMA already referenced and log2-scaled, so only the exp2 ulp (x2), the
// summation trees (~20 roundings) and the recompute's own fma/constant roundings remain: < 5e-6
constexpr double kDeltaFixedBf16 = 1.0e-5;

// Ahat: the bound on |mu_p + sum_{k <= j} Gamma_pk omega_k| over the products p and the partial sums j that the certificate's
// accumulation budget (K + 5) 2^-24 Ahat is proportional to.  Three bounds, the smallest taken: per coordinate
// (max|mu| + sum_k |omega_k| max_p |Gamma_pk| = mumax + absdot), Cauchy-Schwarz with the two maxima taken separately
// (max|mu| + max_p ||Gamma_p|| r, r = ||omega||_2), and Cauchy-Schwarz JOINTLY over the products, max_p (|mu_p| + ||Gamma_p|| r)
// — the product with the largest |mu| is not the one with the largest norm — read off a grid of r (k_table_stats; the
// bound is nondecreasing in r: the grid point at or above r is taken).  C3: 42 -> ~33, i.e. delta -19 %.
__device__ __forceinline__ float ahat_of(const DevSim& d, float mumax, float g2max, float absdot, float sq) {
    const float r = sqrtf(sq) * 1.000001f;
    float joint = mumax + g2max * r;
    const float gi = fmaxf(ceilf(r * 4.0f), 1.0f);
    if (gi <= static_cast<float>(kAhatGrid)) joint = fminf(joint, d.stats[2 * d.KH + 2 + static_cast<uint32_t>(gi) - 1u]);
    return fminf(mumax + absdot, joint) * 1.00001f;
}

// The certificate of every fp32 search, on CORRELATED errors.  The search's quantities are A (the prefix at the start of the
// draw's chunk), S (the total) — both running sums of the SAME sweep terms s_p = e_p (1 + eps_p), |eps_p| <= delta — and
// a, b (the recomputed fp32 prefixes inside the chunk, before / with product v; their terms carry their own errors <= delta).
// Product v is float64's answer iff  C[v-1] <= u S < C[v],  and with T = S - A (the sum of the sweep terms from the chunk's
// start on: every error of A is ALSO in S and cancels in the difference)
//     u S - C[v-1] = u T - (1 - u) A - a,      C[v] - u S = (1 - u) A + b - u T,
// whose computed values are off by at most  delta' (u T + (1 - u) A + a|b) + rho S:  delta' = delta / (1 - delta) on the true
// sums behind T, A, a|b, and rho = 2^-20 for the fp32 roundings of the two stored prefixes (<= 2^-21 each, relative to S and
// A).  At u S ~ A that is 2 delta A T / S where the independent form  C~(1 + delta) < u S~ (1 - delta)  pays 2 delta A — the
// band around a boundary shrinks by the mass BEHIND it, a third of the uncertified draws are left.
// Both tests are linear in u:  u den_lo > num_lo  and  u den_hi < num_hi  — the memo keeps num / den, rounded inwards.
struct CertLin { double num_lo, den_lo, num_hi, den_hi; bool valid; };
__device__ __forceinline__ CertLin cert_correlated(double S, double A, double a, double b, double delta) {
    const double dp = delta * (1.0 + 2.0 * delta);         // >= delta / (1 - delta) for delta <= 1/2
    const double rho = 0x1.0p-20 * 1.001 * S;              // (.001: second-order terms and the float64 roundings of these lines)
    const double T = S - A;                                // exact: both are fp32 values
    CertLin c;
    c.valid = T >= 0.0 && delta < 0.25;
    c.num_lo = (A + a) * (1.0 + dp) + rho;
    c.den_lo = T * (1.0 - dp) + A * (1.0 + dp);
    c.num_hi = (A + b) * (1.0 - dp) - rho;
    c.den_hi = T * (1.0 + dp) + A * (1.0 - dp);
    return c;
}

__device__ __forceinline__ float wave_scan_f32(float x, int lane) {
    for (int o = 1; o < 64; o <<= 1) {
        const float y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    return x;
}

__device__ __forceinline__ double readlane_f64(double x, int l) { return __shfl(x, l); }

// async global -> LDS copy of `bytes` contiguous bytes (gfx950 global_load_lds_dwordx4: the LDS
// destination is wave-uniform base + lane * 16), spread over the block's 4 waves
__device__ __forceinline__ void glds_copy(const char* src, char* dst_lds, uint32_t bytes, int wave, int lane) {
    for (uint32_t off = wave * 1024u; off < bytes; off += 4u * 1024u) {
        if (off + lane * 16u < bytes)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + off + lane * 16u),
                (__attribute__((address_space(3))) void*)(dst_lds + off), 16, 0, 0);
    }
}

// exchange a value between lane l and lane l ^ 32 (the two lanes that share a user)
__device__ __forceinline__ float swap32(float x) {
    // v_permlane32_swap_b32 (gfx950): with both operands = x, r[0] = {lo, lo}, r[1] = {hi, hi}
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

// ------------------------------------------------------------------------------------------
// Shared tail of the two MFMA draw kernels: given the per-chunk / per-super-chunk exp-sums a
// wave left in its scratch, pick super-chunk -> chunk -> product for each of its 32 users,
// certify the pick against float64 (see the header of k_draw_mfma) and emit the row or hand
// the user to k_draw_exact.  `om` = this lane's user's omega32 vector in LDS (2*KH floats).
// ------------------------------------------------------------------------------------------
// where the exp-sum of chunk c of user column j sits in a wave's chunk scratch:
// fp32 kernel: [chunk][32 users]; split-bf16 kernel: [tile of 4 chunks][32 users][4]
#define CHUNK_AT(c, j) (tiled4 ? (((c) >> 2) * 32 + (j)) * 4 + ((c) & 3) : (c) * 32 + (j))

// Where a lane finds / leaves its user's sums: record of super-chunk sc at rec[sc * rec_stride], the four chunk
// sums of product tile ti (16 bytes) at chunk[ti * tile_stride].  Per-wave scratch (users interleaved, one sweep's
// lifetime) or the per-user cache of the sigma_omega == 0 mode.
struct SumsView { float2* rec; uint32_t rec_stride; float* chunk; uint32_t tile_stride; };

__device__ __forceinline__ SumsView sums_view(const DevSim& d, float2* scr, float* scr_chunk, int j, bool active, uint32_t slot) {
    SumsView v;
    if (d.use_cache) {
        const size_t row = active ? d.uid[slot] : d.n_cap;          // inactive lanes: the dummy row
        v.rec = d.cache_rec + row * kMaxSC; v.rec_stride = 1;
        v.chunk = d.cache_chunk + row * d.n_chunks; v.tile_stride = 4;
    } else {
        v.rec = scr + j; v.rec_stride = 32;
        v.chunk = scr_chunk + 4 * j; v.tile_stride = 128;
    }
    return v;
}

template <int KH>
__device__ __forceinline__ void search_and_emit(const DevSim& d, uint32_t t, const float2* scr,
                                                const float* scr_chunk, const float* om_lds,
                                                float Ahat, int n_resc, bool active, uint32_t pos,
                                                uint32_t slot, int j, int h, bool tiled4, double delta_fixed,
                                                const SumsView* view = nullptr) {
        n_resc = max(n_resc, __shfl_xor(n_resc, 32));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // scratch: written by lanes < 32, read below

        // ---- search, part 1 (lane per user; lanes >= 32 mirror): total, target, super-chunk, chunk ----
        // All <= kMaxSC super-chunk records are fetched in one burst (they sit in L2, ~1 us away:
        // walking them with a data-dependent loop cost ~30 us per 128 users) and then live in registers.
        float2 rec[kMaxSC];
#pragma unroll
        for (uint32_t sc = 0; sc < kMaxSC; ++sc)
            rec[sc] = sc < d.n_sc ? (view ? view->rec[sc * view->rec_stride] : scr[sc * 32 + j])
                                  : make_float2(0.0f, -INFINITY);                          // unused: weight 0
        float Q = rec[0].y;                                    // common reference: the largest one
#pragma unroll
        for (uint32_t sc = 1; sc < kMaxSC; ++sc) Q = fmaxf(Q, rec[sc].y);
        double S = 0.0;
#pragma unroll
        for (uint32_t sc = 0; sc < kMaxSC; ++sc) {
            rec[sc].y = __builtin_amdgcn_exp2f(rec[sc].y - Q);
            rec[sc].x *= rec[sc].y;
            if (sc < d.n_sc) S += static_cast<double>(rec[sc].x);
        }
        const uint32_t user = static_cast<uint32_t>(d.first_user + d.uid[slot]);
        const double u_draw = organic_uniform(d, d.uid[slot], user, t);
        const double tau = u_draw * S;
        double pb = 0.0;
        uint32_t sc_star = d.n_sc - 1;
        float f_star = 1.0f;
        bool found_sc = false;
        {
            double run = 0.0;
#pragma unroll
            for (uint32_t sc = 0; sc < kMaxSC; ++sc) {
                const double Wd = static_cast<double>(rec[sc].x);
                if (sc < d.n_sc && !found_sc && run + Wd > tau) { found_sc = true; sc_star = sc; pb = run; f_star = rec[sc].y; }
                if (sc < d.n_sc && !found_sc) run += Wd;
            }
            if (!found_sc) f_star = 1.0f;
        }
        // chunk inside the super-chunk (its chunk sums share the super-chunk's reference)
        uint32_t c_star = 0;
        bool found_c = false;
        {
            const uint32_t c0 = sc_star * d.sc_chunks, c1 = min(c0 + d.sc_chunks, d.n_chunks);
            double run = pb;
            if (tiled4) {
                // [tile][user][4 chunks]: one 16-byte load per tile, four tiles in flight
                for (uint32_t cb = c0; cb < c1; cb += 16) {
                    float4 w4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        w4[i] = cb + 4 * i < c1 ? (view ? *reinterpret_cast<const float4*>(view->chunk + ((cb >> 2) + i) * view->tile_stride)
                                                        : *reinterpret_cast<const float4*>(scr_chunk + (((cb >> 2) + i) * 32 + j) * 4))
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float4 q4 = w4[i >> 2];
                        const float wv = (i & 3) == 0 ? q4.x : (i & 3) == 1 ? q4.y : (i & 3) == 2 ? q4.z : q4.w;
                        const double Wd = static_cast<double>(wv * f_star);
                        const uint32_t c = cb + i;
                        if (c < c1 && !found_c && run + Wd > tau) { found_c = true; c_star = c; pb = run; }
                        if (c < c1 && !found_c) run += Wd;
                    }
                }
            } else {
                for (uint32_t c = c0; c < c1; ++c) {
                    const double Wd = static_cast<double>(scr_chunk[CHUNK_AT(c, j)] * f_star);
                    if (!found_c && run + Wd > tau) { found_c = true; c_star = c; pb = run; }
                    if (!found_c) run += Wd;
                }
            }
        }
        found_c = found_c && found_sc;
        const double delta = static_cast<double>(d.K + 5) * 5.9604644775390625e-08 * static_cast<double>(Ahat) +
                             delta_fixed + kDeltaPerRescale * n_resc;

        // ---- search, part 2: recompute the 32 products of chunk c_star, 16 per lane, in registers ----
        uint32_t my_v = 0;
        bool my_ok = false;
        if (!(d.ablate & 1u)) {
            int vi; double Av, Bv;
            if constexpr (KH <= 16) {               // (gamma32t is always there at K <= 32: gamma32t_wanted)
                // the chunk from the chunk-major copy of Gamma: eight users per pass, eight lanes per user, four products per
                // lane — every load is a 128-byte run per k and user (the row-major gather below: 16 rows of 88 bytes per lane,
                // address-rate-bound: 29 % of the lock-step sweep's time at K = 20)
                constexpr int K2 = 2 * KH;
                const int lane_w = 32 * h + j, grp = lane_w >> 3, gl = lane_w & 7;
                const float remf = static_cast<float>(tau - pb);
                int r_idx = -1;
                float r_a = 0.0f, r_b = 0.0f;
#pragma unroll 1
                for (int ps = 0; ps < 4; ++ps) {
                    const int u = 8 * ps + grp;                        // the user this group works for (its h = 0 lane)
                    const uint32_t cs = static_cast<uint32_t>(__shfl(static_cast<int>(c_star), u));
                    const float Qs = __shfl(Q, u);
                    const float rems = __shfl(remf, u);
                    const float* ou = om_lds + (u - j) * K2;           // that user's omega32 in the wave's stage
                    const float4* gp = reinterpret_cast<const float4*>(d.gamma32t + (static_cast<size_t>(cs) * K2) * 32) + gl;
                    float4 l = *(reinterpret_cast<const float4*>(d.mu32 + cs * 32) + gl);
#pragma unroll
                    for (int kh = 0; kh < K2; kh += KH) {
                        float4 gk[KH];
#pragma unroll
                        for (int k = 0; k < KH; ++k) gk[k] = gp[(kh + k) * 8];
#pragma unroll
                        for (int k = 0; k < KH; ++k) {
                            const float wk = ou[kh + k];
                            l.x = fmaf(gk[k].x, wk, l.x); l.y = fmaf(gk[k].y, wk, l.y);
                            l.z = fmaf(gk[k].z, wk, l.z); l.w = fmaf(gk[k].w, wk, l.w);
                        }
                        asm volatile("" : "+v"(l.x), "+v"(l.y), "+v"(l.z), "+v"(l.w));
                    }
                    const float e0 = __builtin_amdgcn_exp2f(fmaf(l.x, kLog2e, -Qs)), e1 = __builtin_amdgcn_exp2f(fmaf(l.y, kLog2e, -Qs));
                    const float e2 = __builtin_amdgcn_exp2f(fmaf(l.z, kLog2e, -Qs)), e3 = __builtin_amdgcn_exp2f(fmaf(l.w, kLog2e, -Qs));
                    const float q0 = e0, q1 = q0 + e1, q2 = q1 + e2, q3 = q2 + e3;
                    float inc = q3;
#pragma unroll
                    for (int o2 = 1; o2 < 8; o2 <<= 1) {
                        const float y = __shfl_up(inc, o2, 8);
                        if (gl >= o2) inc += y;
                    }
                    float ex = __shfl_up(inc, 1, 8);
                    if (gl == 0) ex = 0.0f;
                    // the product in fp32 is only a proposal: the certificate below is taken from the two prefixes around it
                    const float x0 = ex + q0, x1 = ex + q1, x2 = ex + q2, x3 = ex + q3;
                    const int j0 = x0 > rems ? 0 : x1 > rems ? 1 : x2 > rems ? 2 : x3 > rems ? 3 : -1;
                    const unsigned long long hits = __ballot(j0 >= 0);
                    const uint32_t gmask = static_cast<uint32_t>(hits >> (8 * grp)) & 0xFFu;
                    const int win = 8 * grp + (gmask ? __builtin_ctz(gmask) : 7);          // the group's first hit (else its last lane)
                    const float f_idx = j0 >= 0 ? static_cast<float>(4 * gl + j0) : -1.0f;
                    const float f_a = j0 <= 0 ? (j0 == 0 ? ex : x3) : j0 == 1 ? x0 : j0 == 2 ? x1 : x2;   // (no hit: the chunk's sum)
                    const float f_b = j0 < 0 ? x3 : j0 == 0 ? x0 : j0 == 1 ? x1 : j0 == 2 ? x2 : x3;
                    const float g_idx = __shfl(f_idx, win), g_a = __shfl(f_a, win), g_b = __shfl(f_b, win);
                    // back to the user's own lanes (both halves): user u' is served in pass u' >> 3 by group u' & 7
                    const int from = 8 * (j & 7);
                    const float o_idx = __shfl(g_idx, from), o_a = __shfl(g_a, from), o_b = __shfl(g_b, from);
                    if ((j >> 3) == ps) { r_idx = static_cast<int>(o_idx); r_a = o_a; r_b = o_b; }
                }
                vi = r_idx;
                Av = pb + static_cast<double>(r_a);
                Bv = pb + static_cast<double>(r_b);
            } else {
            float om[2 * KH];
#pragma unroll
            for (int k = 0; k < 2 * KH; ++k) om[k] = om_lds[k];
            const uint32_t p_first = c_star * 32 + 16 * h;        // < P_pad by construction
            float pre[16];
            float runf = 0.0f;
#pragma unroll
            for (int i2 = 0; i2 < 8; ++i2) {
                // two rows = 2*KS floats, KS == 2 mod 4 -> a whole number of aligned float4
                const float4* rp = reinterpret_cast<const float4*>(d.gamma32 + static_cast<size_t>(p_first + 2 * i2) * d.KS);
                float rowpair[2 * (2 * KH + 2)];
                constexpr int KSc = 2 * KH + 2;
#pragma unroll
                for (int v4 = 0; v4 < KSc / 2; ++v4) {
                    const float4 x = rp[v4];
                    rowpair[4 * v4 + 0] = x.x; rowpair[4 * v4 + 1] = x.y; rowpair[4 * v4 + 2] = x.z; rowpair[4 * v4 + 3] = x.w;
                }
                const float2 mu2 = *reinterpret_cast<const float2*>(d.mu32 + p_first + 2 * i2);
                float l0 = mu2.x, l1 = mu2.y;
#pragma unroll
                for (int k = 0; k < 2 * KH; ++k) {
                    l0 = fmaf(rowpair[k], om[k], l0);
                    l1 = fmaf(rowpair[KSc + k], om[k], l1);
                }
                runf += __builtin_amdgcn_exp2f(fmaf(l0, kLog2e, -Q));
                pre[2 * i2] = runf;
                runf += __builtin_amdgcn_exp2f(fmaf(l1, kLog2e, -Q));
                pre[2 * i2 + 1] = runf;
            }
            // prefix of lane h=1 starts after lane h=0's 16 products
            const float t0 = swap32(runf);
            const double base = pb + (h ? static_cast<double>(t0) : 0.0);
            int idx = -1;
            double A = base, B = base;
#pragma unroll
            for (int i = 15; i >= 0; --i) {
                const double px = base + static_cast<double>(pre[i]);
                if (px > tau) { idx = i; B = px; A = i ? base + static_cast<double>(pre[i - 1]) : base; }
            }
            // the user's answer is lane h=0's hit if it has one, else lane h=1's
            const int idx_o = __shfl_xor(idx, 32);
            const double A_o = __shfl_xor(A, 32), B_o = __shfl_xor(B, 32);
            if (h == 0) { if (idx >= 0) { vi = idx; Av = A; Bv = B; } else { vi = idx_o >= 0 ? 16 + idx_o : -1; Av = A_o; Bv = B_o; } }
            else        { if (idx_o >= 0) { vi = idx_o; Av = A_o; Bv = B_o; } else { vi = idx >= 0 ? 16 + idx : -1; Av = A; Bv = B; } }
            }
            const uint32_t v = c_star * 32 + static_cast<uint32_t>(max(vi, 0));
            my_v = v;
            // (S, pb: float64 sums of the sweep's fp32 super-chunk / chunk sums, <= 2^-22 S off the exact sums of its terms)
            const CertLin ct = cert_correlated(S, pb, Av - pb, Bv - pb, delta);
            my_ok = found_c && vi >= 0 && v < d.P && ct.valid &&
                    (v == 0 || u_draw * ct.den_lo > ct.num_lo) &&
                    (v == d.P - 1 || u_draw * ct.den_hi < ct.num_hi);
        } else { my_v = static_cast<uint32_t>(S) % d.P; my_ok = true; }
        // ---- emit (lane per user) ----
        if (active && h == 0) {
            if (my_ok) {
                write_organic_row(d, t, pos, slot, user, my_v);
                if (d.hist_cap) history_add(d, slot, my_v);
            } else if (d.use_cache) {
                // float64 sums are per-user constants in this mode: taken once (front of the list), reused after (back)
                const uint32_t uidx = d.uid[slot];
                if (d.f64_valid[uidx]) d.exact_list[d.n_cap - 1u - atomicAdd(&d.exact_cnt_b[t], 1u)] = pos;
                else {
                    d.exact_list[atomicAdd(&d.exact_cnt[t], 1u)] = pos;
                    d.exact_ref[uidx] = Q;
                }
            } else {
                const uint32_t xi = atomicAdd(&d.exact_cnt[t], 1u);
                d.exact_list[xi] = pos;
                d.exact_ref[xi] = Q;
            }
        }
}

#if RG_HAS(3)
template <int KH>
__global__ void __launch_bounds__(kBlock, (KH <= 16 ? 2 : 1)) k_draw_mfma(DevSim d, uint32_t t) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t tile_f = d.TP * d.KS;                              // floats per Gamma tile
    float* g_buf = reinterpret_cast<float*>(smem_raw);                // [2][TP][KS]
    float* mu_buf = g_buf + 2 * tile_f;                               // [2][TP] (+ pad)
    float* om_stage = mu_buf + 2 * d.TP + 64;                         // [4 waves][32 users][2KH] omega32
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int j = lane & 31, h = lane >> 5;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_tiles = (n_o + 127) / 128;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    const uint32_t n_ptiles = (d.n_chunks * 32 + d.TP - 1) / d.TP;
    const uint32_t cpt = d.TP / 32;                                   // chunks per LDS tile
    // per-wave scratch: exp-sum of every chunk [n_chunks][32 users] and {sum, reference} of
    // every super-chunk [kMaxSC][32]
    const size_t wslot = static_cast<size_t>(blockIdx.x) * 4 + wave;
    float* scr_chunk = d.chunk_scratch + wslot * d.n_chunks * 32;
    float2* scr = d.sc_scratch + wslot * kMaxSC * 32;

    for (uint32_t tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
        const uint32_t pos = tb * 128 + wave * 32 + j;
        const bool active = pos < n_o;
        const uint32_t slot = active ? cur[pos] : 0u;
        // ---- B operand (omega32) and the logit error bound ----
        float b[KH];
        float absdot = 0.0f, sq = 0.0f;
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            const uint32_t k = h * KH + s;
            float w = 0.0f;
            if (active && k < d.K) w = static_cast<float>(d.omega[static_cast<size_t>(slot) * d.OMS + k]);
            b[s] = w;
            om_stage[(wave * 32 + j) * 2 * KH + k] = w;
            absdot = fmaf(fabsf(w), d.stats[k], absdot);
            sq = fmaf(w, w, sq);
        }
        absdot += swap32(absdot);
        sq += swap32(sq);
        const float Ahat = ahat_of(d, mumax, g2max, absdot, sq);

        // ---- pass 1: MFMA logits of chunk c overlap the exp-sum of chunk c-1 (software pipeline) ----
        float q = -1.0e30f;        // per-USER reference in log2 units, constant within a super-chunk
        float cqmax = -INFINITY;   // running max logit (log2 units) seen by this lane
        double s_sc = 0.0;         // running exp-sum of the current super-chunk (both lanes of the user)
        int n_resc = 0;
        f32x16 acc_p0, acc_p1;     // logits of the previous chunk pair, waiting for their exp-sums
        uint32_t ci_p = 0;         // index of its first chunk
        bool have_p = false;

        // exp-sum of one finished chunk: 16 logits per lane -> this user's chunk sum -> scratch
        auto softmax_chunk = [&](const f32x16& lg, uint32_t ci) {
            float cm = lg[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) cm = fmaxf(cm, lg[r]);
            cqmax = fmaxf(cqmax, cm * kLog2e);
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(fmaf(lg[r], kLog2e, -q));
#pragma unroll
            for (int w2 = 8; w2 > 0; w2 >>= 1)
#pragma unroll
                for (int r = 0; r < w2; ++r) e[r] += e[r + w2];
            const float wc = e[0] + swap32(e[0]);
            if (h == 0) scr_chunk[ci * 32 + j] = wc;
            s_sc += static_cast<double>(wc);
        };
        // end of a super-chunk: store {sum, reference}; re-reference if the max ran away
        auto flush_sc = [&](uint32_t ci) {
            if (h == 0) scr[(ci / d.sc_chunks) * 32 + j] = make_float2(static_cast<float>(s_sc), q);
            s_sc = 0.0;
            const float m2 = fmaxf(cqmax, swap32(cqmax));
            if (m2 > q + kRescaleGap) { q = m2; n_resc += 1; }
        };

        __syncthreads();           // every wave is done with both LDS buffers (previous user tile)
        glds_copy(reinterpret_cast<const char*>(d.gamma32), reinterpret_cast<char*>(g_buf), tile_f * 4, wave, lane);
        if (wave == 3) glds_copy(reinterpret_cast<const char*>(d.mu32), reinterpret_cast<char*>(mu_buf), d.TP * 4, 0, lane);
        for (uint32_t ti = 0; ti < n_ptiles; ++ti) {
            __syncthreads();       // (hipcc drains vmcnt before the barrier) tile ti landed; tile ti-1 is free
            if (ti + 1 < n_ptiles) {
                const uint32_t nb = (ti + 1) & 1;
                glds_copy(reinterpret_cast<const char*>(d.gamma32 + static_cast<size_t>(ti + 1) * tile_f),
                          reinterpret_cast<char*>(g_buf + nb * tile_f), tile_f * 4, wave, lane);
                if (wave == 3)
                    glds_copy(reinterpret_cast<const char*>(d.mu32 + static_cast<size_t>(ti + 1) * d.TP),
                              reinterpret_cast<char*>(mu_buf + nb * d.TP), d.TP * 4, 0, lane);
            }
            const float* g_tile = g_buf + (ti & 1) * tile_f;
            const float* mu_tile = mu_buf + (ti & 1) * d.TP;
            const uint32_t c_end = min(cpt, d.n_chunks - ti * cpt);     // even
            for (uint32_t c = 0; c < c_end; c += 2) {
                const uint32_t ci = ti * cpt + c;
                // operands of chunks c, c+1: accumulators start at mu, A rows from the LDS tile
                f32x16 acc0, acc1;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 m0 = *reinterpret_cast<const float4*>(mu_tile + c * 32 + 8 * qq + 4 * h);
                    const float4 m1 = *reinterpret_cast<const float4*>(mu_tile + c * 32 + 32 + 8 * qq + 4 * h);
                    acc0[4 * qq + 0] = m0.x; acc0[4 * qq + 1] = m0.y; acc0[4 * qq + 2] = m0.z; acc0[4 * qq + 3] = m0.w;
                    acc1[4 * qq + 0] = m1.x; acc1[4 * qq + 1] = m1.y; acc1[4 * qq + 2] = m1.z; acc1[4 * qq + 3] = m1.w;
                }
                const float* arow0 = g_tile + (c * 32 + j) * d.KS + h * KH;
                const float* arow1 = arow0 + 32 * d.KS;
                float2 a0[KH / 2], a1[KH / 2];
#pragma unroll
                for (int s = 0; s < KH / 2; ++s) {
                    a0[s] = *reinterpret_cast<const float2*>(arow0 + 2 * s);
                    a1[s] = *reinterpret_cast<const float2*>(arow1 + 2 * s);
                }
                // Two independent MFMA chains, interleaved: consecutive MFMAs never share an
                // accumulator, so neither the exp-sum VALU work of the previous pair (same wave)
                // nor another wave's instructions break a back-to-back dependent issue.
#pragma unroll
                for (int s = 0; s < KH / 2; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s].x, b[2 * s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s].x, b[2 * s], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s].y, b[2 * s + 1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s].y, b[2 * s + 1], acc1, 0, 0, 0);
                }
                if (have_p) {
                    softmax_chunk(acc_p0, ci_p);
                    softmax_chunk(acc_p1, ci_p + 1);
                    if ((ci_p + 2) % d.sc_chunks == 0) flush_sc(ci_p);
                } else {
                    // very first chunk pair of the user tile: its own max sets the reference
                    float cm = fmaxf(acc0[0], acc1[0]);
#pragma unroll
                    for (int r = 1; r < 16; ++r) cm = fmaxf(cm, fmaxf(acc0[r], acc1[r]));
                    cm *= kLog2e;
                    q = fmaxf(fmaxf(cm, swap32(cm)), -1.0e30f);
                }
                acc_p0 = acc0; acc_p1 = acc1; ci_p = ci; have_p = true;
            }
        }
        softmax_chunk(acc_p0, ci_p);           // drain the pipeline
        softmax_chunk(acc_p1, ci_p + 1);
        flush_sc(ci_p);
        search_and_emit<KH>(d, t, scr, scr_chunk, om_stage + (wave * 32 + j) * 2 * KH, Ahat, n_resc,
                            active, pos, slot, j, h, false, kDeltaFixed);
    }
}
#endif

// ------------------------------------------------------------------------------------------
// k_draw_bf16 — the same draw on the bf16 matrix cores with fp32-class accuracy.
//
// Measured on gfx950 (tools/ubench/mfma_coexec.hip, profiles/r1): the f32-input MFMA executes on
// the vector ALU's datapath — its time and the exp-sum's VALU time ADD — while bf16 MFMA runs on
// the separate matrix pipe and overlaps VALU work.  So the logit contraction is moved to bf16
// MFMA without giving up fp32 accuracy: every fp32 operand is split into three bf16 pieces
// (x = x1 + x2 + x3 up to 2^-25 |x|, 8 significant bits each) and the six cross terms with
// i + j <= 4 are accumulated in the MFMA's fp32 accumulator (the dropped ones are <= 2^-23 |x y|):
//     l = mu + G1 w1 + G2 w1 + G3 w1 + G1 w2 + G2 w2 + G1 w3
// as three MFMA groups that SHARE the A fragments: A row = [G1 | G2 | G3] (3K bf16, zero padded),
//     group 1: B = [w1 | w1 | w1]   (N1 k-steps of 16)
//     group 2: B = [w2 | w2 | 0 ]   (N2 k-steps; the zeros of B mask the A columns beyond 2K)
//     group 3: B = [w3 | 0  | 0 ]   (N3 k-steps)
// K = 20: 9 x v_mfma_f32_32x32x16_bf16 (~32 cycles each, overlapping the exp-sum) instead of
// 10 x v_mfma_f32_32x32x2_f32 (64 cycles each, serial with it).  Measured error vs float64:
// <= 4.2 x 2^-24 x sum|terms| (profiles/r1/ubench_bf16_split_accuracy.txt), inside the same
// (K+3) x 2^-24 budget of the certificate; everything after the logits is shared with
// k_draw_mfma (exp-sums, scratch, search, certificate, float64 fallback).
// ------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) short;

#if RG_HAS(3)
template <int KH, int N1, int N2, int N3>
__global__ void __launch_bounds__(kBlock, (N1 <= 6 ? 3 : 1)) k_draw_bf16(DevSim d, uint32_t t, uint32_t S) {
    // Register-lean form: ONE chunk (one accumulator) in flight per wave and no software pipeline,
    // so that 4 waves fit on a SIMD (<= 128 VGPRs) — the matrix pipe, the exp unit and the LDS of
    // a SIMD are kept busy by wave-level interleaving.  (A two-accumulator, ping-pong form of this
    // kernel needed 228+ VGPRs = 2 waves per SIMD and was latency-bound at the same speed as one.)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t tile_b = d.TPB * d.RS;                             // bytes per split tile
    char* g_buf = smem_raw;                                           // [2][TPB][RS]
    float* mu_buf = reinterpret_cast<float*>(g_buf + 2 * tile_b);     // [2][TPB] (+ pad)
    float* om_stage = mu_buf + 2 * d.TPB + 64;                        // [4 waves][32 users][2KH] omega32
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int j = lane & 31, h = lane >> 5;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_tiles = (n_o + 127) / 128;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    const uint32_t cpt = d.TPB / 32;                                  // chunks per LDS tile (multiple of 4)
    // With few user tiles (the long tail of the lock-step loop) the products are split into S
    // slices of whole super-chunks, one block per (user tile, slice), and the search runs in a
    // second kernel (k_draw_search): a step's latency is one slice, not the whole product sweep.
    const uint32_t scps = (d.n_sc + S - 1) / S;                       // super-chunks per slice
    const uint32_t n_work = n_tiles * S;
    float* omu = om_stage + (wave * 32 + j) * 2 * KH;                 // this lane's user's omega32

    for (uint32_t wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
        const uint32_t tb = wk / S, slice = wk % S;
        const uint32_t chunk_lo = min(slice * scps * d.sc_chunks, d.n_chunks);
        const uint32_t chunk_hi = min((slice + 1) * scps * d.sc_chunks, d.n_chunks);
        if (chunk_lo >= chunk_hi) continue;
        const uint32_t pt_lo = chunk_lo / cpt, pt_hi = (chunk_hi + cpt - 1) / cpt;   // product tiles
        // scratch of this (user tile, wave): by block when fused, by user tile when sliced
        const size_t wslot = (S == 1 ? static_cast<size_t>(blockIdx.x) : static_cast<size_t>(tb)) * 4 + wave;
        float* scr_chunk = d.chunk_scratch + wslot * d.n_chunks * 32;
        float2* scr = d.sc_scratch + wslot * kMaxSC * 32;
        const uint32_t pos = tb * 128 + wave * 32 + j;
        const bool active = pos < n_o;
        const uint32_t slot = active ? cur[pos] : 0u;
        __syncthreads();           // every wave is done with the LDS buffers and stage (previous work item)
        glds_copy(reinterpret_cast<const char*>(d.gsplit) + static_cast<size_t>(pt_lo) * tile_b, g_buf + (pt_lo & 1) * tile_b,
                  tile_b, wave, lane);
        if (wave == 3)
            glds_copy(reinterpret_cast<const char*>(d.mu32s + static_cast<size_t>(pt_lo) * d.TPB),
                      reinterpret_cast<char*>(mu_buf + (pt_lo & 1) * d.TPB), d.TPB * 4, 0, lane);
        // ---- omega32 of the user -> LDS stage (also the logit error bound) ----
        float absdot = 0.0f, sq = 0.0f;
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            const uint32_t k = h * KH + s;
            float w = 0.0f;
            if (active && k < d.K) w = static_cast<float>(d.omega[static_cast<size_t>(slot) * d.OMS + k]);
            omu[k] = w;
            absdot = fmaf(fabsf(w), d.stats[k], absdot);
            sq = fmaf(w, w, sq);
        }
        absdot += swap32(absdot);
        sq += swap32(sq);
        const float Ahat = ahat_of(d, mumax, g2max, absdot, sq);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- B fragments: lane (j, h) holds elements ke = 16 s + 8 h + e of its user's B rows ----
        bf16x8 B1[N1], B2[N2], B3[N3];
        {
            const uint32_t K = d.K;
#pragma unroll
            for (int s = 0; s < N1; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t ke = 16 * s + 8 * h + e;
                    unsigned short sp[3] = {0, 0, 0};
                    if (ke < 3 * K) bf16_split3(omu[ke % K], sp);
                    B1[s][e] = static_cast<short>(sp[0]);
                    if (s < N2) B2[s < N2 ? s : 0][e] = static_cast<short>(ke < 2 * K ? sp[1] : 0);
                    if (s < N3) B3[s < N3 ? s : 0][e] = static_cast<short>(ke < K ? sp[2] : 0);
                }
        }
        // the reference rides in the MFMA: columns 16 N1 - 3 .. 16 N1 - 1 of A are 1, the matching
        // B elements (lanes h == 1, elements 5..7 of the last k-step) hold the 3 bf16 pieces of -q
        float q = 0.0f;            // per-USER reference in log2 units (an integer), constant within a super-chunk
        auto set_reference = [&](float qn) {
            q = qn;
            unsigned short sp[3];
            bf16_split3(-qn, sp);
            if (h == 1) {
                B1[N1 - 1][5] = static_cast<short>(sp[0]);
                B1[N1 - 1][6] = static_cast<short>(sp[1]);
                B1[N1 - 1][7] = static_cast<short>(sp[2]);
            }
        };

        double s_sc = 0.0;         // running exp-sum of the current super-chunk
        float wcmax = 0.0f;        // largest chunk sum of the current super-chunk
        int n_resc = 0;

        // logits (log2 units, reference already subtracted) of one 32-product chunk
        auto mfma_chunk = [&](const char* g_tile, const float* mu_tile, uint32_t c, f32x16& acc) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float4 m0 = *reinterpret_cast<const float4*>(mu_tile + c * 32 + 8 * qq + 4 * h);
                acc[4 * qq + 0] = m0.x; acc[4 * qq + 1] = m0.y; acc[4 * qq + 2] = m0.z; acc[4 * qq + 3] = m0.w;
            }
            const char* arow = g_tile + (c * 32 + j) * d.RS + 16 * h;
            bf16x8 A[N1];
#pragma unroll
            for (int s = 0; s < N1; ++s) A[s] = *reinterpret_cast<const bf16x8*>(arow + 32 * s);
#pragma unroll
            for (int s = 0; s < N1; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s], B1[s], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < N2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s], B2[s], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < N3; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s], B3[s], acc, 0, 0, 0);
        };
        // exp-sum of one chunk: 16 exp2 + a (packed) tree sum per lane; returns this lane's partial
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        auto expsum_chunk = [&](f32x16& y) -> float {
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = __builtin_amdgcn_exp2f(y[r]);
            f32x2 p0 = {y[0], y[1]}, p1 = {y[2], y[3]}, p2 = {y[4], y[5]}, p3 = {y[6], y[7]};
            const f32x2 p4 = {y[8], y[9]}, p5 = {y[10], y[11]}, p6 = {y[12], y[13]}, p7 = {y[14], y[15]};
            p0 += p4; p1 += p5; p2 += p6; p3 += p7;        // v_pk_add_f32
            p0 += p2; p1 += p3;
            p0 += p1;
            return p0[0] + p0[1];
        };
        // end of a super-chunk: store {sum, reference}; re-reference if a sum grew past 2^48
        auto flush_sc = [&](uint32_t sc) {
            scr[sc * 32 + j] = make_float2(static_cast<float>(s_sc), q);
            s_sc = 0.0;
            if (wcmax > 2.8e14f) {     // some logit is >= ~43 above the reference (log2 units)
                set_reference(q + floorf(__builtin_amdgcn_logf(wcmax)));   // v_log_f32 = log2
                n_resc += 1;
            }
            wcmax = 0.0f;
        };

        f32x16 y;
        uint32_t sc_cur = chunk_lo / d.sc_chunks;              // super-chunk being accumulated
        uint32_t sc_left = d.sc_chunks / 4;                    // tiles left in it (TPB = 128: 4 chunks per tile)
        for (uint32_t ti = pt_lo; ti < pt_hi; ++ti) {
            __syncthreads();       // tile ti landed (hipcc drains vmcnt before the barrier); tile ti-1 is free
            if (ti + 1 < pt_hi) {
                const uint32_t nb = (ti + 1) & 1;
                glds_copy(reinterpret_cast<const char*>(d.gsplit) + static_cast<size_t>(ti + 1) * tile_b,
                          g_buf + nb * tile_b, tile_b, wave, lane);
                if (wave == 3)
                    glds_copy(reinterpret_cast<const char*>(d.mu32s + static_cast<size_t>(ti + 1) * d.TPB),
                              reinterpret_cast<char*>(mu_buf + nb * d.TPB), d.TPB * 4, 0, lane);
            }
            const char* g_tile = g_buf + (ti & 1) * tile_b;
            const float* mu_tile = mu_buf + (ti & 1) * d.TPB;
            if (ti == pt_lo) {
                // first chunk of the work item with reference 0: its max (an integer after ceil, so
                // exact in bf16 pieces and in exp2 differences) becomes the reference
                mfma_chunk(g_tile, mu_tile, 0, y);
                float cm = y[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) cm = fmaxf(cm, y[r]);
                set_reference(fmaxf(ceilf(fmaxf(cm, swap32(cm))), -1.0e30f));
            }
            // the four chunks of the tile; their sums leave as one 16-byte store per user
            float4 w4;
            mfma_chunk(g_tile, mu_tile, 0, y); w4.x = expsum_chunk(y);
            mfma_chunk(g_tile, mu_tile, 1, y); w4.y = expsum_chunk(y);
            mfma_chunk(g_tile, mu_tile, 2, y); w4.z = expsum_chunk(y);
            mfma_chunk(g_tile, mu_tile, 3, y); w4.w = expsum_chunk(y);
            w4.x += swap32(w4.x); w4.y += swap32(w4.y); w4.z += swap32(w4.z); w4.w += swap32(w4.w);
            // scratch layout [tile][user][4 chunks]; both lanes of the user hold the same sums: no branch
            *reinterpret_cast<float4*>(scr_chunk + (static_cast<size_t>(ti) * 32 + j) * 4) = w4;
            wcmax = fmaxf(fmaxf(wcmax, fmaxf(w4.x, w4.y)), fmaxf(w4.z, w4.w));
            s_sc += static_cast<double>((w4.x + w4.y) + (w4.z + w4.w));
            if (--sc_left == 0) { flush_sc(sc_cur); ++sc_cur; sc_left = d.sc_chunks / 4; }
        }
        if (sc_left != d.sc_chunks / 4) flush_sc(sc_cur);
        if (S == 1) search_and_emit<KH>(d, t, scr, scr_chunk, omu, Ahat, n_resc, active, pos, slot, j, h, true, kDeltaFixedBf16);
    }
}
#endif

// ------------------------------------------------------------------------------------------
// k_draw_bf16p — the same computation as k_draw_bf16 with the instruction stream arranged for
// the matrix pipe (tools/ubench/chunk_il.hip, chunk_loop.hip):
//   * a wave's back-to-back MFMAs keep the SIMD's VALU issue port, so other waves' exp work does
//     NOT fill in behind them, and a dependent accumulator chain leaves ~25 unusable idle cycles
//     per MFMA: the one-accumulator loop above costs MFMA time + VALU time + LDS latency;
//   * with two independent chains (a PAIR of chunks) interleaved in one wave and the exp-sum of
//     the PREVIOUS pair plus the LDS operand loads of the NEXT pair placed in the issue slots
//     between the MFMAs (order pinned with sched_barrier), everything but the MFMA stream hides.
// Per pair: 2 (N1+N2+N3) MFMAs, 32 exps + 2 trees of the previous pair, 2 N1 + 8 ds_read_b128
// of the next pair.  One barrier per product tile, placed between its two pairs: at that point
// every wave holds the tile's operands in registers (so the buffer is refilled with tile + 2)
// and the tile after it has landed (so the second pair's stream can fetch from it).
// Needs ~200 VGPRs = 2 waves per SIMD; one such wave already paces the matrix pipe.
// ------------------------------------------------------------------------------------------
#define RG_PIN() __builtin_amdgcn_sched_barrier(0)

// Certificate budget of the two-way fp16 split on top of the accumulation budget (K+5) 2^-24 Ahat:
// x = h1 + h2 + e with |e| <= max(2^-22 |x|, 2^-25), and the h2 h2 cross term is dropped, so a
// logit is off by <= 3 x 2^-22 sum|g_k w_k| + 2^-25 sum_k (|g_k| + |w_k|) in log2 units, i.e. relative
// error of its exp <= 12 x 2^-24 Ahat + 2^-25 (sum_k max_p |Gamma_pk| + ln 2 sum_k |omega_k|).
__device__ __forceinline__ double f16_extra_delta(const DevSim& d, float Ahat, float absw) {
    float gsum = 0.0f;
    for (uint32_t k = 0; k < d.K; ++k) gsum += d.stats[k];
    return 12.0 * 5.9604644775390625e-08 * static_cast<double>(Ahat) +
           2.98023223876953125e-08 * (static_cast<double>(gsum) + 0.6931471805599453 * static_cast<double>(absw));
}

// Tile DMA the compiler does not see.  hipcc puts s_waitcnt vmcnt(0) in front of the first ds_read
// that follows a global/buffer load to LDS (the DMA may alias the read), which turns the tile
// prefetch into a synchronous load.  The pipelined kernel only reads a tile after the barrier
// that publishes it, so it issues the DMA opaquely (buffer_load_dwordx4 ... lds: LDS address =
// M0 + lane * 16, memory address = resource base + scalar offset + lane offset) and waits for it
// itself (RG_DMA_WAIT) right before that barrier.  The LDS reads stay ordinary compiler-visible loads.
typedef int rg_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_to_lds_b128(rg_v4i rsrc, uint32_t lds_addr, uint32_t lane_off, uint32_t s_off) {
    uint32_t keep_m0;          // M0 is the compiler's: borrowed and put back
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "s"(lds_addr), "v"(lane_off), "s"(rsrc), "s"(s_off) : "memory");
}
// raw buffer resource over [p, p + 2 GiB): base, stride 0, num_records, gfx9 raw-buffer flags
__device__ __forceinline__ rg_v4i raw_buffer_rsrc(const void* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    rg_v4i r;
    r[0] = static_cast<int>(static_cast<uint32_t>(a));
    r[1] = static_cast<int>(static_cast<uint32_t>(a >> 32) & 0xffffu);
    r[2] = 0x7fffffff;
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* generic_ptr) {
    return static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const char*)generic_ptr));
}
#define RG_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// Tile barrier without the fence of __syncthreads (which drains every outstanding store and DMA):
// waits until at most N of this wave's vector-memory operations are still in flight (they complete
// in issue order) and its LDS reads have returned, then rendezvous.
#define RG_TILE_BARRIER(N)                                                     \
    do {                                                                       \
        asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_s_barrier();                                          \
        asm volatile("" ::: "memory");                                         \
    } while (0)

// timing experiments of the sweep's tile loop (RECOGYM_ABLATE bits 4, 5, 7, 8) exist in -DRG_SWEEP_TIMING builds only: the
// same kind of test cost the wide kernel's loop 20 %
#ifdef RG_SWEEP_TIMING
#define RG_SWEEP_ABL(bit) (d.ablate & (bit))
#else
#define RG_SWEEP_ABL(bit) (false)
#endif

#if RG_HAS(4)
template <int KH, int N1, int N2, int N3, bool F16>
__global__ void __launch_bounds__(kBlock, 2) k_draw_bf16p(DevSim d, uint32_t t, uint32_t S) {
    constexpr int NM = F16 ? N1 : N1 + N2 + N3;   // MFMAs per chunk (fp16 two-way split: one group)
    // MFMA slots that carry the exps (and the A loads); the rest carry the mu loads.  The fp16 form is VALU-bound:
    // its exps spread over all slots but the last
    constexpr int EXS = F16 ? (NM > 1 ? NM - 1 : 1) : (NM > 3 ? NM - 3 : 1);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t tile_b = d.TPB * d.RS;                             // bytes per split tile
    char* g_buf = smem_raw;                                           // [3][TPB][RS]: the tile in use and two in flight
    float* mu_buf = reinterpret_cast<float*>(g_buf + 3 * tile_b);     // [3][TPB] (+ pad)
    float* om_stage = mu_buf + 3 * d.TPB + 64;                        // [4 waves][32 users][2KH] omega32
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();   // scalar: per-wave pointers stay in SGPRs
    const int j = lane & 31, h = lane >> 5;
    // (the walked run's sweep, sweep_only: every user of the launch's group [grp_lo, grp_lo + grp_n) is organic at t = 0 and the
    // list is still the identity — the pipeline sweeps one group per launch)
    const uint32_t pos0 = d.sweep_only ? d.grp_lo : 0u;
    const uint32_t n_o = d.sweep_only ? pos0 + d.grp_n : d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_tiles = (n_o - pos0 + 127) / 128;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    const uint32_t scps = (d.n_sc + S - 1) / S;                       // super-chunks per slice
    const uint32_t n_work = n_tiles * S;
    float* omu = om_stage + (wave * 32 + j) * 2 * KH;                 // this lane's user's omega32

    struct PairOps { bf16x8 A0[N1], A1[N1]; };

    for (uint32_t wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
        const uint32_t tb = wk / S, slice = wk % S;
        const uint32_t chunk_lo = min(slice * scps * d.sc_chunks, d.n_chunks);
        const uint32_t chunk_hi = min((slice + 1) * scps * d.sc_chunks, d.n_chunks);
        if (chunk_lo >= chunk_hi) continue;
        const uint32_t pt_lo = chunk_lo / 4, pt_hi = (chunk_hi + 3) / 4;   // product tiles (TPB = 128: 4 chunks each)
        const uint32_t np = 2 * (pt_hi - pt_lo);                          // pairs of chunks
        const size_t wslot = (S == 1 ? static_cast<size_t>(blockIdx.x) : static_cast<size_t>(tb)) * 4 + wave;
        float* scr_chunk = d.chunk_scratch + wslot * d.n_chunks * 32;
        float2* scr = d.sc_scratch + wslot * kMaxSC * 32;
        const uint32_t pos = pos0 + tb * 128 + wave * 32 + j;
        const bool active = pos < n_o;
        const uint32_t slot = active ? cur[pos] : 0u;
        const SumsView view = sums_view(d, scr, scr_chunk, j, active, slot);
        __syncthreads();           // every wave is done with the LDS buffers and stage (previous work item)
        // tile ti -> LDS buffer ti & 1 (async DMA).  Source = buffer resource (SGPRs) + scalar offset +
        // lane * 16: one VGPR of address state, nothing to spill/reload next to the DMA
        const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
        const rg_v4i rs_g = raw_buffer_rsrc(d.gsplit), rs_m = raw_buffer_rsrc(d.mu32s);
        const uint32_t g_lds = lds_addr_of(g_buf), mu_lds = lds_addr_of(mu_buf);
        auto fetch_tile = [&](uint32_t ti) {
            constexpr uint32_t TB = 128u * (32u * N1 + 16u);
            for (uint32_t off = static_cast<uint32_t>(wave) * 1024u; off < TB; off += 4096u)
                dma_to_lds_b128(rs_g, g_lds + ((ti - pt_lo) % 3u) * TB + off, lane16, ti * TB + off);
            if (wave == 3 && lane < 32) dma_to_lds_b128(rs_m, mu_lds + ((ti - pt_lo) % 3u) * 512u, lane16, ti * 512u);
        };
        fetch_tile(pt_lo);
        // ---- omega32 of the user -> LDS stage (also the logit error bound) ----
        float absdot = 0.0f, sq = 0.0f, absw = 0.0f;
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            const uint32_t k = h * KH + s;
            float w = 0.0f;
            if (active && k < d.K) w = static_cast<float>(d.omega[static_cast<size_t>(slot) * d.OMS + k]);
            omu[k] = w;
            absdot = fmaf(fabsf(w), d.stats[k], absdot);
            sq = fmaf(w, w, sq);
            absw += fabsf(w);
        }
        absdot += swap32(absdot);
        sq += swap32(sq);
        absw += swap32(absw);
        const float Ahat = ahat_of(d, mumax, g2max, absdot, sq);
        const double delta_fixed = kDeltaFixedBf16 + (F16 ? f16_extra_delta(d, Ahat, absw) : 0.0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- B fragments, all three groups in MFMA order: [w1|w1|w1|-q] (N1), [w2|w2|0] (N2), [w3|0|0] (N3) ----
        bf16x8 Bm[NM];
        {
            const uint32_t K = d.K;
#pragma unroll
            for (int s = 0; s < N1; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t ke = 16 * s + 8 * h + e;
                    if (F16) {                                 // [w1 | w1 | w2 | 0 .. | -q]
                        unsigned short sp[2] = {0, 0};
                        if (ke < 3 * K) f16_split2(omu[ke % K], sp);
                        Bm[s][e] = static_cast<short>(ke < 2 * K ? sp[0] : sp[1]);
                    } else {
                        unsigned short sp[3] = {0, 0, 0};
                        if (ke < 3 * K) bf16_split3(omu[ke % K], sp);
                        Bm[s][e] = static_cast<short>(sp[0]);
                        if (s < N2) Bm[(N1 + (s < N2 ? s : 0)) % NM][e] = static_cast<short>(ke < 2 * K ? sp[1] : 0);
                        if (s < N3) Bm[(N1 + N2 + (s < N3 ? s : 0)) % NM][e] = static_cast<short>(ke < K ? sp[2] : 0);
                    }
                }
        }
        float q = 0.0f;            // reference (log2 units, an integer) of the MFMAs being issued
        auto set_reference = [&](float qn) {
            if (F16) {
                // one fp16 piece: an integer |q| <= 2047 is exact (beyond that nothing certifies anyway)
                qn = fminf(fmaxf(qn, -2047.0f), 2047.0f);
                q = qn;
                if (h == 1) Bm[N1 - 1][7] = static_cast<short>(__builtin_bit_cast(unsigned short, static_cast<_Float16>(-qn)));
                return;
            }
            q = qn;
            unsigned short sp[3];
            bf16_split3(-qn, sp);
            if (h == 1) {
                Bm[N1 - 1][5] = static_cast<short>(sp[0]);
                Bm[N1 - 1][6] = static_cast<short>(sp[1]);
                Bm[N1 - 1][7] = static_cast<short>(sp[2]);
            }
        };
        // this lane's operand rows in buffer 0, pair 0 (everything else is a constant offset from these)
        constexpr uint32_t RSc = 32 * N1 + 16, TILE_B = 128 * RSc;
        const char* a_lane = g_buf + j * RSc + 16 * h;
        const char* m_lane = reinterpret_cast<const char*>(mu_buf) + 16 * h;
        auto a_base = [&](uint32_t pi) { return a_lane + ((pi >> 1) % 3u) * TILE_B + (pi & 1) * (64 * RSc); };
        auto m_base = [&](uint32_t pi) { return m_lane + ((pi >> 1) % 3u) * (128 * 4) + (pi & 1) * (64 * 4); };
        auto load_a = [&](PairOps& o, const char* ab, int idx) {        // A row block idx of the pair's chunk 0 / 1
            if (idx < N1) o.A0[idx < N1 ? idx : 0] = *reinterpret_cast<const bf16x8*>(ab + 32 * idx);
            else o.A1[idx - N1 < N1 ? idx - N1 : 0] = *reinterpret_cast<const bf16x8*>(ab + 32 * RSc + 32 * (idx - N1));
        };
        auto load_mu = [&](f32x16& acc, const char* mb, int which, int qq) {   // mu quad qq, into the accumulator it seeds
            const float4 m = *reinterpret_cast<const float4*>(mb + 128 * which + 32 * qq);
            acc[4 * qq] = m.x; acc[4 * qq + 1] = m.y; acc[4 * qq + 2] = m.z; acc[4 * qq + 3] = m.w;
        };
        auto amap = [](int m) { return F16 ? m : (m < N1 ? m : (m < N1 + N2 ? m - N1 : m - N1 - N2)); };
        auto mm = [](const bf16x8& a, const bf16x8& b, const f32x16& c) -> f32x16 {
            using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
            if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        };
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        // One pair: MFMAs of (cur) into (a0, a1), which already hold the pair's mu | exp-sum of (p0, p1) ->
        // (s0, s1) | A rows of pair pi_next -> nxt, its mu -> (p0, p1) once their exps are done.
        // The exps feed four running packed sums per chunk as they are produced (slots < EXS), so
        // the logit registers are free for the mu quads fetched in the last slots.
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        auto stream = [&](const PairOps& co, PairOps& no, uint32_t pi_next, f32x16& a0, f32x16& a1,
                          f32x16& p0, f32x16& p1, float& s0, float& s1) {
            f32x2 x0[4], x1[4];
            const char* ab = a_base(pi_next);
            const char* mb = m_base(pi_next);
            RG_PIN();
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                a0 = mm(co.A0[amap(m)], Bm[m], a0);
                if (m < EXS) {
                    asm volatile("" : "+v"(p0));                        // (exps may not float above this slot)
#pragma unroll
                    for (int i = (2 * m) * (2 * N1) / (2 * EXS); i < (2 * m + 1) * (2 * N1) / (2 * EXS); ++i) load_a(no, ab, i);
#pragma unroll
                    for (int e = (m * 8 / EXS) * 2; e < ((m + 1) * 8 / EXS) * 2; e += 2) {       // exps in pairs
                        f32x2 y = {__builtin_amdgcn_exp2f(p0[e]), __builtin_amdgcn_exp2f(p0[e + 1])};
                        asm volatile("" : "+v"(y));                     // with the pin on p0 above: keeps these pure ops in this slot
                        // (four independent running sums)
                        if (e < 8) x0[e / 2] = y; else x0[(e / 2) & 3] += y;
                    }
                } else {
#pragma unroll
                    for (int qq = (m - EXS) * 4 / (NM > EXS ? NM - EXS : 1); qq < (m - EXS + 1) * 4 / (NM > EXS ? NM - EXS : 1); ++qq) load_mu(p0, mb, 0, qq);
                }
                RG_PIN();
                a1 = mm(co.A1[amap(m)], Bm[m], a1);
                if (m < EXS) {
                    asm volatile("" : "+v"(p1));
#pragma unroll
                    for (int i = (2 * m + 1) * (2 * N1) / (2 * EXS); i < (2 * m + 2) * (2 * N1) / (2 * EXS); ++i) load_a(no, ab, i);
#pragma unroll
                    for (int e = (m * 8 / EXS) * 2; e < ((m + 1) * 8 / EXS) * 2; e += 2) {
                        f32x2 y = {__builtin_amdgcn_exp2f(p1[e]), __builtin_amdgcn_exp2f(p1[e + 1])};
                        asm volatile("" : "+v"(y));                     // with the pin on p1 above: keeps these pure ops in this slot
                        if (e < 8) x1[e / 2] = y; else x1[(e / 2) & 3] += y;
                    }
                } else {
#pragma unroll
                    for (int qq = (m - EXS) * 4 / (NM > EXS ? NM - EXS : 1); qq < (m - EXS + 1) * 4 / (NM > EXS ? NM - EXS : 1); ++qq) load_mu(p1, mb, 1, qq);
                }
                RG_PIN();
            }
            if (NM == EXS) {       // single-MFMA class: no slot left for the mu quads
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) { load_mu(p0, mb, 0, qq); load_mu(p1, mb, 1, qq); }
            }
            x0[0] += x0[2]; x0[1] += x0[3]; x0[0] += x0[1];
            x1[0] += x1[2]; x1[1] += x1[3]; x1[0] += x1[1];
            s0 = x0[0][0] + x0[0][1];
            s1 = x1[0][0] + x1[0][1];
            RG_PIN();
        };
        auto tree = [](const f32x16& y) -> float {
            f32x2 x0 = {y[0], y[1]}, x1 = {y[2], y[3]}, x2 = {y[4], y[5]}, x3 = {y[6], y[7]};
            const f32x2 x4 = {y[8], y[9]}, x5 = {y[10], y[11]}, x6 = {y[12], y[13]}, x7 = {y[14], y[15]};
            x0 += x4; x1 += x5; x2 += x6; x3 += x7; x0 += x2; x1 += x3; x0 += x1;
            return x0[0] + x0[1];
        };

        // ---- per-chunk bookkeeping, one pair behind the MFMAs ----
        double s_sc = 0.0;         // running exp-sum of the super-chunk being summed
        float wcmax = 0.0f;        // its largest chunk sum
        int n_resc = 0;
        float q_done = 0.0f;       // reference the pending sums were taken with
        float q_next = 0.0f;       // reference to switch to at the next super-chunk start
        const bool prefix_mode = d.sweep_only == 2u && S == 1 && d.use_cache;
        double run_pref = 0.0;     // prefix_mode: running prefix of the chunk sums ...
        float q_run = 0.0f;        // ... on this reference (0 = not started: the first tile sets it)
        float* scp_row = prefix_mode ? d.walk_scp + (active ? static_cast<size_t>(d.uid[slot]) : static_cast<size_t>(d.n_cap)) * kMaxSC : nullptr;
        uint32_t sc_cur = chunk_lo / d.sc_chunks;
        uint32_t sc_left = d.sc_chunks / 4;                    // tiles left in it
        float2 wlo = make_float2(0.f, 0.f);
        auto book = [&](uint32_t pe, float s0, float s1) {    // sums of pair pe (chunks 2pe, 2pe+1 of the work item)
            if RG_SWEEP_ABL(256u) { wcmax += s0 + s1; return; }
            s0 += swap32(s0);
            s1 += swap32(s1);
            if (!(pe & 1)) { wlo = make_float2(s0, s1); return; }
            const uint32_t ti = pt_lo + (pe >> 1);
            const float4 w4 = make_float4(wlo.x, wlo.y, s0, s1);
            // scratch layout [tile][user][4 chunks]; both lanes of the user hold the same sums: one of them stores
            // (unpredicated, the duplicate store doubled the kernel's write traffic: 3.1 KB per draw, profiles/r2)
            if (prefix_mode) {
                // k_walk2's form: the running prefix (float64) on the reference these sums were taken with, rounded to fp32;
                // a reference switch rescales the running sum exactly (power of two) — the entries stored before it stay
                // on theirs and are rescaled by k_cache_prefix for the (rare) users it happened to (cache_resc != 0)
                if (q_done != q_run) { run_pref *= static_cast<double>(__builtin_amdgcn_exp2f(q_run - q_done)); q_run = q_done; }
                // (fp32 inside the tile, on the fp32 rounding of the float64 running prefix: <= 5 roundings of 2^-24 relative to the
                // prefix — part of the 2^-21 the header's delta grants the stored prefixes — and ONE float64 add per tile: this
                // kernel is bound by its VALU work)
                const float base = static_cast<float>(run_pref);
                const float p1 = w4.x, p2 = p1 + w4.y, p3 = p2 + w4.z, p4 = p3 + w4.w;
                run_pref += static_cast<double>(p4);
                if (h == 0) *reinterpret_cast<float4*>(view.chunk + static_cast<size_t>(ti) * view.tile_stride) =
                    make_float4(base + p1, base + p2, base + p3, base + p4);
            } else
            if (h == 0 && !RG_SWEEP_ABL(16u)) *reinterpret_cast<float4*>(view.chunk + static_cast<size_t>(ti) * view.tile_stride) = w4;
            wcmax = fmaxf(fmaxf(wcmax, fmaxf(w4.x, w4.y)), fmaxf(w4.z, w4.w));
            s_sc += static_cast<double>((w4.x + w4.y) + (w4.z + w4.w));
            if (--sc_left == 0) {
                if (prefix_mode && h == 0) scp_row[sc_cur] = static_cast<float>(run_pref);
                if (h == 0) view.rec[sc_cur * view.rec_stride] = make_float2(static_cast<float>(s_sc), q_done);
                s_sc = 0.0;
                // some logit is >= ~43 above the reference: re-reference from the next super-chunk
                // that has not started (its MFMAs are a pair ahead of these sums)
                if (wcmax > 2.8e14f) q_next = fmaxf(q_next, q_done + floorf(__builtin_amdgcn_logf(wcmax)));
                wcmax = 0.0f;
                ++sc_cur;
                sc_left = d.sc_chunks / 4;
            }
        };

        PairOps oa, ob;
        f32x16 a0, a1, p0, p1;
        RG_DMA_WAIT();
        __syncthreads();           // tile pt_lo landed
        if (pt_lo + 1 < pt_hi) fetch_tile(pt_lo + 1);
        if (pt_lo + 2 < pt_hi) fetch_tile(pt_lo + 2);
#pragma unroll
        for (int i = 0; i < 2 * N1; ++i) load_a(oa, a_base(0), i);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { load_mu(a0, m_base(0), 0, qq); load_mu(a1, m_base(0), 1, qq); load_mu(p0, m_base(0), 0, qq); }
        RG_PIN();
        {   // first chunk with reference 0: its max (an integer after ceil, so exact in bf16 pieces
            // and in exp2 differences) becomes the reference
#pragma unroll
            for (int m = 0; m < NM; ++m) p0 = mm(oa.A0[amap(m)], Bm[m], p0);
            float cm = p0[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) cm = fmaxf(cm, p0[r]);
            set_reference(fmaxf(ceilf(fmaxf(cm, swap32(cm))), -1.0e30f));
            q_done = q_next = q;
        }
        // head: pair 0's MFMAs with nothing to exp yet; pair 1's A rows and mu arrive meanwhile
        RG_PIN();
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            a0 = mm(oa.A0[amap(m)], Bm[m], a0);
            if (m < N1) load_a(ob, a_base(1), m);
            else if (m < N1 + 4) load_mu(p0, m_base(1), 0, m - N1);
            RG_PIN();
            a1 = mm(oa.A1[amap(m)], Bm[m], a1);
            if (m < N1) load_a(ob, a_base(1), N1 + m);
            else if (m < N1 + 4) load_mu(p1, m_base(1), 1, m - N1);
            RG_PIN();
        }
        if (NM < N1 + 4) {
#pragma unroll
            for (int qq = (NM > N1 ? NM - N1 : 0); qq < 4; ++qq) { load_mu(p0, m_base(1), 0, qq); load_mu(p1, m_base(1), 1, qq); }
        }
        RG_PIN();
        uint32_t sc_issue_left = d.sc_chunks / 4;              // tiles left in the super-chunk being ISSUED
        // Steady state, straight-line (no branch touches an accumulator, or the allocator starts copying
        // 16-register tuples around): [second pair of tile T | first pair of tile T + 1] per iteration.
        uint32_t pi = 1;
        for (; pi + 1 < np; pi += 2) {
            float s0, s1;
            const uint32_t T = pt_lo + (pi >> 1);
            // ---- tile barrier: every wave holds tile T's operands (its buffer is refilled with tile T + 3);
            // tile T + 1 has landed.  Issued behind its DMA, per wave: the DMA of tile T + 2 (>= 4 operations) and
            // the scratch stores of the tiles finished since (0, 1, then always 2) -> those may stay in flight ----
            if (T + 2 >= pt_hi) RG_TILE_BARRIER(0);            // nothing was issued behind tile T + 1 but stores
            else if (pi == 1) RG_TILE_BARRIER(4);               // DMA(T + 2)
            else if (pi == 3) RG_TILE_BARRIER(5);               // + one store
            else RG_TILE_BARRIER(6);                            // + two stores
            if (T + 3 < pt_hi && !RG_SWEEP_ABL(32u)) fetch_tile(T + 3);
            stream(ob, oa, pi + 1, p0, p1, a0, a1, s0, s1);                  // MFMAs of pair pi | sums of pair pi - 1
            book(pi - 1, s0, s1);
            if (--sc_issue_left == 0) sc_issue_left = d.sc_chunks / 4;
            // ---- first pair of tile T + 1 ----
            const bool sc_start = sc_issue_left == d.sc_chunks / 4;          // a super-chunk starts: the pending sums
            if (sc_start) {                                                  // belong to the one before
                q_done = q;
                if (q_next != q) { set_reference(q_next); n_resc += 1; }
            }
            stream(oa, ob, pi + 2, a0, a1, p0, p1, s0, s1);                  // MFMAs of pair pi + 1 | sums of pair pi
            book(pi, s0, s1);                                                // (may flush the finished super-chunk with q_done)
            if (sc_start) q_done = q;
        }
        {   // the last pair (second pair of the last tile), then its own sums
            float s0, s1;
            stream(ob, oa, pi, p0, p1, a0, a1, s0, s1);                      // (operand fetch of a "next" pair: this one again, unused)
            book(pi - 1, s0, s1);
#pragma unroll
            for (int r = 0; r < 16; ++r) { p0[r] = __builtin_amdgcn_exp2f(p0[r]); p1[r] = __builtin_amdgcn_exp2f(p1[r]); }
            q_done = q;
            book(pi, tree(p0), tree(p1));
        }
        if (sc_left != d.sc_chunks / 4 && h == 0) {            // partial last super-chunk
            view.rec[sc_cur * view.rec_stride] = make_float2(static_cast<float>(s_sc), q_done);
            if (prefix_mode) scp_row[sc_cur] = static_cast<float>(run_pref);
        }
        if (d.use_cache && S == 1 && active && h == 0) d.cache_resc[d.uid[slot]] = static_cast<uint8_t>(min(n_resc, 255));
        if (prefix_mode && scp_row && d.fin_in_sweep && active && h == 0 && n_resc == 0) {
            // what k_cache_finalize and k_cache_prefix would leave for this user (one reference for the whole sweep: nothing to
            // rescale): Q and the certificate's delta in its cache row, omega32 behind them, the unused super-chunk prefixes,
            // the hot row {S~, delta + 2^-21 for the stored prefixes' roundings, Q, an empty memo}
            const size_t urow = d.uid[slot];
            float4* row4 = reinterpret_cast<float4*>(d.cache_row + urow * d.cache_row_f);
            const double delta = static_cast<double>(d.K + 5) * 5.9604644775390625e-08 * static_cast<double>(Ahat) + delta_fixed;
            const float dlt = static_cast<float>(delta * 1.000001);          // rounded up: the budget must not shrink
            row4[8] = make_float4(q, dlt, 0.0f, 0.0f);
            const float* ou = om_stage + (wave * 32 + j) * 2 * KH;
#pragma unroll
            for (int k4 = 0; k4 < (2 * KH) / 4; ++k4) row4[11 + k4] = make_float4(ou[4 * k4], ou[4 * k4 + 1], ou[4 * k4 + 2], ou[4 * k4 + 3]);
#pragma unroll
            for (int k = ((2 * KH) / 4) * 4; k < 2 * KH; ++k) reinterpret_cast<float*>(row4)[44 + k] = ou[k];
            for (uint32_t sc = d.n_sc; sc < kMaxSC; ++sc) scp_row[sc] = INFINITY;
            *reinterpret_cast<float4*>(d.walk_hot + urow * 32) =
                make_float4(static_cast<float>(run_pref), dlt * 1.000001f + 4.8e-7f, q, __builtin_bit_cast(float, 0u));
        }
        if (S == 1 && !RG_SWEEP_ABL(128u) && !d.sweep_only) search_and_emit<KH>(d, t, scr, scr_chunk, omu, Ahat, n_resc, active, pos, slot, j, h, true, delta_fixed, &view);
    }
}
#endif

// second kernel of the sliced mode: the search over the sums all slices of a user tile left
#if RG_HAS(3)
template <int KH>
__global__ void __launch_bounds__(kBlock) k_draw_search(DevSim d, uint32_t t) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* om_stage = reinterpret_cast<float*>(smem_raw);             // [4 waves][32 users][2KH]
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int j = lane & 31, h = lane >> 5;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_tiles = (n_o + 127) / 128;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    float* omu = om_stage + (wave * 32 + j) * 2 * KH;
    for (uint32_t tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
        const size_t wslot = static_cast<size_t>(tb) * 4 + wave;
        const uint32_t pos = tb * 128 + wave * 32 + j;
        const bool active = pos < n_o;
        const uint32_t slot = active ? cur[pos] : 0u;
        float absdot = 0.0f, sq = 0.0f, absw = 0.0f;
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            const uint32_t k = h * KH + s;
            float w = 0.0f;
            if (active && k < d.K) w = static_cast<float>(d.omega[static_cast<size_t>(slot) * d.OMS + k]);
            omu[k] = w;
            absdot = fmaf(fabsf(w), d.stats[k], absdot);
            sq = fmaf(w, w, sq);
            absw += fabsf(w);
        }
        absdot += swap32(absdot);
        sq += swap32(sq);
        absw += swap32(absw);
        const float Ahat = ahat_of(d, mumax, g2max, absdot, sq);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const SumsView view = sums_view(d, d.sc_scratch + wslot * kMaxSC * 32, d.chunk_scratch + wslot * d.n_chunks * 32, j, active, slot);
        const int n_resc = (d.use_cache && active) ? d.cache_resc[d.uid[slot]] : 0;
        search_and_emit<KH>(d, t, d.sc_scratch + wslot * kMaxSC * 32, d.chunk_scratch + wslot * d.n_chunks * 32, omu,
                            Ahat, n_resc, active, pos, slot, j, h, true, kDeltaFixedBf16 + (d.f16 ? f16_extra_delta(d, Ahat, absw) : 0.0), &view);
        __builtin_amdgcn_wave_barrier();
    }
}
#endif


// ------------------------------------------------------------------------------------------
// k_draw_cached — the organic draw of a user whose exp-sums are in the per-user cache
// (sigma_omega == 0, every step after the first): search only, no product sweep.
//
// Phase 1-2, a lane per user (64 per wave): the user's <= 32 super-chunk records (256 contiguous
// bytes) -> total, target u S, super-chunk; the chunk sums of that super-chunk -> chunk.
// Phase 3, two users at a time, a lane per product: the 32 products of the chosen chunk are
// recomputed in fp32 from Gamma32 stored chunk by chunk and k-major (every load is one 128-byte run per
// user; a lane-per-user gather of 32 rows cost 88 scattered 16-byte loads per lane and made the first
// version of this path address-rate-bound: 1.06 ns per draw), prefix sum across the 32 lanes,
// index, the two neighbouring prefix values and the margin certificate of search_and_emit.  The
// result travels back to the user's own lane; rows, view history and the hand-over to the float64
// resolve are done a lane per user again.
// ------------------------------------------------------------------------------------------
// after step 0 (slot == user index: nothing has been repacked yet), a lane per user
#if RG_HAS(4)
template <int KH>
__global__ void __launch_bounds__(kBlock) k_cache_finalize(DevSim d) {
    constexpr int K2 = 2 * KH;
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    float gsum = 0.0f;
    if (d.f16) for (uint32_t k = 0; k < d.K; ++k) gsum += d.stats[k];
    for (uint32_t i = d.grp_lo + blockIdx.x * kBlock + threadIdx.x; i < d.grp_lo + d.grp_n; i += gridDim.x * kBlock) {
        if (d.fin_in_sweep && d.cache_resc[i] == 0) continue;        // (the sweep left this user's row itself)
        // everything is staged in registers and leaves as 16-byte stores (a row is 256-byte aligned)
        float4* row4 = reinterpret_cast<float4*>(d.cache_row + static_cast<size_t>(i) * d.cache_row_f);
        // omega32 and the logit error bound, exactly as the sweep kernel computes them
        float om[K2];
        float absdot = 0.0f, sq = 0.0f, absw = 0.0f;
        {
            const double* om_row = d.omega + static_cast<size_t>(i) * d.OMS;
#pragma unroll
            for (int k2 = 0; k2 < KH; ++k2) {
                double2 w2 = make_double2(0.0, 0.0);
                if (static_cast<uint32_t>(2 * k2) < d.K) w2 = *reinterpret_cast<const double2*>(om_row + 2 * k2);
                om[2 * k2] = static_cast<float>(w2.x);
                om[2 * k2 + 1] = static_cast<uint32_t>(2 * k2 + 1) < d.K ? static_cast<float>(w2.y) : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < K2; ++k) {
                absdot = fmaf(fabsf(om[k]), d.stats[k], absdot);
                sq = fmaf(om[k], om[k], sq);
                absw += fabsf(om[k]);
            }
        }
        const float Ahat = ahat_of(d, mumax, g2max, absdot, sq);
        double delta = static_cast<double>(d.K + 5) * 5.9604644775390625e-08 * static_cast<double>(Ahat) + kDeltaFixedBf16 +
                       kDeltaPerRescale * static_cast<double>(d.cache_resc[i]);
        if (d.f16) delta += 12.0 * 5.9604644775390625e-08 * static_cast<double>(Ahat) +
                            2.98023223876953125e-08 * (static_cast<double>(gsum) + 0.6931471805599453 * static_cast<double>(absw));
        float2 rec[kMaxSC];
        {
            const float4* rp = reinterpret_cast<const float4*>(d.cache_rec + static_cast<size_t>(i) * kMaxSC);
#pragma unroll
            for (uint32_t q = 0; q < kMaxSC / 2; ++q) {
                const float4 x = rp[q];
                rec[2 * q] = make_float2(x.x, x.y); rec[2 * q + 1] = make_float2(x.z, x.w);
            }
        }
        float Q = -INFINITY;
#pragma unroll
        for (uint32_t sc = 0; sc < kMaxSC; ++sc) if (sc < d.n_sc) Q = fmaxf(Q, rec[sc].y);
        uint32_t offw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float W[kMaxSC];
#pragma unroll
        for (uint32_t sc = 0; sc < kMaxSC; ++sc) {
            float x = 0.0f;
            uint32_t off = 127u;                                        // unused / out of range: weight 0, never chosen
            if (sc < d.n_sc) {
                const float dq = Q - rec[sc].y;                         // references are integers (log2 units)
                if (dq < 127.0f) { off = static_cast<uint32_t>(dq); x = rec[sc].x * __builtin_amdgcn_exp2f(-dq); }
            }
            W[sc] = x;
            offw[sc >> 2] |= off << (8 * (sc & 3));
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) row4[q] = make_float4(W[4 * q], W[4 * q + 1], W[4 * q + 2], W[4 * q + 3]);
        row4[8] = make_float4(Q, static_cast<float>(delta * 1.000001), 0.0f, 0.0f);   // delta rounded up: the budget must not shrink
        row4[9] = make_float4(__builtin_bit_cast(float, offw[0]), __builtin_bit_cast(float, offw[1]),
                              __builtin_bit_cast(float, offw[2]), __builtin_bit_cast(float, offw[3]));
        row4[10] = make_float4(__builtin_bit_cast(float, offw[4]), __builtin_bit_cast(float, offw[5]),
                               __builtin_bit_cast(float, offw[6]), __builtin_bit_cast(float, offw[7]));
#pragma unroll
        for (int k4 = 0; k4 < K2 / 4; ++k4) row4[11 + k4] = make_float4(om[4 * k4], om[4 * k4 + 1], om[4 * k4 + 2], om[4 * k4 + 3]);
#pragma unroll
        for (int k = (K2 / 4) * 4; k < K2; ++k) reinterpret_cast<float*>(row4)[44 + k] = om[k];
    }
}
#endif

#if RG_HAS(4)
template <int KH>
__global__ void __launch_bounds__(kBlock, (KH <= 16 ? 3 : 2)) k_draw_cached(DevSim d, uint32_t t) {
    constexpr int K2 = 2 * KH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    float* om_w = reinterpret_cast<float*>(smem_raw) + static_cast<size_t>(wave) * 64 * K2;   // [64 users][K2] omega32
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const uint32_t n_groups = (n_o + 63) / 64;
    for (uint32_t grp = blockIdx.x * (kBlock / 64) + wave; grp < n_groups; grp += gridDim.x * (kBlock / 64)) {
        const uint32_t pos = grp * 64 + lane;
        const bool active = pos < n_o;
        const uint32_t slot = active ? cur[pos] : 0u;
        const uint32_t uidx = active ? d.uid[slot] : 0u;
        const size_t row = active ? uidx : d.n_cap;                       // inactive lanes read the dummy row
        // ---- phase 1: the user's row — scaled super-chunk sums, reference, delta, offsets, omega32 ----
        const float4* rp = reinterpret_cast<const float4*>(d.cache_row + row * d.cache_row_f);
        float W[kMaxSC];
#pragma unroll
        for (int i = 0; i < kMaxSC / 4; ++i) {
            const float4 x = rp[i];
            W[4 * i] = x.x; W[4 * i + 1] = x.y; W[4 * i + 2] = x.z; W[4 * i + 3] = x.w;
        }
        const float4 hdr = rp[8];
        const float4 of0 = rp[9], of1 = rp[10];
        {
            float* o = om_w + lane * K2;
#pragma unroll
            for (int k4 = 0; k4 < K2 / 4; ++k4) *reinterpret_cast<float4*>(o + 4 * k4) = rp[11 + k4];
#pragma unroll
            for (int k = (K2 / 4) * 4; k < K2; ++k) o[k] = reinterpret_cast<const float*>(rp)[44 + k];
        }
        const float Q = hdr.x;
        const double delta = static_cast<double>(hdr.y);
        double S = 0.0;
#pragma unroll
        for (uint32_t sc = 0; sc < kMaxSC; ++sc) S += static_cast<double>(W[sc]);     // unused records hold 0
        const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
        const double u_draw = organic_uniform(d, uidx, user, t);
        const double tau = u_draw * S;
        double pb = 0.0;
        uint32_t sc_star = d.n_sc - 1;
        bool found_sc = false;
        {
            double run = 0.0;
#pragma unroll
            for (uint32_t sc = 0; sc < kMaxSC; ++sc) {
                const double Wd = static_cast<double>(W[sc]);
                if (sc < d.n_sc && !found_sc && run + Wd > tau) { found_sc = true; sc_star = sc; pb = run; }
                if (sc < d.n_sc && !found_sc) run += Wd;
            }
        }
        // scale of that super-chunk's chunk sums: 2^-(offset of its reference)
        uint32_t offw;
        {
            const uint32_t q = sc_star >> 2;
            const float4 o4 = q < 4 ? of0 : of1;
            const float ow = (q & 3) == 0 ? o4.x : (q & 3) == 1 ? o4.y : (q & 3) == 2 ? o4.z : o4.w;
            offw = (__builtin_bit_cast(uint32_t, ow) >> (8 * (sc_star & 3))) & 0xFFu;
        }
        if (offw >= 127u) found_sc = false;
        const float f_star = found_sc ? __builtin_amdgcn_exp2f(-static_cast<float>(offw)) : 1.0f;
        // ---- phase 2: the chunk inside that super-chunk ----
        uint32_t c_star = 0;
        bool found_c = false;
        {
            const uint32_t c0 = sc_star * d.sc_chunks, c1 = min(c0 + d.sc_chunks, d.n_chunks);
            const float* cp = d.cache_chunk + row * d.n_chunks;
            double run = pb;
            for (uint32_t cb = c0; cb < c1; cb += 16) {
                float4 w4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    w4[i] = cb + 4 * i < c1 ? *reinterpret_cast<const float4*>(cp + cb + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float4 q4 = w4[i >> 2];
                    const float wv = (i & 3) == 0 ? q4.x : (i & 3) == 1 ? q4.y : (i & 3) == 2 ? q4.z : q4.w;
                    const double Wd = static_cast<double>(wv * f_star);
                    const uint32_t c = cb + i;
                    if (c < c1 && !found_c && run + Wd > tau) { found_c = true; c_star = c; pb = run; }
                    if (c < c1 && !found_c) run += Wd;
                }
            }
        }
        found_c = found_c && found_sc;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // omega32 stage written above, read by other lanes below
        __builtin_amdgcn_wave_barrier();
        // ---- phase 3: two users per pass — user i on lanes 0-31, user i + 32 on lanes 32-63 (each user's own lane
        // sits in the half that works for it) — a lane per product of the chosen chunk ----
        const int half = lane >> 5, p = lane & 31;
        const uint32_t n_here = min(32u, n_o - grp * 64);
        uint32_t my_v = 0;
        bool my_ok = false;
        // the table values of pass i + 1 are requested before pass i is worked on: a pass is a chain of ~15
        // dependent cross-lane / memory round trips, and nothing else of this wave would overlap the L2 latency
        float gn[K2], mun;
        uint32_t csn = static_cast<uint32_t>(__shfl(static_cast<int>(c_star), 32 * half));
        {
            const float* gp = d.gamma32t + (static_cast<size_t>(csn) * K2) * 32 + p;
#pragma unroll
            for (int k = 0; k < K2; ++k) gn[k] = gp[k * 32];
            mun = d.mu32[csn * 32 + p];
        }
        if (d.ablate & 1024u) { my_v = c_star * 32; my_ok = c_star * 32 < d.P; }
        else
        for (uint32_t i = 0; i < n_here; ++i) {
            const int src = static_cast<int>(i) + 32 * half;             // the user this half works for
            const uint32_t cs = csn;
            float g[K2];
#pragma unroll
            for (int k = 0; k < K2; ++k) g[k] = gn[k];
            float l = mun;
            if (i + 1 < n_here) {
                csn = static_cast<uint32_t>(__shfl(static_cast<int>(c_star), src + 1));
                const float* gp = d.gamma32t + (static_cast<size_t>(csn) * K2) * 32 + p;
#pragma unroll
                for (int k = 0; k < K2; ++k) gn[k] = gp[k * 32];
                mun = d.mu32[csn * 32 + p];
            }
            const float Qs = __shfl(Q, src);
            const double pbs = __shfl(pb, src), taus = __shfl(tau, src);
            const float* o = om_w + src * K2;
#pragma unroll
            for (int k4 = 0; k4 < K2 / 4; ++k4) {
                const float4 w4 = *reinterpret_cast<const float4*>(o + 4 * k4);
                l = fmaf(g[4 * k4], w4.x, l); l = fmaf(g[4 * k4 + 1], w4.y, l);
                l = fmaf(g[4 * k4 + 2], w4.z, l); l = fmaf(g[4 * k4 + 3], w4.w, l);
            }
#pragma unroll
            for (int k = (K2 / 4) * 4; k < K2; ++k) l = fmaf(g[k], o[k], l);
            const float e = __builtin_amdgcn_exp2f(fmaf(l, kLog2e, -Qs));
            float incl = e;                                              // inclusive prefix over the half's 32 lanes
#pragma unroll
            for (int o2 = 1; o2 < 32; o2 <<= 1) {
                const float y = __shfl_up(incl, o2, 32);
                if (p >= o2) incl += y;
            }
            const double px = pbs + static_cast<double>(incl);
            const unsigned long long hits = __ballot(px > taus);
            const uint32_t hmask = static_cast<uint32_t>(half ? (hits >> 32) : hits);
            const int idx = hmask ? __builtin_ctz(hmask) : -1;
            const int li = half * 32 + max(idx, 0);
            const double Bv = __shfl(px, li);
            const double Av = idx > 0 ? __shfl(px, li - 1) : pbs;
            if (lane == src) {
                const uint32_t v = cs * 32 + static_cast<uint32_t>(max(idx, 0));
                my_v = v;
                const CertLin ct = cert_correlated(S, pb, Av - pb, Bv - pb, delta);
                my_ok = found_c && idx >= 0 && v < d.P && ct.valid &&
                        (v == 0 || u_draw * ct.den_lo > ct.num_lo) &&
                        (v == d.P - 1 || u_draw * ct.den_hi < ct.num_hi);
            }
        }
        // ---- emit (a lane per user) ----
        if (active) {
            if (my_ok) {
                write_organic_row(d, t, pos, slot, user, my_v);
                if (d.hist_cap && !(d.ablate & 2048u)) history_add(d, slot, my_v);
            } else if (d.f64_valid[uidx]) d.exact_list[d.n_cap - 1u - atomicAdd(&d.exact_cnt_b[t], 1u)] = pos;
            else {
                d.exact_list[atomicAdd(&d.exact_cnt[t], 1u)] = pos;
                d.exact_ref[uidx] = Q;
            }
        }
        __builtin_amdgcn_wave_barrier();                                 // the omega32 stage is reused by the next group
    }
}
#endif

#if RG_HAS(4)
finalize_kernel_t finalize_kernel_for(const DevSim& d) {
    switch (d.KH) {
        case 4: return k_cache_finalize<4>;
        case 10: return k_cache_finalize<10>;
        case 16: return k_cache_finalize<16>;
        default: return k_cache_finalize<32>;
    }
}
cached_kernel_t cached_kernel_for(const DevSim& d) {
    switch (d.KH) {
        case 4: return k_draw_cached<4>;
        case 10: return k_draw_cached<10>;
        case 16: return k_draw_cached<16>;
        default: return k_draw_cached<32>;
    }
}
#endif

// kernel selection by (KH, N1, N2, N3)
#if RG_HAS(3)
search_kernel_t search_kernel_for(const DevSim& d) {
    switch (d.KH) {
        case 4: return k_draw_search<4>;
        case 10: return k_draw_search<10>;
        case 16: return k_draw_search<16>;
        case 32: return k_draw_search<32>;
        default: return k_draw_search<64>;
    }
}
#endif
#if RG_HAS(4)
draw_kernel_t bf16p_kernel_for(const DevSim& d) {
    if (d.f16) {
#define RG_CASE(kh, a) if (d.KH == kh && d.N1 == a) return k_draw_bf16p<kh, a, 0, 0, true>;
        RG_CASE(4, 1) RG_CASE(4, 2) RG_CASE(10, 2) RG_CASE(10, 3) RG_CASE(10, 4) RG_CASE(16, 4)
#undef RG_CASE
        return nullptr;
    }
#define RG_CASE(kh, a, b, c) if (d.KH == kh && d.N1 == a && d.N2 == b && d.N3 == c) return k_draw_bf16p<kh, a, b, c, false>;
    RG_CASE(4, 1, 1, 1) RG_CASE(4, 2, 1, 1) RG_CASE(10, 3, 2, 1) RG_CASE(10, 4, 3, 2)
#undef RG_CASE
    return nullptr;
}
#endif
#if RG_HAS(3)
draw_kernel_t bf16_kernel_for(const DevSim& d) {
#define RG_CASE(kh, a, b, c) if (d.KH == kh && d.N1 == a && d.N2 == b && d.N3 == c) return k_draw_bf16<kh, a, b, c>;
    RG_CASE(4, 1, 1, 1) RG_CASE(4, 2, 1, 1) RG_CASE(10, 3, 2, 1) RG_CASE(10, 4, 3, 2)
    RG_CASE(16, 4, 3, 2) RG_CASE(16, 6, 4, 2) RG_CASE(32, 12, 8, 4)
#undef RG_CASE
    return nullptr;
}
mfma_kernel_t mfma_kernel_for(uint32_t KH) {
    switch (KH) {
        case 4: return k_draw_mfma<4>;
        case 10: return k_draw_mfma<10>;
        case 16: return k_draw_mfma<16>;
        case 32: return k_draw_mfma<32>;
        default: return k_draw_mfma<64>;
    }
}
#endif


// ------------------------------------------------------------------------------------------
// k_draw_f16w — the two-way fp16 split sweep for WIDE embeddings (21 < K <= 64: BASELINE config 4's K = 64).
//
// Same arithmetic, table and certificate as k_draw_bf16p<.., F16>: A row = [G1 | G2 | G1 | 0.. | 1], B row =
// [w1 | w1 | w2 | 0.. | -q], N1 = ceil((3K + 1) / 16) k-steps (13 at K = 64) of v_mfma_f32_32x32x16_f16 per
// 32-product chunk.  What differs is the shape around it:
//   * the matrix pipe binds here (13 MFMAs = 416 pipe cycles per chunk against ~220 cycles of exp/sum VALU
//     work), so the A operands are NOT double-buffered per pair in registers (2 x 104 VGPRs at N1 = 13): they
//     are read from the LDS tile k-step by k-step, next to the MFMA that consumes them;
//   * a block is 8 waves = 256 users per pass over the table (the split table is 43 MB at P = 10^5: at 128
//     users per pass the L2 -> LDS stream alone would need ~2/3 of a CU's L2 bandwidth);
//   * tiles are one PAIR of chunks (64 products, 27 KB at N1 = 13), three LDS buffers, DMA two tiles ahead,
//     counted vmcnt at the tile barrier (as in k_draw_bf16p).
// The exp-sums of pair n - 1 sit in the issue slots between the MFMAs of pair n (two independent accumulator
// chains), order pinned with sched_barrier.
// ------------------------------------------------------------------------------------------
// timing experiments of the tile loop (RECOGYM_ABLATE bits 8-14: no book-keeping / exps / table stream / second operand
// read / tile barrier / mu reads / MFMAs) exist in the -DRG_F16W_TIMING build only: the tests alone cost the loop 20 %
#ifdef RG_F16W_TIMING
#define RG_F16W_ABL(bit) (d.ablate & (bit))
#else
#define RG_F16W_ABL(bit) (false)
#endif
#ifdef RG_F16W_TIMING
// -DRG_F16W_TIMING: s_memtime per section of the tile loop, summed over wave 0 of every block (tools/wide_probe.py)
static __device__ unsigned long long g_f16w_t[8];
#define RG_TSEC(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define RG_TSEC(i) do {} while (0)
#endif

#if RG_HAS(5)
template <int KH, int N1, int UG>
__global__ void __launch_bounds__(512 / UG, 1) k_draw_f16w(DevSim d, uint32_t t, uint32_t S) {
    // Nothing that lives across the tile loop may be spilled: a reload inside the loop is followed by `s_waitcnt
    // vmcnt(0)`, which also waits for the tile DMA in flight (the asm DMA is invisible to the compiler's counter
    // model) — a memory round trip per tile and wave.  The mu tile's buffer descriptor and this lane's LDS address
    // were two such values (measured: 3/4 of the kernel's time); they are rebuilt where they are used, the first from
    // the kernel-argument segment.
    const __attribute__((address_space(4))) char* kargs = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr uint32_t RSc = 32 * N1 + 16, TILE_B = 64 * RSc, NT = TILE_B / 1024;     // 1 KB per wave-wide DMA instruction
    // UG groups of 32 users per wave, 8 / UG waves per block (256 users either way).  UG = 2: every A fragment read
    // from LDS feeds two MFMAs (half the LDS traffic) but one wave per SIMD; UG = 1: two waves per SIMD
    constexpr int NW = 8 / UG;
    using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* g_buf = smem_raw;                                           // [3][64][RSc]
    float* mu_buf = reinterpret_cast<float*>(g_buf + 3 * TILE_B);     // [3][64]
    float* om_stage = mu_buf + 3 * 64;                                // [8 groups][32 users][2KH] omega32
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const int j = lane & 31, h = lane >> 5;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_tiles = (n_o + 255) / 256;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    const uint32_t scps = (d.n_sc + S - 1) / S;                       // super-chunks per slice
    const uint32_t n_work = n_tiles * S;
    // this wave's DMA instructions per tile (they complete in issue order: the tile barrier may leave these in flight)
    const int my_dma = static_cast<int>((NT - wave + NW - 1) / NW) + (wave == NW - 1 ? 1 : 0);

    for (uint32_t wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
        const uint32_t tb = wk / S, slice = wk % S;
        const uint32_t chunk_lo = min(slice * scps * d.sc_chunks, d.n_chunks);
        const uint32_t chunk_hi = min((slice + 1) * scps * d.sc_chunks, d.n_chunks);
        if (chunk_lo >= chunk_hi) continue;
        const uint32_t pt_lo = chunk_lo / 2, pt_hi = (chunk_hi + 1) / 2;      // tiles = pairs of chunks
        uint32_t pos[UG], slot[UG];
        bool active[UG];
        SumsView view[UG];
        float* omu[UG];
        float2* scr[UG];
        float* scr_chunk[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            const size_t wslot = (S == 1 ? static_cast<size_t>(blockIdx.x) : static_cast<size_t>(tb)) * (NW * UG) + wave * UG + g;
            scr_chunk[g] = d.chunk_scratch + wslot * d.n_chunks * 32;
            scr[g] = d.sc_scratch + wslot * kMaxSC * 32;
            pos[g] = tb * 256 + (wave * UG + g) * 32 + j;
            active[g] = pos[g] < n_o;
            slot[g] = active[g] ? cur[pos[g]] : 0u;
            view[g] = sums_view(d, scr[g], scr_chunk[g], j, active[g], slot[g]);
            omu[g] = om_stage + ((wave * UG + g) * 32 + j) * 2 * KH;      // this lane's user's omega32
        }
        __syncthreads();           // every wave is done with the LDS buffers and stage (previous work item)
        const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
        const rg_v4i rs_g = raw_buffer_rsrc(d.gsplit);
        const uint32_t g_lds = lds_addr_of(g_buf), mu_lds = lds_addr_of(mu_buf);
        auto fetch_tile = [&](uint32_t ti) {
            if (RG_F16W_ABL(1024u) && ti > pt_lo + 2) return;        // timing experiment: no table stream (stale tiles)
            for (uint32_t off = static_cast<uint32_t>(wave) * 1024u; off < TILE_B; off += NW * 1024u)
                dma_to_lds_b128(rs_g, g_lds + ((ti - pt_lo) % 3u) * TILE_B + off, lane16, ti * TILE_B + off);
            if (wave == NW - 1) {
                asm volatile("" : "+s"(kargs));
                const rg_v4i rs_m = raw_buffer_rsrc(((const DevSim*)kargs)->mu32s);
                if (lane < 16) dma_to_lds_b128(rs_m, mu_lds + ((ti - pt_lo) % 3u) * 256u, lane16, ti * 256u);
            }
        };
        fetch_tile(pt_lo);
        // ---- omega32 of the users -> LDS stage (also the logit error bound) ----
        float Ahat[UG];
        double delta_fixed[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            float absdot = 0.0f, sq = 0.0f, absw = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < KH; ++s2) {
                const uint32_t k = h * KH + s2;
                float w = 0.0f;
                if (active[g] && k < d.K) w = static_cast<float>(d.omega[static_cast<size_t>(slot[g]) * d.OMS + k]);
                omu[g][k] = w;
                absdot = fmaf(fabsf(w), d.stats[k], absdot);
                sq = fmaf(w, w, sq);
                absw += fabsf(w);
            }
            absdot += swap32(absdot);
            sq += swap32(sq);
            absw += swap32(absw);
            Ahat[g] = ahat_of(d, mumax, g2max, absdot, sq);
            delta_fixed[g] = kDeltaFixedBf16 + f16_extra_delta(d, Ahat[g], absw);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- B fragments [w1 | w1 | w2 | 0 .. | -q]: lane (j, h) holds elements ke = 16 s + 8 h + e of its user's row ----
        bf16x8 Bm[UG][N1];
        {
            const uint32_t K = d.K;
#pragma unroll
            for (int g = 0; g < UG; ++g)
#pragma unroll
                for (int s2 = 0; s2 < N1; ++s2)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t ke = 16 * s2 + 8 * h + e;
                        unsigned short sp[2] = {0, 0};
                        if (ke < 3 * K) f16_split2(omu[g][ke % K], sp);
                        Bm[g][s2][e] = static_cast<short>(ke < 2 * K ? sp[0] : sp[1]);
                    }
        }
        float q[UG];               // reference (log2 units, an integer) the MFMAs being issued subtract
#pragma unroll
        for (int g = 0; g < UG; ++g) q[g] = 0.0f;
        auto set_reference = [&](int g, float qn) {
            qn = fminf(fmaxf(qn, -2047.0f), 2047.0f);       // one fp16 piece: an integer |q| <= 2047 is exact
            q[g] = qn;
            if (h == 1) Bm[g][N1 - 1][7] = static_cast<short>(__builtin_bit_cast(unsigned short, static_cast<_Float16>(-qn)));
        };
        auto mm = [](const bf16x8& a, const bf16x8& b, const f32x16& c) -> f32x16 {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        };
        // this lane's operand rows in buffer 0 (chunk 0 of the pair; chunk 1 is 32 rows further)
        const char* a_lane = g_buf + j * RSc + 16 * h;
        const char* m_lane = reinterpret_cast<const char*>(mu_buf) + 16 * h;
        auto load_mu = [&](f32x16& acc, const char* mb, int which) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float4 m = *reinterpret_cast<const float4*>(mb + 128 * which + 32 * qq);
                acc[4 * qq] = m.x; acc[4 * qq + 1] = m.y; acc[4 * qq + 2] = m.z; acc[4 * qq + 3] = m.w;
            }
        };
        // ---- bookkeeping of finished pairs (one pair behind the MFMAs), per user group ----
        double s_sc[UG];           // running exp-sum of the super-chunk being summed
        float wcmax[UG];           // its largest chunk sum
        int n_resc[UG];
        float q_next[UG];          // reference to switch to at the next super-chunk start
#pragma unroll
        for (int g = 0; g < UG; ++g) { s_sc[g] = 0.0; wcmax[g] = 0.0f; n_resc[g] = 0; q_next[g] = 0.0f; }
        const uint32_t sc_pairs = d.sc_chunks / 2;
        uint32_t sc_cur = chunk_lo / d.sc_chunks, sc_left = sc_pairs;
        auto book = [&](int g, uint32_t ti_done, float s0, float s1, float q_used, bool flush) {   // sums of the pair of tile ti_done
            if RG_F16W_ABL(256u) { wcmax[g] += s0 + s1; return; }   // timing experiment: no reduction across lanes, no stores
            s0 += swap32(s0);
            s1 += swap32(s1);
            const uint32_t ci = 2 * ti_done;
            // scratch layout of the 4-chunk tiles the search reads: [tile of 4][user][4 chunks]
            if (h == 0) *reinterpret_cast<float2*>(view[g].chunk + static_cast<size_t>(ci >> 2) * view[g].tile_stride + (ci & 3)) = make_float2(s0, s1);
            wcmax[g] = fmaxf(wcmax[g], fmaxf(s0, s1));
            s_sc[g] += static_cast<double>(s0 + s1);
            if (flush) {
                if (h == 0) view[g].rec[sc_cur * view[g].rec_stride] = make_float2(static_cast<float>(s_sc[g]), q_used);
                s_sc[g] = 0.0;
                // some logit is >= ~43 above the reference: re-reference from the next super-chunk that has not started
                if (wcmax[g] > 2.8e14f) q_next[g] = fmaxf(q_next[g], q_used + floorf(__builtin_amdgcn_logf(wcmax[g])));
                wcmax[g] = 0.0f;
            }
        };
        RG_DMA_WAIT();
        __syncthreads();           // tile pt_lo landed
        if (pt_lo + 1 < pt_hi) fetch_tile(pt_lo + 1);
        if (pt_lo + 2 < pt_hi) fetch_tile(pt_lo + 2);
#pragma unroll
        for (int g = 0; g < UG; ++g) {   // first chunk with reference 0: its max (an integer after ceil, exact in one fp16 piece) becomes the reference
            f32x16 y;
            load_mu(y, m_lane, 0);
#pragma unroll
            for (int s2 = 0; s2 < N1; ++s2) y = mm(*reinterpret_cast<const bf16x8*>(a_lane + 32 * s2), Bm[g][s2], y);
            float cm = y[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) cm = fmaxf(cm, y[r]);
            set_reference(g, fmaxf(ceilf(fmaxf(cm, swap32(cm))), -1.0e30f));
            q_next[g] = q[g];
        }
        f32x16 p[UG][2];           // logits of the previous pair (per group: chunk 0, chunk 1), waiting for their exp-sums
#pragma unroll
        for (int g = 0; g < UG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[g][0][r] = 0.0f; p[g][1][r] = 0.0f; }
        float q_prev[UG];          // references they were taken with
#pragma unroll
        for (int g = 0; g < UG; ++g) q_prev[g] = q[g];
        uint32_t sc_issue_left = sc_pairs;                      // pairs left in the super-chunk being ISSUED
#ifdef RG_F16W_TIMING
        unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
        for (uint32_t ti = pt_lo; ti < pt_hi; ++ti) {
            RG_TSEC(4);
            if (ti > pt_lo && !RG_F16W_ABL(4096u)) {                 // (4096: timing experiment without the tile barrier)
                // tile ti has landed once at most this wave's DMA of tile ti + 1 (issued after it) is still in flight
                if (ti + 1 >= pt_hi) RG_TILE_BARRIER(0);
                else if (my_dma >= 8) RG_TILE_BARRIER(8);
                else if (my_dma == 7) RG_TILE_BARRIER(7);
                else if (my_dma == 6) RG_TILE_BARRIER(6);
                else if (my_dma == 5) RG_TILE_BARRIER(5);
                else if (my_dma == 4) RG_TILE_BARRIER(4);
                else if (my_dma == 3) RG_TILE_BARRIER(3);
                else RG_TILE_BARRIER(2);
                RG_TSEC(0);
                if (ti + 2 < pt_hi) fetch_tile(ti + 2);        // into the buffer of tile ti - 1: every wave is past it
            }
            RG_TSEC(5);
            const uint32_t bsel = (ti - pt_lo) % 3u;
            int hl = lane >> 5, jl = lane & 31;
            asm volatile("" : "+v"(hl), "+v"(jl));                 // (rebuilt here: see the note on spills at the top)
            const char* ab = g_buf + jl * RSc + 16 * hl + bsel * TILE_B;
            const char* mb = reinterpret_cast<const char*>(mu_buf) + 16 * hl + bsel * 256u;
            if (sc_issue_left == sc_pairs) {                    // a super-chunk starts
#pragma unroll
                for (int g = 0; g < UG; ++g) if (q_next[g] != q[g]) { set_reference(g, q_next[g]); n_resc[g] += 1; }
            }
            if (--sc_issue_left == 0) sc_issue_left = sc_pairs;
            f32x16 a[UG][2];
#pragma unroll
            for (int g = 0; g < UG; ++g) {
                if RG_F16W_ABL(8192u) {                                // timing experiment: no mu tile reads
#pragma unroll
                    for (int r = 0; r < 16; ++r) { a[g][0][r] = 0.0f; a[g][1][r] = 0.0f; }
                } else { load_mu(a[g][0], mb, 0); load_mu(a[g][1], mb, 1); }
            }
            f32x2 x[UG][2][4];
            const bool have_p = ti > pt_lo;
            // A operands: a ring RD k-steps deep, read RD - 1 steps ahead of the MFMAs that consume them (one wave per
            // SIMD has nobody to hide an LDS round trip behind: deeper there)
            constexpr int RD = UG == 2 ? 5 : 3;
            bf16x8 A0r[RD], A1r[RD];
#pragma unroll
            for (int s2 = 0; s2 < RD - 1 && s2 < N1; ++s2) {
                A0r[s2] = *reinterpret_cast<const bf16x8*>(ab + 32 * s2);
                A1r[s2] = *reinterpret_cast<const bf16x8*>(ab + 32 * RSc + 32 * s2);
            }
            RG_PIN();
            RG_TSEC(1);
            // the exps of the previous tile (2 UG accumulators x 16) spread over the 2 UG N1 MFMA slots of this one
            constexpr int NSLOT = 2 * UG * N1, NEP = 16 * UG, EPS = (NEP + NSLOT - 1) / NSLOT;      // exp PAIRS (per slot)
            auto exps = [&](int slot_i) {
                if RG_F16W_ABL(512u) return;                          // timing experiment: MFMA stream only
#pragma unroll
                for (int e = slot_i * EPS; e < (slot_i + 1) * EPS && e < NEP; ++e) {
                    const int g = e >> 4, c = (e >> 3) & 1, r = e & 7;       // accumulator (g, c), register pair r
                    asm volatile("" : "+v"(p[g][c]));
                    f32x2 y = {__builtin_amdgcn_exp2f(p[g][c][2 * r]), __builtin_amdgcn_exp2f(p[g][c][2 * r + 1])};
                    asm volatile("" : "+v"(y));
                    if (r < 4) x[g][c][r] = y; else x[g][c][r & 3] += y;
                }
            };
#pragma unroll
            for (int s2 = 0; s2 < N1; ++s2) {
                if (s2 + RD - 1 < N1) {
                    A0r[(s2 + RD - 1) % RD] = *reinterpret_cast<const bf16x8*>(ab + 32 * (s2 + RD - 1));
                    if RG_F16W_ABL(2048u) A1r[(s2 + RD - 1) % RD] = A0r[(s2 + RD - 1) % RD];      // timing experiment: half the LDS operand reads
                    else A1r[(s2 + RD - 1) % RD] = *reinterpret_cast<const bf16x8*>(ab + 32 * RSc + 32 * (s2 + RD - 1));
                }
#pragma unroll
                for (int g = 0; g < UG; ++g) {
                    if (!RG_F16W_ABL(16384u)) a[g][0] = mm(A0r[s2 % RD], Bm[g][s2], a[g][0]);   // (16384: timing experiment without the MFMAs)
                    exps((2 * s2) * UG + g);
                    RG_PIN();
                }
#pragma unroll
                for (int g = 0; g < UG; ++g) {
                    if (!RG_F16W_ABL(16384u)) a[g][1] = mm(A1r[s2 % RD], Bm[g][s2], a[g][1]);
                    exps((2 * s2 + 1) * UG + g);
                    RG_PIN();
                }
            }
            RG_TSEC(2);
            const bool flush = have_p && sc_left == 1;
            if (have_p) {
#pragma unroll
                for (int g = 0; g < UG; ++g) {
                    float sm[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        x[g][c][0] += x[g][c][2]; x[g][c][1] += x[g][c][3]; x[g][c][0] += x[g][c][1];
                        sm[c] = x[g][c][0][0] + x[g][c][0][1];
                    }
                    book(g, ti - 1, sm[0], sm[1], q_prev[g], flush);
                }
                if (flush) { ++sc_cur; sc_left = sc_pairs; } else --sc_left;
            }
#pragma unroll
            for (int g = 0; g < UG; ++g) { p[g][0] = a[g][0]; p[g][1] = a[g][1]; q_prev[g] = q[g]; }
            RG_TSEC(3);
        }
#ifdef RG_F16W_TIMING
        if (wave == 0 && lane == 0) {
            for (int i = 0; i < 6; ++i) atomicAdd(&g_f16w_t[i], tacc[i]);
            atomicAdd(&g_f16w_t[6], static_cast<unsigned long long>(pt_hi - pt_lo));
        }
#endif
        {   // the last pair's own sums
            const bool flush = sc_left == 1;
#pragma unroll
            for (int g = 0; g < UG; ++g) {
                float e0 = 0.0f, e1 = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { e0 += __builtin_amdgcn_exp2f(p[g][0][r]); e1 += __builtin_amdgcn_exp2f(p[g][1][r]); }
                book(g, pt_hi - 1, e0, e1, q_prev[g], flush);
            }
            if (flush) { ++sc_cur; sc_left = sc_pairs; } else --sc_left;
        }
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            if (sc_left != sc_pairs && h == 0)       // partial last super-chunk
                view[g].rec[sc_cur * view[g].rec_stride] = make_float2(static_cast<float>(s_sc[g]), q_prev[g]);
            if (d.use_cache && S == 1 && active[g] && h == 0) d.cache_resc[d.uid[slot[g]]] = static_cast<uint8_t>(min(n_resc[g], 255));
        }
        if (S == 1 && !d.sweep_only) {
#pragma unroll
            for (int g = 0; g < UG; ++g)
                search_and_emit<KH>(d, t, scr[g], scr_chunk[g], omu[g], Ahat[g], n_resc[g], active[g], pos[g], slot[g], j, h, true,
                                    delta_fixed[g], &view[g]);
        }
    }
}
#endif

// user groups per wave of the wide kernel (RECOGYM_F16W_UG: 1 = 8 waves x 32 users, 2 = 4 waves x 64 users)
inline int f16w_ug() {
    const char* e = getenv("RECOGYM_F16W_UG");
    return (e && e[0] == '2') ? 2 : 1;
}
#if RG_HAS(5)
draw_kernel_t f16w_kernel_for(const DevSim& d) {
    const int ug = f16w_ug();
#define RG_CASE(kh, a) if (d.KH == kh && d.N1 == a) return ug == 2 ? k_draw_f16w<kh, a, 2> : k_draw_f16w<kh, a, 1>;
    RG_CASE(16, 7) RG_CASE(32, 7) RG_CASE(32, 10) RG_CASE(32, 13)
#undef RG_CASE
    return nullptr;
}
#endif

// RG_POLICY_LOGREG_FROZEN for one user, computed by the whole wave: lane = class (c, c + 64, ...), so the
// coef_t rows of the viewed products are read as coalesced 512-byte runs instead of one gather per
// lane and class.  Same arithmetic as policy_act's scalar loop (per class: viewed products ascending,
// multiply then add, intercept last); the wave reduction keeps the smallest class index among equal
// maxima = numpy's first-maximum argmax.  `slot` must be wave-uniform.
__device__ uint32_t logreg_act_wave(const DevSim& d, uint32_t slot, int lane) {
    const hent_t* hr = hist_row(d, slot) + 1;             // entries after the header
    const uint32_t nd = h_cnt(hr[-1]);
    double best_s = -INFINITY;
    uint32_t best_c = 0xFFFFFFFFu;
    // four class blocks per pass and four history entries per batch: 16 independent loads in flight per
    // lane (one load per term on a dependent chain left this latency-bound); per class the terms are
    // still added in ascending product order
    for (uint32_t c0 = 0; c0 < d.lr_n; c0 += 256) {
        double sc[4] = {0.0, 0.0, 0.0, 0.0};
        uint32_t cc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cc[q] = min(c0 + 64u * q + lane, d.lr_n - 1);     // clamped: masked below
        for (uint32_t i0 = 0; i0 < nd; i0 += 4) {
            double w[4][4], cnt[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const hent_t x = hr[min(i0 + e, nd - 1)];
                cnt[e] = static_cast<double>(h_cnt(x));
                const double* row = d.lr_coef_t + static_cast<size_t>(h_prod(x)) * d.lr_n;
#pragma unroll
                for (int q = 0; q < 4; ++q) w[e][q] = row[cc[q]];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i0 + e < nd) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) sc[q] = __dadd_rn(sc[q], __dmul_rn(cnt[e], w[e][q]));
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t c = c0 + 64u * q + lane;
            if (c < d.lr_n) {
                const double v = __dadd_rn(sc[q], d.lr_intercept[c]);
                if (best_c == 0xFFFFFFFFu || v > best_s) { best_s = v; best_c = c; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double os = __shfl_xor(best_s, o);
        const uint32_t oc = __shfl_xor(best_c, o);
        if (oc != 0xFFFFFFFFu && (best_c == 0xFFFFFFFFu || os > best_s || (os == best_s && oc < best_c))) { best_s = os; best_c = oc; }
    }
    return static_cast<uint32_t>(d.lr_classes[best_c]);
}


// ------------------------------------------------------------------------------------------
// Frozen LogregMulticlassIps at scale (BASELINE config 5: 10^4 classes).  a = classes[argmax_c (b_c + sum_p views_p W[p][c])]
// depends on the view history only, so it is computed when the history has changed (5-6 times per user, not once per
// event) and kept per user:
//   k_logreg_select  (lane per live user) the users that need an act at this step — bandit users whose history changed
//                    since their last act, organic users that stop at this step (their phantom row) — into lr_list;
//   k_logreg_acts    (wave per listed user, lane = class) scores in fp32 from the fp32 copy of coef^T (half the bytes,
//                    twice the fma rate of the float64 walk): |s~_c - s_c| <= (nd + 3) 2^-24 (max|b| + sum_p views_p
//                    max_c |W[p][c]|) for every class, so when the best fp32 score leads the second best by more than
//                    twice that bound it IS sklearn's argmax; otherwise (near-ties, exact ties) the float64 walk in
//                    scipy's summation order (logreg_act_wave) decides — predict() bit for bit either way.
// ------------------------------------------------------------------------------------------
#if RG_HAS(6)
__global__ void __launch_bounds__(kBlock) k_logreg_select(DevSim d, uint32_t t) {
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC], n_b = d.step_cnt[2 * t + RG_STATE_BANDIT], n = n_o + n_b;
    const uint32_t* cur_o = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const uint32_t* cur_b = list_ptr(d, t & 1, RG_STATE_BANDIT);
    const uint32_t n_iter = (n + kBlock - 1) / kBlock;
    for (uint32_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
        const uint32_t i = it * kBlock + threadIdx.x;
        bool need = false;
        uint32_t slot = 0;
        if (i < n) {
            const bool is_org = i < n_o;
            slot = is_org ? cur_o[i] : cur_b[i - n_o];
            const uint32_t uidx = d.uid[slot];
            need = d.lr_dirty[uidx] != 0;
            if (need && is_org) {
                // an organic user needs an act only for its phantom row: when this step's transition stops it
                const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
                const rg_u32x4 w = rg_draw(d.seed, user, t, 0, RG_DRAW_EVENT);
                const double u_trans = rg_uniform(w.w[2], w.w[3]);
                const int ns = (d.cdf_o0 <= u_trans) + (d.cdf_o1 <= u_trans);
                need = ns == RG_STATE_STOP && !((d.first_user + uidx) < d.organic_only_below);
            }
        }
        const unsigned long long m = __ballot(need);
        uint32_t base = 0;
        if (m && lane_id() == 0) base = atomicAdd(&d.lr_cnt[t], static_cast<uint32_t>(__popcll(m)));
        base = __shfl(static_cast<int>(base), 0);
        if (need) d.lr_list[base + prefix_in_mask(m)] = slot;
    }
}
#endif

#if RG_HAS(6)
__global__ void __launch_bounds__(kBlock) k_logreg_acts(DevSim d, uint32_t t) {
    const int lane = lane_id();
    const uint32_t n = d.lr_cnt[t];
    const uint32_t waves_total = gridDim.x * (kBlock / 64);
    unsigned long long c_acts = 0, c_rows = 0, c_exact = 0;
    for (uint32_t w = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w < n; w += waves_total) {
        const uint32_t slot = d.lr_list[w];
        const uint32_t uidx = d.uid[slot];
        uint32_t action = 0;
        bool done = false;
        const hent_t* hr = hist_row(d, slot) + 1;
        const uint32_t nd = h_cnt(hr[-1]);
        c_acts += 1; c_rows += nd;
        if (d.lr_coef32_t && nd <= 32 && nd > 0) {
            // history entries in registers of the first nd lanes, broadcast by readlane
            const hent_t mine = static_cast<uint32_t>(lane) < nd ? hr[lane] : 0ull;
            float Ahat = d.lr_bmax;
            for (uint32_t i = 0; i < nd; ++i) {
                const hent_t x = __shfl(mine, static_cast<int>(i));
                Ahat = fmaf(static_cast<float>(h_cnt(x)), d.lr_wmax[h_prod(x)], Ahat);
            }
            float best = -INFINITY, second = -INFINITY;
            uint32_t best_c = 0;
            for (uint32_t c0 = 0; c0 < d.lr_n; c0 += 256) {
                float sc[4];
                uint32_t cc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cc[q] = min(c0 + 64u * q + lane, d.lr_n - 1);                  // clamped: masked below
                    sc[q] = d.lr_intercept32[cc[q]];
                }
                for (uint32_t i = 0; i < nd; ++i) {
                    const hent_t x = __shfl(mine, static_cast<int>(i));
                    const float cnt = static_cast<float>(h_cnt(x));
                    const float* row = d.lr_coef32_t + static_cast<size_t>(h_prod(x)) * d.lr_n;
#pragma unroll
                    for (int q = 0; q < 4; ++q) sc[q] = fmaf(cnt, row[cc[q]], sc[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t c = c0 + 64u * q + lane;
                    if (c < d.lr_n) {
                        if (sc[q] > best) { second = best; best = sc[q]; best_c = c; }
                        else if (sc[q] > second) second = sc[q];
                    }
                }
            }
            // wave top-2 over disjoint class sets: the best score with its class, and the best of everything else
            // (equal best scores leave a margin of 0: not certified, the float64 walk breaks the tie like numpy)
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o), os = __shfl_xor(second, o);
                const uint32_t oc = __shfl_xor(best_c, o);
                const float ns = fmaxf(fminf(best, ob), fmaxf(second, os));
                if (ob > best) best_c = oc;
                best = fmaxf(best, ob);
                second = ns;
            }
            const float bound = static_cast<float>(nd + 3) * 5.9604644775390625e-08f * Ahat * 1.01f;
            if (d.lr_n == 1 || best - second > 2.0f * bound) { action = static_cast<uint32_t>(d.lr_classes[best_c]); done = true; }
        }
        if (!done) { action = logreg_act_wave(d, slot, lane); c_exact += 1; }   // float64, scipy's summation order
        if (lane == 0) { d.lr_action[uidx] = action; d.lr_dirty[uidx] = 0; }
    }
    if (lane == 0 && c_acts) {
        atomicAdd(&d.counters[RG_CNT_LR_ACTS], c_acts);
        atomicAdd(&d.counters[RG_CNT_LR_ROWS], c_rows);
        if (c_exact) atomicAdd(&d.counters[RG_CNT_LR_EXACT], c_exact);
    }
}
#endif

// ------------------------------------------------------------------------------------------
// k_logreg_screen + k_logreg_decide — the frozen LogReg act by SCREEN AND REFINE (BASELINE config 5: 10^4 classes, where an
// act streams the coef^T rows of the user's viewed products: 40 KB per row in fp32, a 400 MB table that no cache holds).
//   screen   every class score in fp32 from an fp16 copy of coef^T (20 KB per row; the 200 MB table fits the Infinity
//            Cache): |s~_c - s_c| <= B for every class, B = sum_p views_p (2^-11 wmax_p + 2^-25)   (fp16 rounding, subnormals)
//                                                      + (nd + 3) 2^-24 (max|b| + sum_p views_p wmax_p)   (fp32 accumulation).
//            The argmax of the true scores is then among the CANDIDATES {c : s~_c >= max s~ - 2B}.  A step of the lock-step
//            loop has few acts (a few 10^3: about one per wave slot of the GPU), so its time is the latency of ONE act —
//            20 class blocks of 512, each a round trip for the rows — not throughput: the classes of an act are split
//            into kLrSplit RANGES, a wave per (act, range): it keeps the range's maximum and the classes within 2B of the
//            running maximum (a superset of the range's candidates; at most kLrCand survive the range's final maximum);
//   decide   a wave per act: the maximum over the ranges, the candidates within 2B of it (<= 64); one: certified; more (near-ties,
//            exact ties): their scores in float64 in scipy's csr_matvecs order (products ascending, multiply then add,
//            intercept last), a lane per candidate — nd scattered 8-byte reads each instead of a second pass over whole
//            rows; first maximum wins, like numpy's argmax.
// sklearn's predict() bit for bit, as before; a range with more than kLrCand candidates (degenerate models): the float64
// walk over all classes.  Needs n_classes % 8 == 0 (16-byte loads of 8 halves); the host keeps the fp32 kernel otherwise.
// ------------------------------------------------------------------------------------------
// 8 ranges x 8 candidates = the 64 lanes of the deciding wave.  Measured on config 5 (act kernels per LogReg-arm run):
// 1 range (one wave per act, 20 blocks in sequence) 390 ms, 8 ranges (three blocks each) 323 ms, 20 ranges of one block
// 388 ms (every wave pays the history read and the bound again): profiles/r3/ab_call5*, ab_call10*.
constexpr uint32_t kLrSplit = 8, kLrCand = 8;
// per act and range: {range maximum, candidates (0xFFFFFFFF: too many), 2B, -} then kLrCand x {class, score}
constexpr uint32_t kLrPartWords = 4 + 2 * kLrCand;
#if RG_HAS(6)
__global__ void __launch_bounds__(kBlock) k_logreg_screen(DevSim d, uint32_t t) {
    __shared__ uint32_t s_cand[kBlock / 64][32];
    __shared__ float s_cval[kBlock / 64][32];
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t n = d.lr_cnt[t];
    const uint32_t C = d.lr_n;
    const uint32_t RC = ((C + kLrSplit - 1) / kLrSplit + 7u) & ~7u;       // classes per range (a multiple of 8)
    const uint32_t waves_total = gridDim.x * (kBlock / 64);
    for (uint32_t item = blockIdx.x * (kBlock / 64) + wave; item < n * kLrSplit; item += waves_total) {
        const uint32_t w = item / kLrSplit, r = item % kLrSplit;
        const uint32_t slot = d.lr_list[w];
        const hent_t* hr = hist_row(d, slot) + 1;
        const uint32_t nd = h_cnt(hr[-1]);
        // ---- the error bound of this history ----
        float A = 0.0f, V = 0.0f;
        for (uint32_t i = lane; i < nd; i += 64) {
            const hent_t x = hr[i];
            const float cnt = static_cast<float>(h_cnt(x));
            A = fmaf(cnt, d.lr_wmax[h_prod(x)], A);
            V += cnt;
        }
        for (int o = 32; o > 0; o >>= 1) { A += __shfl_xor(A, o); V += __shfl_xor(V, o); }
        const float B = (A * 4.8828125e-4f + V * 2.98023224e-8f + static_cast<float>(nd + 3) * 5.9604644775390625e-08f * (d.lr_bmax + A)) * 1.02f;
        const float thr = 2.0f * B * 1.01f + 1e-30f;
        const uint32_t c_lo = r * RC, c_hi = min(c_lo + RC, C);
        float rb = -INFINITY;
        uint32_t n_cand = 0;
        bool overflow = false;
        for (uint32_t c0 = c_lo; c0 < c_hi && !overflow; c0 += 512) {
            const uint32_t c = c0 + 8u * static_cast<uint32_t>(lane);
            const bool in = c < c_hi;                                // (a lane's 8 classes are all in or all out)
            const uint32_t cl = in ? c : c_lo;
            float acc[8];
            {
                const float4 b0 = *reinterpret_cast<const float4*>(d.lr_intercept32 + cl);
                const float4 b1 = *reinterpret_cast<const float4*>(d.lr_intercept32 + cl + 4);
                acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
            }
            for (uint32_t i0 = 0; i0 < nd; i0 += 8) {                // eight rows in flight
                half8 hv[8];
                float cn[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const hent_t x = hr[min(i0 + e, nd - 1)];
                    cn[e] = i0 + e < nd ? static_cast<float>(h_cnt(x)) : 0.0f;
                    hv[e] = *reinterpret_cast<const half8*>(d.lr_coef16_t + static_cast<size_t>(h_prod(x)) * C + cl);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (i0 + e < nd) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(cn[e], static_cast<float>(hv[e][j]), acc[j]);
                    }
            }
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, acc[j]);
            if (!in) m = -INFINITY;
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            rb = fmaxf(rb, m);
            const float cut = rb - thr;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool pass = in && acc[j] >= cut;
                const unsigned long long pm = __ballot(pass);
                if (pm && !overflow) {
                    const uint32_t np = static_cast<uint32_t>(__popcll(pm));
                    if (n_cand + np > 32u) overflow = true;
                    else {
                        if (pass) { const uint32_t k = n_cand + prefix_in_mask(pm); s_cand[wave][k] = c + j; s_cval[wave][k] = acc[j]; }
                        n_cand += np;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- what survives the range's final maximum ----
        uint32_t* part = d.lr_part + (static_cast<size_t>(w) * kLrSplit + r) * kLrPartWords;
        const bool mine = !overflow && static_cast<uint32_t>(lane) < n_cand;
        const bool keep = mine && s_cval[wave][mine ? lane : 0] >= rb - thr;
        const unsigned long long km = __ballot(keep);
        uint32_t n_keep = static_cast<uint32_t>(__popcll(km));
        if (overflow || n_keep > kLrCand) n_keep = 0xFFFFFFFFu;
        else if (keep) {
            const uint32_t k = prefix_in_mask(km);
            part[4 + 2 * k] = s_cand[wave][lane];
            part[5 + 2 * k] = __builtin_bit_cast(uint32_t, s_cval[wave][lane]);
        }
        if (lane == 0) { part[0] = __builtin_bit_cast(uint32_t, rb); part[1] = n_keep; part[2] = __builtin_bit_cast(uint32_t, thr); part[3] = nd; }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void __launch_bounds__(kBlock) k_logreg_decide(DevSim d, uint32_t t) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t n = d.lr_cnt[t];
    const uint32_t C = d.lr_n;
    const uint32_t waves_total = gridDim.x * (kBlock / 64);
    unsigned long long c_acts = 0, c_rows = 0, c_exact = 0;
    for (uint32_t w = blockIdx.x * (kBlock / 64) + wave; w < n; w += waves_total) {
        const uint32_t slot = d.lr_list[w];
        const uint32_t uidx = d.uid[slot];
        const hent_t* hr = hist_row(d, slot) + 1;
        const uint32_t* part = d.lr_part + static_cast<size_t>(w) * kLrSplit * kLrPartWords;
        // lane = (range, candidate index); the last 64 - kLrSplit kLrCand lanes have no range
        const uint32_t r_raw = static_cast<uint32_t>(lane) / kLrCand, k = static_cast<uint32_t>(lane) % kLrCand;
        const bool has_r = r_raw < kLrSplit;
        const uint32_t r = has_r ? r_raw : 0u;
        const uint32_t* pr = part + r * kLrPartWords;
        const float rmax = has_r ? __builtin_bit_cast(float, pr[0]) : -INFINITY;
        const uint32_t nk = has_r ? pr[1] : 0u;
        const float thr = __builtin_bit_cast(float, part[2]);
        const uint32_t nd = part[3];
        c_acts += 1; c_rows += nd;
        float gmax = rmax;
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
        const bool overflow = __ballot(nk == 0xFFFFFFFFu) != 0ull;
        uint32_t action;
        if (overflow) { action = logreg_act_wave(d, slot, lane); c_exact += 1; }
        else {
            const bool have = k < nk;
            const uint32_t cc = have ? pr[4 + 2 * k] : 0u;
            const float cv = have ? __builtin_bit_cast(float, pr[5 + 2 * k]) : -INFINITY;
            const bool keep = have && cv >= gmax - thr;
            const unsigned long long km = __ballot(keep);
            uint32_t best_c = 0xFFFFFFFFu;
            if (__popcll(km) == 1) best_c = static_cast<uint32_t>(__shfl(static_cast<int>(cc), __builtin_ctzll(km)));
            else {
                double sc = -INFINITY;
                if (keep) {
                    sc = 0.0;
                    for (uint32_t i = 0; i < nd; ++i) {
                        const hent_t x = hr[i];
                        sc = __dadd_rn(sc, __dmul_rn(static_cast<double>(h_cnt(x)), d.lr_coef_t[static_cast<size_t>(h_prod(x)) * C + cc]));
                    }
                    sc = __dadd_rn(sc, d.lr_intercept[cc]);
                    best_c = cc;
                }
                for (int o = 32; o > 0; o >>= 1) {
                    const double os = __shfl_xor(sc, o);
                    const uint32_t oc = __shfl_xor(best_c, o);
                    if (oc != 0xFFFFFFFFu && (best_c == 0xFFFFFFFFu || os > sc || (os == sc && oc < best_c))) { sc = os; best_c = oc; }
                }
                c_exact += 1;
            }
            action = static_cast<uint32_t>(d.lr_classes[best_c]);
        }
        if (lane == 0) { d.lr_action[uidx] = action; d.lr_dirty[uidx] = 0; }
    }
    if (lane == 0 && c_acts) {
        atomicAdd(&d.counters[RG_CNT_LR_ACTS], c_acts);
        atomicAdd(&d.counters[RG_CNT_LR_ROWS], c_rows);
        if (c_exact) atomicAdd(&d.counters[RG_CNT_LR_EXACT], c_exact);
    }
}
#endif

// ------------------------------------------------------------------------------------------
// k_advance — one Markov transition for every live user (lane per user).
// ------------------------------------------------------------------------------------------
constexpr int kAdvBlock = 256;
#if RG_HAS(6)
__global__ void __launch_bounds__(kAdvBlock) k_advance(DevSim d, uint32_t t, const int32_t* actions) {
    constexpr int kSub = 1;                     // block iterations that share one reservation
    __shared__ uint32_t s_cnt_o[kSub][kAdvBlock / 64], s_cnt_b[kSub][kAdvBlock / 64], s_cnt_d[kSub][kAdvBlock / 64], s_base_o, s_base_b, s_base_d;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_b = d.step_cnt[2 * t + RG_STATE_BANDIT];
    const uint32_t n = n_o + n_b;
    const uint32_t* cur_o = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const uint32_t* cur_b = list_ptr(d, t & 1, RG_STATE_BANDIT);
    uint32_t* next_o = list_ptr(d, (t + 1) & 1, RG_STATE_ORGANIC);
    uint32_t* next_b = list_ptr(d, (t + 1) & 1, RG_STATE_BANDIT);
    uint32_t* next_cnt = d.step_cnt + 2 * (t + 1);
    const int wave = threadIdx.x >> 6, lane = lane_id();
    if (blockIdx.x == 0 && threadIdx.x == 0) d.log_base[t + 1] = d.log_base[t] + n;

    uint32_t clicks = 0, phantoms = 0;
    const uint32_t n_iter = (n + kSub * kAdvBlock - 1) / (kSub * kAdvBlock);
    for (uint32_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
      int ns_j[kSub];
      uint32_t slot_j[kSub];
      unsigned long long mo_j[kSub], mb_j[kSub], md_j[kSub];
      bool dr_j[kSub];
      double ds_j[kSub];
#pragma unroll
      for (int sub = 0; sub < kSub; ++sub) {
        const uint32_t i = (it * kSub + sub) * kAdvBlock + threadIdx.x;
        int ns = RG_STATE_STOP;       // inactive lanes look dead
        uint32_t slot = 0;
        bool drift_me = false;
        double drift_sig = 0.0;
        uint32_t lr_a = 0;            // RG_POLICY_LOGREG_FROZEN: this user's action for its current view history
        if (d.policy == RG_POLICY_LOGREG_FROZEN && i < n)
            // the policy reads only the view history: its act was computed by k_logreg_acts when the history last changed
            // and serves the bandit event and the phantom row alike
            lr_a = d.lr_action[d.uid[i < n_o ? cur_o[i] : cur_b[i - n_o]]];
        if (i < n) {
            const bool is_org = i < n_o;
            slot = is_org ? cur_o[i] : cur_b[i - n_o];
            const uint32_t uidx = d.uid[slot];
            const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
            const rg_u32x4 w = rg_draw(d.seed, user, t, 0, RG_DRAW_EVENT);
            const double u_trans = rg_uniform(w.w[2], w.w[3]);
            bool click = false;
            if (!is_org) {
                // 97 % of the bandit events cannot click whatever beta[a] . omega is (kNoClickBelow): they read neither row
                const double u_click = rg_uniform(w.w[0], w.w[1]);
                const bool need_ctr = d.aux_pclick != nullptr || !(u_click < kNoClickBelow);
                // Touch the two cache lines of the user's omega row (and the head of its view
                // history) NOW: they arrive while the policy draws and walks the history, instead
                // of costing another HBM round trip after it — this kernel is latency-bound.
                const double* om_row = d.omega + static_cast<size_t>(slot) * d.OMS;
                double touch0 = 0.0, touch1 = 0.0;
                if (need_ctr) { touch0 = om_row[0]; touch1 = om_row[d.K - 1]; }
                // K even and <= 24 (rows are 16-byte aligned): the whole omega row is fetched here as 16-byte
                // loads and held across the policy, so that only beta's row is left on the critical path
                const bool pre = d.K <= 24 && !(d.K & 1);
                double2 wpre[12];
                if (pre && need_ctr) {
#pragma unroll
                    for (int k2 = 0; k2 < 12; ++k2)
                        wpre[k2] = *reinterpret_cast<const double2*>(om_row + 2 * min(static_cast<uint32_t>(k2), d.K / 2 - 1));
                }
                uint32_t touch2 = 0;
                if (d.hist_cap) touch2 = static_cast<uint32_t>(d.hist[static_cast<size_t>(slot) * d.hist_cap]);
                // step_offline: the policy acts (abstract.py:202-221), then draw_click
                double ps;
                uint32_t a;
                if (d.policy == RG_POLICY_EXTERNAL) { a = static_cast<uint32_t>(actions[uidx]); ps = __builtin_nan(""); }
                else if (d.policy == RG_POLICY_LOGREG_FROZEN) { a = lr_a; ps = 1.0; }
                else a = policy_act(d, slot, user, t, &ps);
                // beta[a] . omega, k ascending (the oracle's association); loads are issued eight
                // k at a time — a plain loop leaves one HBM round trip per k on the critical path
                const double* b = d.beta + static_cast<size_t>(a) * d.K;
                const double* om = d.omega + static_cast<size_t>(slot) * d.OMS;
                double x = 0.0;
                if (!need_ctr) {}
                else if (pre) {
                    double2 bpre[12];
#pragma unroll
                    for (int k2 = 0; k2 < 12; ++k2)
                        bpre[k2] = *reinterpret_cast<const double2*>(b + 2 * min(static_cast<uint32_t>(k2), d.K / 2 - 1));
#pragma unroll
                    for (int k2 = 0; k2 < 12; ++k2)
                        if (static_cast<uint32_t>(2 * k2) < d.K) { x += bpre[k2].x * wpre[k2].x; x += bpre[k2].y * wpre[k2].y; }
                } else
                for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {
                    double wv[8], bv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t k = min(k0 + i, d.K - 1);
                        wv[i] = om[k];
                        bv[i] = b[k];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (k0 + i < d.K) x += bv[i] * wv[i];
                }
                asm volatile("" ::"v"(touch0), "v"(touch1), "v"(touch2));   // keeps the early loads alive
                double ctr = 0.0;
                if (need_ctr) {
                    ctr = ff64(x + d.mu_b[a]);
                    const double p0 = 1.0 - ctr;
                    click = (p0 / (p0 + ctr)) <= u_click;
                }
                clicks += click;
                const uint64_t row = d.log_base[t] + i;
                if (d.log && row < d.log_cap) {
                    rg_event e;
                    e.u = user; e.t = t;
                    e.code = RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a;
                    e.ps = static_cast<float>(ps);
                    d.log[row] = e;
                    if (d.aux_ps) d.aux_ps[row] = ps;
                    if (d.aux_pclick) d.aux_pclick[row] = ctr;
                    if (d.aux_time) d.aux_time[row] = d.utime[uidx];
                }
            }
            // update_state (reco_env_v1.py:85-100)
            const double c0 = is_org ? d.cdf_o0 : d.cdf_b0, c1 = is_org ? d.cdf_o1 : d.cdf_b1;
            ns = (c0 <= u_trans) + (c1 <= u_trans);
            // NormalTimeGenerator: the clock advances by |mu + sigma z| (normal_time_generator.py:25) and the drift's
            // standard deviation is scaled by that time delta (1 when it is exactly 0; reco_env_v1.py:91-92)
            double omega_k = 1.0;
            if (d.time_mode) {
                double z0, z1;
                normal_pair(d.seed, user, t, 0, RG_DRAW_TIME, &z0, &z1);
                const double dt = fabs(d.time_mu + d.time_sigma * z0);
                d.utime[uidx] = d.utime[uidx] + dt;
                omega_k = dt == 0.0 ? 1.0 : dt;
            }
            // omega drifts when the DRAWN next state is organic (reco_env_v1.py:95-98; the click override below does not
            // redraw it): listed for k_drift, which runs right behind this kernel
            drift_me = d.sigma_omega != 0.0 && (d.change_omega_for_bandits || ns == RG_STATE_ORGANIC);
            drift_sig = d.sigma_omega * omega_k;
            if (click) ns = RG_STATE_ORGANIC;          // abstract.py:180-181
            const bool organic_only = (d.first_user + uidx) < d.organic_only_below;
            if (organic_only && ns != RG_STATE_ORGANIC) {
                ns = RG_STATE_STOP;                    // warm-up users end with their first session
                d.n_events[uidx] = t + 1;
            } else if (ns == RG_STATE_STOP) {
                d.n_events[uidx] = t + 1;
                if (d.policy != RG_POLICY_EXTERNAL) {
                    // final step_offline(done=True): one more act, reward 0 (abstract.py:223-233,311-316)
                    double ps = 1.0;
                    const uint32_t a = d.policy == RG_POLICY_LOGREG_FROZEN ? lr_a : policy_act(d, slot, user, t + 1, &ps);
                    rg_event e;
                    e.u = user; e.t = t + 1; e.code = RG_EV_BANDIT | RG_EV_PHANTOM | a;
                    e.ps = static_cast<float>(ps);
                    d.phantom[uidx] = e;
                    d.phantom_ps[uidx] = ps;
                    if (d.time_mode) d.phantom_time[uidx] = d.utime[uidx];       // (already advanced past the last event)
                    d.has_phantom[uidx] = 1;
                    phantoms += 1;
                }
            }
        }
        ns_j[sub] = ns; slot_j[sub] = slot;
        mo_j[sub] = __ballot(ns == RG_STATE_ORGANIC);
        mb_j[sub] = __ballot(ns == RG_STATE_BANDIT);
        md_j[sub] = __ballot(drift_me);
        dr_j[sub] = drift_me; ds_j[sub] = drift_sig;
        if (lane == 0) { s_cnt_o[sub][wave] = __popcll(mo_j[sub]); s_cnt_b[sub][wave] = __popcll(mb_j[sub]); s_cnt_d[sub][wave] = __popcll(md_j[sub]); }
      }
        // Ordered compaction of the survivors into next step's lists: ballot + mbcnt inside the
        // wave, one returning 64-bit atomic per block iteration reserves room in both lists.
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t to = 0, tb = 0, td = 0;
#pragma unroll
            for (int sub = 0; sub < kSub; ++sub)
#pragma unroll
                for (int w2 = 0; w2 < kAdvBlock / 64; ++w2) { to += s_cnt_o[sub][w2]; tb += s_cnt_b[sub][w2]; td += s_cnt_d[sub][w2]; }
            s_base_d = td ? atomicAdd(&d.drift_cnt[t], td) : 0u;
            // step_cnt[t+1] = {organic, bandit} is an aligned u32 pair: reserve both lists at once
            unsigned long long base = 0;
            if (to | tb)
                base = atomicAdd(reinterpret_cast<unsigned long long*>(next_cnt),
                                 static_cast<unsigned long long>(to) | (static_cast<unsigned long long>(tb) << 32));
            s_base_o = static_cast<uint32_t>(base);
            s_base_b = static_cast<uint32_t>(base >> 32);
        }
        __syncthreads();
        uint32_t off_o = s_base_o, off_b = s_base_b, off_d = s_base_d;
#pragma unroll
        for (int sub = 0; sub < kSub; ++sub) {
            uint32_t wo = off_o, wb = off_b, wd = off_d;
            for (int w2 = 0; w2 < wave; ++w2) { wo += s_cnt_o[sub][w2]; wb += s_cnt_b[sub][w2]; wd += s_cnt_d[sub][w2]; }
            if (ns_j[sub] == RG_STATE_ORGANIC) next_o[wo + prefix_in_mask(mo_j[sub])] = slot_j[sub];
            if (ns_j[sub] == RG_STATE_BANDIT) next_b[wb + prefix_in_mask(mb_j[sub])] = slot_j[sub];
            if (dr_j[sub]) {
                const uint32_t e = wd + prefix_in_mask(md_j[sub]);
                d.drift_list[e] = slot_j[sub];
                if (d.time_mode) d.drift_sig[e] = ds_j[sub];
            }
#pragma unroll
            for (int w2 = 0; w2 < kAdvBlock / 64; ++w2) { off_o += s_cnt_o[sub][w2]; off_b += s_cnt_b[sub][w2]; off_d += s_cnt_d[sub][w2]; }
        }
        __syncthreads();
    }
    // counters: one atomic per wave per kernel
    for (int o = 32; o > 0; o >>= 1) { clicks += __shfl_xor(clicks, o); phantoms += __shfl_xor(phantoms, o); }
    if (lane == 0) {
        if (clicks) atomicAdd(&d.counters[RG_CNT_CLICKS], static_cast<unsigned long long>(clicks));
        if (phantoms) atomicAdd(&d.counters[RG_CNT_PHANTOM], static_cast<unsigned long long>(phantoms));
    }
}
#endif

// k_drift — omega <- omega + sigma_omega (time delta) Z(K) (reco_env_v1.py:95-98) of the users k_advance listed at step t: a lane per
// (user, Box-Muller pair), the K normals addressed by (user, t, pair) as everywhere else.
#if RG_HAS(6)
__global__ void __launch_bounds__(kBlock) k_drift(DevSim d, uint32_t t) {
    const uint32_t n = d.drift_cnt[t];
    const uint32_t KP = (d.K + 1) / 2;
    const uint64_t items = static_cast<uint64_t>(n) * KP;
    for (uint64_t it = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; it < items; it += static_cast<uint64_t>(gridDim.x) * kBlock) {
        const uint32_t e = static_cast<uint32_t>(it / KP), j = static_cast<uint32_t>(it % KP);
        const uint32_t slot = d.drift_list[e];
        const uint32_t user = static_cast<uint32_t>(d.first_user + d.uid[slot]);
        const double sig = d.time_mode ? d.drift_sig[e] : d.sigma_omega;
        double z0, z1;
        normal_pair(d.seed, user, t, j, RG_DRAW_DRIFT, &z0, &z1);
        double* o0 = d.omega + static_cast<size_t>(slot) * d.OMS + 2 * j;
        *o0 = *o0 + sig * z0;
        if (2 * j + 1 < d.K) { double* o1 = o0 + 1; *o1 = *o1 + sig * z1; }
    }
}
search_kernel_t drift_kernel() { return k_drift; }
#endif

// ------------------------------------------------------------------------------------------
// k_tail — the end of a run, user by user instead of step by step.
//
// Once few users are left (10 M users: ~1 300 of the ~1 800 lock-step steps serve < 1 % of the
// events) a lock-step step costs its launch/latency floor (~110 us) whatever the population.
// Trajectories are independent, so the remaining users are handed to this kernel instead: a
// block takes a user (ticket counter) and walks it to its end — the organic draws in float64
// across the block (the arithmetic of k_exact_*: lane per product, 64-product chunk sums, prefix
// search), the click / transition / policy / history work of k_advance on thread 0.  Rows go to
// log rows log_base[t0] + ticket (the sorted log does not depend on raw positions); events of
// steps > t0 are counted in the kCntTail* counters (step t0's are in step_cnt[t0]).
// ------------------------------------------------------------------------------------------

#if RG_HAS(6)
__global__ void __launch_bounds__(kBlock) k_tail(DevSim d, uint32_t t0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* om = reinterpret_cast<double*>(smem_raw);                       // [K rounded up to 2]
    double* csum = om + ((d.K + 1) & ~1u);                                   // [n_chunks rounded up to 4]
    __shared__ uint32_t s_next, s_v;
    __shared__ int s_state, s_drift;
    __shared__ double s_max[kBlock / 64];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t n_o = d.step_cnt[2 * t0 + RG_STATE_ORGANIC], n_b = d.step_cnt[2 * t0 + RG_STATE_BANDIT];
    const uint32_t n = n_o + n_b;
    const uint32_t n_chunks = d.PT / 64;
    const uint32_t* cur_o = list_ptr(d, t0 & 1, RG_STATE_ORGANIC);
    const uint32_t* cur_b = list_ptr(d, t0 & 1, RG_STATE_BANDIT);
    unsigned long long c_org = 0, c_ban = 0, c_clicks = 0, c_ph = 0;          // thread 0 only
    uint32_t c_maxt = 0;

    // block maximum of the logits (pass 0) or chunk sums of exp(l - ref) into csum + that maximum
    auto sweep = [&](bool sums, double ref) -> double {
        double wmax = -INFINITY;
        for (uint32_t g = wave; g * 4 < n_chunks; g += kBlock / 64) {
            double l[4];
            logit64x4(d, om, g * 256 + lane, l);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                wmax = fmax(wmax, l[u]);
                if (sums) {
                    const double sm = wave_sum(exp64(l[u] - ref));
                    if (lane == 0) csum[g * 4 + u] = sm;                     // chunks past P: every logit -inf -> 0
                }
            }
        }
        wmax = wave_max(wmax);
        __syncthreads();                       // s_max of the previous sweep has been read
        if (lane == 0) s_max[wave] = wmax;
        __syncthreads();
        double m = s_max[0];
        for (int w2 = 1; w2 < kBlock / 64; ++w2) m = fmax(m, s_max[w2]);
        return m;
    };

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_next = static_cast<uint32_t>(atomicAdd(&d.counters[kCntTailTicket], 1ull));
        __syncthreads();
        const uint32_t i = s_next;
        if (i >= n) break;
        const uint32_t slot = i < n_o ? cur_o[i] : cur_b[i - n_o];
        int state = i < n_o ? RG_STATE_ORGANIC : RG_STATE_BANDIT;
        const uint32_t uidx = d.uid[slot];
        const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
        for (uint32_t k = threadIdx.x; k < d.K; k += kBlock) om[k] = d.omega[static_cast<size_t>(slot) * d.OMS + k];
        bool have_ref = false;
        double Mref = 0.0;
        __syncthreads();
        for (uint32_t t = t0;; ++t) {
            const rg_u32x4 w = rg_draw(d.seed, user, t, 0, RG_DRAW_EVENT);
            if (state == RG_STATE_ORGANIC) {
                // ---- organic product draw, float64 across the block ----
                if (!have_ref) { Mref = sweep(false, 0.0); have_ref = true; }
                double m = sweep(true, Mref);
                // any shift near the maximum gives the same decisions (1e-16 level); if omega drifted
                // the kept reference far from it, take the sums again with the fresh one
                if (!(fabs(m - Mref) <= 400.0)) { Mref = m; m = sweep(true, Mref); }
                if (wave == 0) {
                    double total = 0.0;
                    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64) {
                        const uint32_t c = c0 + lane;
                        total += __shfl(wave_scan(c < n_chunks ? csum[c] : 0.0, lane), 63);
                    }
                    const double target = rg_uniform(w.w[0], w.w[1]) * total;
                    uint32_t cstar = n_chunks - 1;
                    double before = 0.0, run = 0.0;
                    bool found = false;
                    for (uint32_t c0 = 0; c0 < n_chunks && !found; c0 += 64) {
                        const uint32_t c = c0 + lane;
                        const double x = c < n_chunks ? csum[c] : 0.0;
                        const double incl = wave_scan(x, lane);
                        const unsigned long long hit = __ballot(c < n_chunks && run + incl > target);
                        if (hit) {
                            const int L = __builtin_ctzll(hit);
                            cstar = c0 + L;
                            before = run + __shfl(incl - x, L);
                            found = true;
                        } else run += __shfl(incl, 63);
                    }
                    if (!found) before = run - csum[n_chunks - 1];
                    const uint32_t p = cstar * 64 + lane;
                    double lg = 0.0;
                    const double* g = d.gammaT + p;
                    for (uint32_t k = 0; k < d.K; ++k) lg += g[static_cast<size_t>(k) * d.PT] * om[k];
                    lg = p < d.P ? lg + d.mu_o[p] : -INFINITY;
                    const double incl = wave_scan(exp64(lg - Mref), lane);
                    const unsigned long long hit = __ballot(p < d.P && before + incl > target);
                    const uint32_t v = hit ? cstar * 64 + static_cast<uint32_t>(__builtin_ctzll(hit))
                                           : min(cstar * 64 + 63, d.P - 1);
                    if (lane == 0) s_v = v;
                }
                Mref = m;                                  // reference of this user's next draw
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                if (t > t0) { if (state == RG_STATE_ORGANIC) c_org += 1; else c_ban += 1; }
                const double u_trans = rg_uniform(w.w[2], w.w[3]);
                bool click = false;
                if (state == RG_STATE_ORGANIC) {
                    const uint32_t v = s_v;
                    const uint64_t row = d.log_base[t0] + atomicAdd(&d.counters[kCntTailRows], 1ull);
                    if (d.log && row < d.log_cap) {
                        rg_event e;
                        e.u = user; e.t = t; e.code = v; e.ps = __builtin_nanf("");
                        d.log[row] = e;
                    }
                    if (d.lpv) d.lpv[slot] = v;
                    if (d.hist_cap) history_add(d, slot, v);
                } else {
                    double ps;
                    const uint32_t a = policy_act(d, slot, user, t, &ps);
                    double ctr = 0.0;
                    click = false;
                    if (d.aux_pclick || !(rg_uniform(w.w[0], w.w[1]) < kNoClickBelow)) {
                        const double* b = d.beta + static_cast<size_t>(a) * d.K;
                        double x = 0.0;
                        for (uint32_t k = 0; k < d.K; ++k) x += b[k] * om[k];
                        ctr = ff64(x + d.mu_b[a]);
                        const double p0 = 1.0 - ctr;
                        click = (p0 / (p0 + ctr)) <= rg_uniform(w.w[0], w.w[1]);
                    }
                    c_clicks += click;
                    const uint64_t row = d.log_base[t0] + atomicAdd(&d.counters[kCntTailRows], 1ull);
                    if (d.log && row < d.log_cap) {
                        rg_event e;
                        e.u = user; e.t = t;
                        e.code = RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a;
                        e.ps = static_cast<float>(ps);
                        d.log[row] = e;
                        if (d.aux_ps) d.aux_ps[row] = ps;
                        if (d.aux_pclick) d.aux_pclick[row] = ctr;
                    }
                }
                const double c0 = state == RG_STATE_ORGANIC ? d.cdf_o0 : d.cdf_b0;
                const double c1 = state == RG_STATE_ORGANIC ? d.cdf_o1 : d.cdf_b1;
                int ns = (c0 <= u_trans) + (c1 <= u_trans);
                s_drift = d.sigma_omega != 0.0 && (d.change_omega_for_bandits || ns == RG_STATE_ORGANIC);
                if (click) ns = RG_STATE_ORGANIC;
                const bool organic_only = (d.first_user + uidx) < d.organic_only_below;
                if (organic_only && ns != RG_STATE_ORGANIC) {
                    ns = RG_STATE_STOP;
                    d.n_events[uidx] = t + 1;
                } else if (ns == RG_STATE_STOP) {
                    d.n_events[uidx] = t + 1;
                    double ps;
                    const uint32_t a = policy_act(d, slot, user, t + 1, &ps);
                    rg_event e;
                    e.u = user; e.t = t + 1; e.code = RG_EV_BANDIT | RG_EV_PHANTOM | a;
                    e.ps = static_cast<float>(ps);
                    d.phantom[uidx] = e;
                    d.phantom_ps[uidx] = ps;
                    d.has_phantom[uidx] = 1;
                    c_ph += 1;
                } else if (t + 2 >= kMaxSteps) {
                    ns = RG_STATE_STOP;                    // same bound as the lock-step loop; reported by the host
                    d.n_events[uidx] = t + 1;
                    atomicAdd(&d.counters[kCntTailLimit], 1ull);
                }
                if (ns == RG_STATE_STOP) c_maxt = max(c_maxt, t + 1);
                s_state = ns;
            }
            __syncthreads();
            state = s_state;
            if (s_drift && state != RG_STATE_STOP) {
                // omega drift of this step (k_advance applies it before the click override, which
                // only changes the state) — pair j by thread j
                for (uint32_t j = threadIdx.x; 2 * j < d.K; j += kBlock) {
                    double z0, z1;
                    normal_pair(d.seed, user, t, j, RG_DRAW_DRIFT, &z0, &z1);
                    om[2 * j] = om[2 * j] + d.sigma_omega * z0;
                    if (2 * j + 1 < d.K) om[2 * j + 1] = om[2 * j + 1] + d.sigma_omega * z1;
                }
            }
            __syncthreads();
            if (state == RG_STATE_STOP) break;
        }
    }
    if (threadIdx.x == 0) {
        if (c_org) atomicAdd(&d.counters[kCntTailOrganic], c_org);
        if (c_ban) atomicAdd(&d.counters[kCntTailBandit], c_ban);
        if (c_clicks) atomicAdd(&d.counters[RG_CNT_CLICKS], c_clicks);
        if (c_ph) atomicAdd(&d.counters[RG_CNT_PHANTOM], c_ph);
        if (c_maxt) atomicMax(&d.counters[kCntTailMaxT], static_cast<unsigned long long>(c_maxt));
    }
}
search_kernel_t logreg_select_kernel() { return k_logreg_select; }
search_kernel_t logreg_acts_kernel() { return k_logreg_acts; }
search_kernel_t logreg_screen_kernel() { return k_logreg_screen; }
search_kernel_t logreg_decide_kernel() { return k_logreg_decide; }
advance_kernel_t advance_kernel() { return k_advance; }
search_kernel_t tail_kernel() { return k_tail; }
#endif

// closes the books of the tail: step t0 + 1 exists, is empty, and starts after the tail's rows
#if RG_HAS(1)
__global__ void k_tail_finish(DevSim d, uint32_t t0) {
    d.log_base[t0 + 1] = d.log_base[t0] + d.counters[kCntTailRows];
    d.step_cnt[2 * (t0 + 1)] = 0;
    d.step_cnt[2 * (t0 + 1) + 1] = 0;
}
#endif


// ------------------------------------------------------------------------------------------
// k_walk — sigma_omega == 0: the whole run user-major instead of step-major.
//
// With omega fixed, nothing a user does depends on any other user or on a shared product sweep: after the
// one batched sweep that fills the per-user cache (k_draw_bf16p at t = 0) a trajectory is a chain of
// cached draws (k_draw_cached's arithmetic), policy acts, click draws and transitions (k_advance's
// arithmetic) addressed by (user, t).  So a lane takes a user and walks it to its end, and takes the next
// user from the queue when it stops: no live lists, no compaction, no repack, no per-step launches (the
// lock-step form spent ~200 us of launch/latency floor per step on ~800 steps), and the ~260 k users in
// flight (omega, cache row, view history: < 1 KB each) stay in the Infinity Cache instead of being
// re-gathered from HBM every step.  Per-lane times differ (a refilled lane starts at t = 0): every draw is
// addressed, rows carry (u, t), and rg_sim_sort_log orders them.
//
// Draws the certificate rejects need the user's float64 sums.  A lane cannot take them alone, and a
// wave-wide sweep per such draw is 3x less efficient than the user-per-lane kernel, so the user is PARKED
// (appended to park_list with its time) and its lane refilled; after round 1, k_exact_sums_u takes the
// sums of all parked users in one batch and round 2 walks them to their end — the parked draw and any
// later uncertified draw of theirs are float64 picks from the stored sums (exact_pick_wave), inline.
//
// Raw log: a wave reserves rows in chunks (one atomic per `chunk_rows` rows, not per row or per step) and
// marks the entries it does not use (kHoleCode); the sort skips them.
// ------------------------------------------------------------------------------------------
// users a lane of k_walk holds at a time
#ifndef RG_WALK_USERS
#define RG_WALK_USERS 1
#endif
constexpr int kWalkUsers = RG_WALK_USERS;
// LDS of one wave of k_walk: omega32 of its 2 x 64 users [entry][2 KH][64] + the mailbox + the rank table
__host__ __device__ inline size_t walk_wave_lds(uint32_t KH) { return static_cast<size_t>(kWalkUsers) * 2 * KH * 64 * 4 + 64 * 24 + 64 * 4; }

#if RG_HAS(7)
template <int KH, int OCC, bool DENSE>
__global__ void __launch_bounds__(kBlock, OCC) k_walk(DevSim d_arg, uint32_t n_work, int round, uint32_t chunk_rows,
                                                       uint32_t in_base, uint32_t out_base) {
    // The ~60 fields of DevSim this kernel uses do not fit the scalar registers next to its own state: kept live across
    // the loop they were spilled into VGPR lanes (v_writelane / v_readlane: ~10 % of the kernel's VALU instructions, the
    // unit that bounds it).  They are read from the kernel-argument segment instead — scalar loads, at the point of use:
    // the pointer is laundered once per iteration so that the loads are not hoisted out of the loop again.
    (void)d_arg;
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    const DevSim& d = *(const DevSim*)kargs;
    constexpr int K2 = 2 * KH;
    constexpr int kEmpty = 3;
    // a user that stops still owes its phantom row (one more policy act, abstract.py:311-316): it takes it on its lane's
    // NEXT step, through the one policy_act call site of the loop (a second inlined copy of the policy cost ~15 % of
    // the kernel's instructions and was executed on half of the steps)
    constexpr int kPhantom = 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    // A lane holds kWalkUsers users and, each iteration, advances the first of them that is in the state the wave
    // processes (see below): with one user per lane ~45 % of the lanes had nothing to do in an iteration.
    constexpr int kIdle = 5;                                       // this iteration: none of the lane's users takes part
    // per wave: omega32 of the lanes' users [entry][K2][64] (k-major: conflict-free; a lane keeps a user to its end, so
    // omega is fetched once per USER; the recompute and the fp32 click decision read it) | mailbox [64] {idx, A, B}
    char* wbase = smem_raw + static_cast<size_t>(wave) * walk_wave_lds(KH);
    float* om32 = reinterpret_cast<float*>(wbase);
    double* mbox = reinterpret_cast<double*>(om32 + kWalkUsers * K2 * 64);   // [64][3]
    uint32_t* slots = reinterpret_cast<uint32_t*>(mbox + 64 * 3);  // [64] lanes of the searching users, by rank
    const uint32_t n_cc = d.PT / 64;
    uint32_t slotA[kWalkUsers], tA[kWalkUsers];
    int stA[kWalkUsers];
    bool pendA[kWalkUsers];                                        // round 2: the parked draw, to be picked in float64
#pragma unroll
    for (int e = 0; e < kWalkUsers; ++e) { slotA[e] = 0; tA[e] = 0; stA[e] = kEmpty; pendA[e] = false; }
    uint32_t res_next = 0, res_end = 0;                            // this wave's reservoir of queue tickets
    uint64_t row_next = 0, row_end = 0;                            // this wave's reserved raw-log rows
    uint32_t park_next = 0, park_end = 0;                          // this wave's reserved park_list entries
    bool exhausted = false;
    unsigned long long c_org = 0, c_ban = 0, c_clicks = 0, c_ph = 0, c_pick = 0, c_sweeps = 0;
    uint32_t c_maxt = 0, c_limit = 0;

    for (;;) {
        asm volatile("" : "+s"(kargs));
        const DevSim& d = *(const DevSim*)kargs;
        // ---- refill the entries whose user has stopped (or was parked) ----
#pragma unroll
        for (int e = 0; e < kWalkUsers; ++e) {
        unsigned long long dead = __ballot(stA[e] == kEmpty);
        if (dead && !exhausted && (static_cast<uint32_t>(__popcll(dead)) >= d.walk_refill || dead == ~0ull)) {
            for (int pass = 0; pass < 2 && dead; ++pass) {
                if (res_next == res_end) {
                    if (exhausted) break;
                    uint32_t base = 0;
                    if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_ticket, 64ull));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= n_work) { exhausted = true; break; }
                    res_next = base; res_end = min(base + 64u, n_work);
                }
                const uint32_t take = min(static_cast<uint32_t>(__popcll(dead)), res_end - res_next);
                const uint32_t r = prefix_in_mask(dead);
                const bool mine = ((dead >> lane) & 1ull) != 0 && r < take;   // (a lane that drew an unused entry in pass 1 is not in `dead`)
                if (mine) {
                    const uint32_t idx = res_next + r;
                    uint32_t s2 = idx;
                    if (round >= 2) s2 = d.park_list[in_base + idx];
                    if (s2 != 0xFFFFFFFFu) {
                        slotA[e] = s2;
                        stA[e] = RG_STATE_ORGANIC;                   // every user starts organic
                        tA[e] = 0u;
                        pendA[e] = false;
                        if (round >= 2) {
                            // a parked user sits at an organic draw to be picked in float64; a handed-over one anywhere
                            const uint32_t pt = d.park_t[s2];
                            tA[e] = pt & 0xFFFFFFu; stA[e] = static_cast<int>((pt >> 24) & 7u); pendA[e] = (pt >> 27) & 1u;
                            if (round == 2) { d.f64_valid[s2] = 1; c_sweeps += pendA[e] ? 1 : 0; }   // the batch between the rounds took its sums (counted for the parked)
                        }
                        // omega32 = float(omega), as k_cache_finalize left it in the user's cache row (floats 44 .. 44 + K2)
                        const float4* rp = reinterpret_cast<const float4*>(d.cache_row + static_cast<size_t>(s2) * d.cache_row_f);
                        float* o = om32 + e * (K2 * 64) + lane;
#pragma unroll
                        for (int k4 = 0; k4 < K2 / 4; ++k4) {
                            const float4 x = rp[11 + k4];
                            o[(4 * k4) * 64] = x.x; o[(4 * k4 + 1) * 64] = x.y; o[(4 * k4 + 2) * 64] = x.z; o[(4 * k4 + 3) * 64] = x.w;
                        }
#pragma unroll
                        for (int k = (K2 / 4) * 4; k < K2; ++k) o[k * 64] = reinterpret_cast<const float*>(rp)[44 + k];
                    }
                }
                res_next += take;
                dead = __ballot(stA[e] == kEmpty && !mine);          // lanes that drew an unused entry wait for the next refill
            }
        }
        }
        // (sorting the block's users by state through LDS so that waves are all-organic or all-bandit was measured
        // SLOWER, 352 vs 301 ms on C3: its two barriers per step serialise the block on its organic wave's latency chain)
        bool any_user = false, has_org = false, has_ban = false;
#pragma unroll
        for (int e = 0; e < kWalkUsers; ++e) {
            any_user = any_user || stA[e] != kEmpty;
            has_org = has_org || stA[e] == RG_STATE_ORGANIC;
            has_ban = has_ban || stA[e] == RG_STATE_BANDIT || stA[e] == kPhantom;
        }
        const unsigned long long live = __ballot(any_user);
        if (!live) { if (exhausted) break; else continue; }
        // ---- hand-over: once the queue is empty a wave would drain for several user lifetimes with ever fewer live
        // lanes (an iteration costs the same whatever their number).  With few left it passes its users on — they are
        // appended to the list the next round reads, with their time, state and pending-pick flag — and ends; the
        // last round walks everyone to the end. ----
        if (exhausted && round < 3 && d.walk_handover && static_cast<uint32_t>(__popcll(live)) <= d.walk_handover) {
#pragma unroll
            for (int e = 0; e < kWalkUsers; ++e) {
                const bool give = stA[e] != kEmpty;
                const unsigned long long gm = __ballot(give);
                if (!gm) continue;
                const uint32_t np = static_cast<uint32_t>(__popcll(gm));
                if (park_next + np > park_end) {
                    for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
                    uint32_t base = 0;
                    if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_park, 64ull));
                    base = __builtin_amdgcn_readfirstlane(base);
                    park_next = base; park_end = base + 64;
                }
                if (give) {
                    d.park_list[out_base + park_next + prefix_in_mask(gm)] = slotA[e];
                    d.park_t[slotA[e]] = tA[e] | (static_cast<uint32_t>(stA[e]) << 24) | (pendA[e] ? 1u << 27 : 0u);
                    // round 1: the float64 batch takes the sums of every listed user with the reference found here — the
                    // user's common reference, as a parked draw would have left it (float 32 of its cache row)
                    if (round == 1) d.exact_ref[slotA[e]] = d.cache_row[static_cast<size_t>(slotA[e]) * d.cache_row_f + 32];
                    stA[e] = kEmpty;
                }
                park_next += np;
            }
            break;
        }
        {
        // ---- ONE kind of event per iteration: the organic draw and the bandit event are different code, and a wave whose
        // lanes are in both states executes both for every step at ~25 active lanes each.  The users are independent and
        // every draw is addressed by (user, t), so the lanes in the minority state simply wait an iteration: the wave runs
        // the path more of its lanes are ready for (organic weighted by walk_bias / 4: its path is the longer one). ----
        bool run_org = true, run_ban = true;
        if (d.walk_bias) {
            const uint32_t n_ro = static_cast<uint32_t>(__popcll(__ballot(has_org)));
            const uint32_t n_rb = static_cast<uint32_t>(__popcll(__ballot(has_ban)));
            run_org = n_ro != 0 && n_ro * d.walk_bias >= n_rb * 4u;
            run_ban = !run_org;
        }
        // this iteration's user of the lane: its first one in a state that is processed
        int sel = -1;
#pragma unroll
        for (int e = kWalkUsers - 1; e >= 0; --e)
            if ((run_org && stA[e] == RG_STATE_ORGANIC) || (run_ban && (stA[e] == RG_STATE_BANDIT || stA[e] == kPhantom))) sel = e;
        uint32_t slot = slotA[0], t = tA[0];
        int st = sel == 0 ? stA[0] : kIdle;
        bool pending = pendA[0];
#pragma unroll
        for (int e = 1; e < kWalkUsers; ++e)
            if (sel == e) { slot = slotA[e]; t = tA[e]; st = stA[e]; pending = pendA[e]; }
        const uint32_t user = static_cast<uint32_t>(d.first_user + slot);
        const float* om_sel = om32 + max(sel, 0) * (K2 * 64);     // [K2][64]
        // ---- one raw-log row per lane that emits an event (not for the pending phantom rows: they have their own array) ----
        const unsigned long long rowm = __ballot((run_org && st == RG_STATE_ORGANIC) || (run_ban && st == RG_STATE_BANDIT));
        const uint32_t nlive = static_cast<uint32_t>(__popcll(rowm));
        if (row_next + nlive > row_end) {
            for (uint64_t r = row_next + lane; r < row_end; r += 64)
                if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&d.counters[kCntTailRows], static_cast<unsigned long long>(chunk_rows));
            base = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base >> 32))) << 32) |
                   __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base));
            row_next = base; row_end = base + chunk_rows;
        }
        const uint64_t my_row = row_next + prefix_in_mask(rowm);
        row_next += nlive;
        const bool alive = (run_org && st == RG_STATE_ORGANIC) || (run_ban && st == RG_STATE_BANDIT);
        const rg_u32x4 w = rg_draw(d.seed, user, t, 0, RG_DRAW_EVENT);
        const bool is_org = alive && st == RG_STATE_ORGANIC;
        bool parked = false;
        // =========================== organic product draw ===========================
        const unsigned long long org_mask = __ballot(is_org);
        if (org_mask && RG_WALK_ABL(20)) {          // timing experiment: no draw at all
            if (is_org) {
                if (d.log && my_row < d.log_cap) { rg_event e; e.u = user; e.t = t; e.code = (user + t) % d.P; e.ps = __builtin_nanf(""); d.log[my_row] = e; }
                c_org += 1;
            }
        } else
        if (org_mask) {
            const bool search = is_org && !pending;
            const size_t row = search ? slot : d.n_cap;
            // ---- the user's cache row (k_draw_cached phase 1) ----
            const float4* rp = reinterpret_cast<const float4*>(d.cache_row + row * d.cache_row_f);
            float W[kMaxSC];
#pragma unroll
            for (int i = 0; i < kMaxSC / 4; ++i) {
                const float4 x = rp[i];
                W[4 * i] = x.x; W[4 * i + 1] = x.y; W[4 * i + 2] = x.z; W[4 * i + 3] = x.w;
            }
            const float4 hdr = rp[8];
            const float4 of0 = rp[9], of1 = rp[10];
            const float Q = hdr.x;
            const double delta = static_cast<double>(hdr.y);
            double S = 0.0;
#pragma unroll
            for (uint32_t sc = 0; sc < kMaxSC; ++sc) S += static_cast<double>(W[sc]);
            const double u_org = d.u_override ? d.u_override[slot] : rg_uniform(w.w[0], w.w[1]);
            const double tau = u_org * S;
            double pb = 0.0;
            uint32_t sc_star = d.n_sc - 1;
            bool found_sc = false;
            {
                double run = 0.0;
#pragma unroll
                for (uint32_t sc = 0; sc < kMaxSC; ++sc) {
                    const double Wd = static_cast<double>(W[sc]);
                    if (sc < d.n_sc && !found_sc && run + Wd > tau) { found_sc = true; sc_star = sc; pb = run; }
                    if (sc < d.n_sc && !found_sc) run += Wd;
                }
            }
            uint32_t offw;
            {
                const uint32_t q = sc_star >> 2;
                const float4 o4 = q < 4 ? of0 : of1;
                const float ow = (q & 3) == 0 ? o4.x : (q & 3) == 1 ? o4.y : (q & 3) == 2 ? o4.z : o4.w;
                offw = (__builtin_bit_cast(uint32_t, ow) >> (8 * (sc_star & 3))) & 0xFFu;
            }
            if (offw >= 127u) found_sc = false;
            const float f_star = found_sc ? __builtin_amdgcn_exp2f(-static_cast<float>(offw)) : 1.0f;
            // ---- the chunk inside that super-chunk (phase 2) ----
            uint32_t c_star = 0;
            bool found_c = false;
            {
                const uint32_t c0 = sc_star * d.sc_chunks, c1 = min(c0 + d.sc_chunks, d.n_chunks);
                const float* cp = d.cache_chunk + row * d.n_chunks;
                double run = pb;
                for (uint32_t cb = c0; cb < c1; cb += 16) {
                    float4 w4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        w4[i] = cb + 4 * i < c1 ? *reinterpret_cast<const float4*>(cp + cb + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float4 q4 = w4[i >> 2];
                        const float wv = (i & 3) == 0 ? q4.x : (i & 3) == 1 ? q4.y : (i & 3) == 2 ? q4.z : q4.w;
                        const double Wd = static_cast<double>(wv * f_star);
                        const uint32_t c = cb + i;
                        if (c < c1 && !found_c && run + Wd > tau) { found_c = true; c_star = c; pb = run; }
                        if (c < c1 && !found_c) run += Wd;
                    }
                }
            }
            found_c = found_c && found_sc;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // ---- the 32 products of the chosen chunk (phase 3): eight searching users per pass, eight lanes per
            // user, four products per lane.  A pass is ONE latency chain (user parameters -> 21 coalesced
            // 16-byte loads -> 80 fma -> 4 exp -> 3-step prefix across the user's lanes -> compare); two users
            // per pass of 32-lane prefixes cost a chain per pair and made the walk 4x slower ----
            const int grp = lane >> 3, gl = lane & 7;
            unsigned long long todo = __ballot(search);
            if RG_WALK_ABL(16) { todo = 0; if (search) { mbox[lane * 3] = 0.0; mbox[lane * 3 + 1] = 0.0; mbox[lane * 3 + 2] = 1e300; } }
            while (todo) {
                int src = -1;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int bit = todo ? __builtin_ctzll(todo) : -1;
                    if (g == grp) src = bit;
                    if (todo) todo &= todo - 1;
                }
                const bool has = src >= 0;
                const int s2 = has ? src : 0;
                const uint32_t cs = static_cast<uint32_t>(__shfl(static_cast<int>(c_star), s2));
                const float Qs = __shfl(Q, s2);
                const double pbs = __shfl(pb, s2), taus = __shfl(tau, s2);
                const float4* gp = reinterpret_cast<const float4*>(d.gamma32t + (static_cast<size_t>(cs) * K2) * 32) + gl;
                float4 l = *(reinterpret_cast<const float4*>(d.mu32 + cs * 32) + gl);
                const float* o = om32 + __shfl(max(sel, 0), s2) * (K2 * 64) + s2;
#pragma unroll
                for (int kh = 0; kh < K2; kh += KH) {          // two halves: KH 16-byte loads in flight, then their fmas
                    float4 gk[KH];
#pragma unroll
                    for (int k = 0; k < KH; ++k) gk[k] = gp[(kh + k) * 8];
#pragma unroll
                    for (int k = 0; k < KH; ++k) {
                        const float wk = o[(kh + k) * 64];
                        l.x = fmaf(gk[k].x, wk, l.x); l.y = fmaf(gk[k].y, wk, l.y);
                        l.z = fmaf(gk[k].z, wk, l.z); l.w = fmaf(gk[k].w, wk, l.w);
                    }
                    asm volatile("" : "+v"(l.x), "+v"(l.y), "+v"(l.z), "+v"(l.w));   // keeps the second half's loads behind these
                }
                const float e0 = __builtin_amdgcn_exp2f(fmaf(l.x, kLog2e, -Qs)), e1 = __builtin_amdgcn_exp2f(fmaf(l.y, kLog2e, -Qs));
                const float e2 = __builtin_amdgcn_exp2f(fmaf(l.z, kLog2e, -Qs)), e3 = __builtin_amdgcn_exp2f(fmaf(l.w, kLog2e, -Qs));
                const float q0 = e0, q1 = q0 + e1, q2 = q1 + e2, q3 = q2 + e3;      // prefix inside the lane
                float inc = q3;                                                     // ... and across the user's 8 lanes
#pragma unroll
                for (int o2 = 1; o2 < 8; o2 <<= 1) {
                    const float y = __shfl_up(inc, o2, 8);
                    if (gl >= o2) inc += y;
                }
                float ex = __shfl_up(inc, 1, 8);                                    // prefix before this lane's products
                if (gl == 0) ex = 0.0f;
                const double pxb = pbs + static_cast<double>(ex);
                const double px0 = pbs + static_cast<double>(ex + q0), px1 = pbs + static_cast<double>(ex + q1);
                const double px2 = pbs + static_cast<double>(ex + q2), px3 = pbs + static_cast<double>(ex + q3);
                const int j0 = px0 > taus ? 0 : px1 > taus ? 1 : px2 > taus ? 2 : px3 > taus ? 3 : -1;
                const unsigned long long hits = __ballot(has && j0 >= 0);
                const uint32_t gmask = static_cast<uint32_t>(hits >> (8 * grp)) & 0xFFu;
                if (has) {
                    if (gmask) {
                        if (gl == __builtin_ctz(gmask)) {
                            mbox[src * 3] = static_cast<double>(4 * gl + j0);
                            mbox[src * 3 + 1] = j0 == 0 ? pxb : j0 == 1 ? px0 : j0 == 2 ? px1 : px2;
                            mbox[src * 3 + 2] = j0 == 0 ? px0 : j0 == 1 ? px1 : j0 == 2 ? px2 : px3;
                        }
                    } else if (gl == 0) { mbox[src * 3] = -1.0; mbox[src * 3 + 1] = pbs; mbox[src * 3 + 2] = pbs; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            uint32_t v = 0;
            bool ok = false;
            if (search) {
                const int idx = static_cast<int>(mbox[lane * 3]);
                const double Av = mbox[lane * 3 + 1], Bv = mbox[lane * 3 + 2];
                v = c_star * 32 + static_cast<uint32_t>(max(idx, 0));
                const CertLin ct = cert_correlated(S, pb, Av - pb, Bv - pb, delta);
                ok = found_c && idx >= 0 && v < d.P && ct.valid &&
                     (v == 0 || u_org * ct.den_lo > ct.num_lo) &&
                     (v == d.P - 1 || u_org * ct.den_hi < ct.num_hi);
            }
            // ---- uncertified: float64 pick from the user's stored sums, or park the user until they exist ----
            const bool need64 = is_org && !ok;
            const bool have64 = need64 && d.f64_valid[slot] != 0;
            parked = need64 && !have64;
            unsigned long long picks = __ballot(have64);
            while (picks) {
                const int L = __builtin_ctzll(picks);
                picks &= picks - 1;
                const uint32_t s_slot = static_cast<uint32_t>(__shfl(static_cast<int>(slot), L));
                const double s_u = __shfl(u_org, L);
                const double M = static_cast<double>(d.exact_ref[s_slot]) * 0.69314718055994530942;
                const uint32_t pv = exact_pick_wave(d, d.exact_sums + static_cast<size_t>(s_slot) * n_cc,
                                                    d.omega + static_cast<size_t>(s_slot) * d.OMS, M, s_u, 1u, lane);
                if (lane == L) { v = pv; c_pick += 1; }
                __builtin_amdgcn_wave_barrier();
            }
            if (parked) {
                d.park_t[slot] = t | (static_cast<uint32_t>(RG_STATE_ORGANIC) << 24) | (1u << 27);
                d.exact_ref[slot] = Q;
            }
            if (is_org && !parked) {
                if (d.log && my_row < d.log_cap) {
                    rg_event e;
                    e.u = user; e.t = t; e.code = v; e.ps = __builtin_nanf("");
                    d.log[my_row] = e;
                }
                if (d.lpv) d.lpv[slot] = v;
                if (d.hist_cap && !RG_WALK_ABL(17)) history_add(d, slot, v);
                c_org += 1;
                pending = false;
            }
        }
        // ---- park list entries for the users parked in this step ----
        const unsigned long long pmask = __ballot(parked);
        if (pmask) {
            const uint32_t np = static_cast<uint32_t>(__popcll(pmask));
            if (park_next + np > park_end) {
                for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
                uint32_t base = 0;
                if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_park, 64ull));
                base = __builtin_amdgcn_readfirstlane(base);
                park_next = base; park_end = base + 64;
            }
            if (parked) {
                d.park_list[out_base + park_next + prefix_in_mask(pmask)] = slot;
                if (d.log && my_row < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[my_row] = e; }
                st = kEmpty;
            }
            park_next += np;
        }
        // =========================== bandit event + transition (k_advance's arithmetic) ===========================
        const bool is_ban = run_ban && st == RG_STATE_BANDIT, is_ph = run_ban && st == kPhantom;
        double ps = 1.0;
        uint32_t a = 0;
        if (is_ban || is_ph) a = RG_WALK_ABL(18) ? (user + t) % d.P : policy_act<DENSE>(d, slot, user, t, &ps);
        if (is_ph) {       // final step_offline(done = True): the act above, reward 0 (abstract.py:223-233,311-316); t is already the row's time
            rg_event e;
            e.u = user; e.t = t; e.code = RG_EV_BANDIT | RG_EV_PHANTOM | a;
            e.ps = static_cast<float>(ps);
            d.phantom[slot] = e;
            d.phantom_ps[slot] = ps;
            d.has_phantom[slot] = 1;
            c_ph += 1;
            st = kEmpty;
        }
        if (alive && !parked) {
            const double u_trans = rg_uniform(w.w[2], w.w[3]);
            bool click = false;
            // The click is a Bernoulli draw against ff(beta[a].omega + mu_b[a]) (three nested sigmoids: three float64
            // exps and four divisions).  Its outcome is decided in fp32 wherever the fp32 value of 1 - ff is further from
            // the uniform than the fp32 error bound (fp32 dot: (K + 2) 2^-24 sum|beta_k omega_k|, damped by the chain's
            // slope <= 0.05; three v_exp / v_rcp at ~1e-6); the float64 evaluation below is for the lanes inside that
            // band (~4e-5 of the acts) and for runs that export the click probability.
            bool click_known = false;
            if (is_ban && !d.aux_pclick && rg_uniform(w.w[0], w.w[1]) < kNoClickBelow) click_known = true;     // (click = false)
            else
            if (is_ban && !d.aux_pclick && !RG_WALK_ABL(21)) {
                const float* om_l = om_sel + lane;
                const int dec = click_decide32<((K2 + 3) / 4) * 4>(d.beta32 + static_cast<size_t>(a) * d.KB4, [&](int k) { return om_l[k * 64]; },
                                                                   d.K, d.KB4, static_cast<float>(d.mu_b[a]), rg_uniform(w.w[0], w.w[1]));
                if (dec >= 0) { click = dec != 0; click_known = true; }
            }
            if (is_ban && click_known) {
                c_clicks += click;
                c_ban += 1;
                if (d.log && my_row < d.log_cap) {
                    rg_event e;
                    e.u = user; e.t = t;
                    e.code = RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a;
                    e.ps = static_cast<float>(ps);
                    d.log[my_row] = e;
                    if (d.aux_ps) d.aux_ps[my_row] = ps;
                }
            }
            if (is_ban && !click_known) {
                const double* b = d.beta + static_cast<size_t>(a) * d.K;
                const double* om = d.omega + static_cast<size_t>(slot) * d.OMS;
                double x = 0.0;
                if RG_WALK_ABL(19) {}
                else if (!(d.K & 1)) {
                    // rows of K even are 16-byte aligned: half as many (scattered) load requests as 8-byte loads
                    for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {
                        double2 wv[4], bv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t k = min(k0 + 2 * i, d.K - 2);
                            wv[i] = *reinterpret_cast<const double2*>(om + k);
                            bv[i] = *reinterpret_cast<const double2*>(b + k);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (k0 + 2 * i < d.K) { x += bv[i].x * wv[i].x; x += bv[i].y * wv[i].y; }
                    }
                } else
                for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {
                    double wv[8], bv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t k = min(k0 + i, d.K - 1);
                        wv[i] = om[k];
                        bv[i] = b[k];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (k0 + i < d.K) x += bv[i] * wv[i];
                }
                const double ctr = RG_WALK_ABL(19) ? 0.01 : ff64(x + d.mu_b[a]);
                const double p0 = 1.0 - ctr;
                click = (p0 / (p0 + ctr)) <= rg_uniform(w.w[0], w.w[1]);
                c_clicks += click;
                c_ban += 1;
                if (d.log && my_row < d.log_cap) {
                    rg_event e;
                    e.u = user; e.t = t;
                    e.code = RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a;
                    e.ps = static_cast<float>(ps);
                    d.log[my_row] = e;
                    if (d.aux_ps) d.aux_ps[my_row] = ps;
                    if (d.aux_pclick) d.aux_pclick[my_row] = ctr;
                }
            }
            const double c0 = is_org ? d.cdf_o0 : d.cdf_b0, c1 = is_org ? d.cdf_o1 : d.cdf_b1;
            int ns = (c0 <= u_trans) + (c1 <= u_trans);
            if (click) ns = RG_STATE_ORGANIC;                  // abstract.py:180-181 (sigma_omega == 0: no drift to apply)
            const bool organic_only = (d.first_user + slot) < d.organic_only_below;
            if (organic_only && ns != RG_STATE_ORGANIC) {
                ns = RG_STATE_STOP;
                d.n_events[slot] = t + 1;
            } else if (ns == RG_STATE_STOP) {
                d.n_events[slot] = t + 1;
                ns = kPhantom;                                 // the phantom row's act: this lane's next step
            } else if (t + 2 >= kMaxSteps) {
                ns = RG_STATE_STOP;
                d.n_events[slot] = t + 1;
                c_limit += 1;
            }
            if (ns == RG_STATE_STOP || ns == kPhantom) c_maxt = max(c_maxt, t + 1);
            if (ns == RG_STATE_STOP) st = kEmpty;
            else { st = ns; t += 1; }
        }
#pragma unroll
        for (int e = 0; e < kWalkUsers; ++e)
            if (sel == e) { stA[e] = st; tA[e] = t; pendA[e] = pending; }
        }   // if (live)
    }
    // ---- leftovers of the reserved chunks, counters ----
    for (uint64_t r = row_next + lane; r < row_end; r += 64)
        if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
    for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
    for (int o = 32; o > 0; o >>= 1) {
        c_org += __shfl_xor(c_org, o); c_ban += __shfl_xor(c_ban, o); c_clicks += __shfl_xor(c_clicks, o);
        c_ph += __shfl_xor(c_ph, o); c_pick += __shfl_xor(c_pick, o); c_sweeps += __shfl_xor(c_sweeps, o);
        c_maxt = max(c_maxt, static_cast<uint32_t>(__shfl_xor(static_cast<int>(c_maxt), o)));
        c_limit += static_cast<uint32_t>(__shfl_xor(static_cast<int>(c_limit), o));
    }
    if (lane == 0) {
        if (c_org) atomicAdd(&d.counters[kCntTailOrganic], c_org);
        if (c_ban) atomicAdd(&d.counters[kCntTailBandit], c_ban);
        if (c_clicks) atomicAdd(&d.counters[RG_CNT_CLICKS], c_clicks);
        if (c_ph) atomicAdd(&d.counters[RG_CNT_PHANTOM], c_ph);
        if (c_pick) atomicAdd(&d.counters[RG_CNT_EXACT_DRAWS], c_pick);
        if (c_sweeps) atomicAdd(&d.counters[RG_CNT_EXACT_SWEEPS], c_sweeps);
        if (c_maxt) atomicMax(&d.counters[kCntTailMaxT], static_cast<unsigned long long>(c_maxt));
        if (c_limit) atomicAdd(&d.counters[kCntTailLimit], static_cast<unsigned long long>(c_limit));
    }
}
#endif

// ------------------------------------------------------------------------------------------
// k_walk2 — the user-major walk, second form (the default where it applies; k_walk above remains for the other
// configurations and as RECOGYM_WALK=1).  Same contract, lists, rounds, parking and hand-over as k_walk; what changed is
// what an event costs in DEPENDENT memory round trips, the thing that bound k_walk (61 % of its wave cycles in
// s_waitcnt at three waves per SIMD):
//   * prefix form of the per-user sums (k_cache_prefix, once per run): the 32 super-chunk sums and the chunk sums of a user
//     become fp32 prefix sums on the user's common reference, so the two search levels are "count the prefixes <= u S"
//     (one compare per element, no float64 running sum, no per-super-chunk scale);
//   * a per-user MEMO of certified draws: the first time the search certifies product v for a user, the u-interval that
//     is certified for v — [C~[v-1](1+d)/(S~(1-d)), C~[v](1-d)/(S~(1+d))] rounded inwards — joins the user's hot row
//     (9 entries in one 128-byte line).  A user's softmax never changes (sigma_omega = 0) and is peaked (its top product
//     holds 46 % of the mass on C3, the top 8 hold 82 %), so most later draws of the user land in a memoised interval:
//     one load, no search.  A memo hit IS a certificate (the same inequality), so the logged index is float64's either way;
//   * three event kinds per wave iteration instead of two: organic draws answered by the memo, organic draws that need the
//     search (they wait until >= 16 lanes of the wave do: the search's passes then run full), bandit events;
//   * the user's view history (header + 15 products: most users' whole history) lives in LDS for the user's stay on the
//     lane (write-through to its row in HBM): the OrganicUserEventCounter act and the view insertion touch no memory;
//   * omega32 of the lane's user in 2 KH registers (the chunk recompute fetches the searching users' by ds_bpermute),
//     counters in scalar registers: <= 128 VGPRs, four waves per SIMD.
// ------------------------------------------------------------------------------------------
constexpr int kHotEntries = 9;          // memo entries of a user: floats [4 + 3 j, 7 + 3 j) of its hot row = {product, u_lo, u_hi}
constexpr int kWSlow = 6;               // lane state: organic draw that missed the memo (RG_STATE_* = 0..2, empty 3, phantom 4)
constexpr int kWClick = 7;              // lane state: bandit event whose click needs ctr (uniform >= kNoClickBelow): taken in batches
__host__ __device__ inline size_t walk2_wave_lds(int hist) { return (hist ? 16 * 64 * 8 : 0) + 64 * 12; }
#if RG_HAS(7)

// The prefix form.  Eight lanes per user, 32 chunks per pass (one 128-byte line of the user's chunk sums): scaled to the
// user's common reference Q (exact powers of two), summed in float64 in chunk order, stored back in place as fp32
// prefixes; the prefix at the end of every super-chunk also goes to the user's scp row, the total into its hot row.
// fused = 1: k_draw_bf16p already stored prefixes (sweep_only = 2), each on the reference of its super-chunk: what is left is
// the hot row's header and, for the users whose reference moved during the sweep (cache_resc != 0: rare), the exact rescaling
// (powers of two) of their entries to the common reference.
__global__ void __launch_bounds__(kBlock) k_cache_prefix(DevSim d, int fused) {
    const int lane = lane_id(), grp = lane >> 3, gl = lane & 7;
    const uint32_t n_groups = (d.grp_n + 7) / 8;
    const uint32_t waves = gridDim.x * (kBlock / 64);
    for (uint32_t ug = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); ug < n_groups; ug += waves) {
        const uint32_t i = d.grp_lo + ug * 8 + grp;
        // (fin_in_sweep: only the users whose reference moved during the sweep — the others' prefixes need no rescaling and the
        // sweep left their hot rows)
        const bool act = i < d.grp_lo + d.grp_n && !(d.fin_in_sweep && d.cache_resc[i < d.grp_lo + d.grp_n ? i : d.n_cap] == 0);
        const size_t row = act ? i : d.n_cap;
        const float4* r4 = reinterpret_cast<const float4*>(d.cache_row + row * d.cache_row_f);
        const float4 hdr = r4[8], of0 = r4[9], of1 = r4[10];
        float* cp = d.cache_chunk + row * d.n_chunks;
        float* scp = d.walk_scp + row * kMaxSC;
        double run = 0.0;
        const bool moved = fused && act && d.cache_resc[row] != 0;
        if (fused) {
            if (moved) {
                for (uint32_t c0 = 0; c0 < d.n_chunks; c0 += 32) {
                    const uint32_t c = c0 + 4 * gl;
                    if (c >= d.n_chunks) continue;
                    const uint32_t sc = min(c / d.sc_chunks, kMaxSC - 1u);
                    const uint32_t q = sc >> 2;
                    const float4 o4 = q < 4 ? of0 : of1;
                    const float ow = (q & 3) == 0 ? o4.x : (q & 3) == 1 ? o4.y : (q & 3) == 2 ? o4.z : o4.w;
                    const uint32_t off = (__builtin_bit_cast(uint32_t, ow) >> (8 * (sc & 3))) & 0xFFu;
                    const float f = off >= 127u ? 0.0f : __builtin_amdgcn_exp2f(-static_cast<float>(off));
                    float4 w = *reinterpret_cast<const float4*>(cp + c);
                    w.x *= f; w.y *= f; w.z *= f; w.w *= f;
                    *reinterpret_cast<float4*>(cp + c) = w;
                    if ((c + 4) % d.sc_chunks == 0 || c + 4 == d.n_chunks) scp[sc] = w.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            run = act ? static_cast<double>(scp[d.n_sc - 1]) : 0.0;
        } else
        for (uint32_t c0 = 0; c0 < d.n_chunks; c0 += 32) {
            const uint32_t c = c0 + 4 * gl;
            const bool in = c < d.n_chunks;
            const float4 w = in ? *reinterpret_cast<const float4*>(cp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const uint32_t sc = min(c / d.sc_chunks, kMaxSC - 1u);       // (sc_chunks % 4 == 0: one super-chunk per float4)
            const uint32_t q = sc >> 2;
            const float4 o4 = q < 4 ? of0 : of1;
            const float ow = (q & 3) == 0 ? o4.x : (q & 3) == 1 ? o4.y : (q & 3) == 2 ? o4.z : o4.w;
            const uint32_t off = (__builtin_bit_cast(uint32_t, ow) >> (8 * (sc & 3))) & 0xFFu;
            const float f = off >= 127u ? 0.0f : __builtin_amdgcn_exp2f(-static_cast<float>(off));
            const double p0 = static_cast<double>(w.x * f), p1 = p0 + static_cast<double>(w.y * f);
            const double p2 = p1 + static_cast<double>(w.z * f), p3 = p2 + static_cast<double>(w.w * f);
            double inc = p3;
#pragma unroll
            for (int o2 = 1; o2 < 8; o2 <<= 1) {
                const double y = __shfl_up(inc, o2, 8);
                if (gl >= o2) inc += y;
            }
            const double base = run + (inc - p3);
            const float4 out = make_float4(static_cast<float>(base + p0), static_cast<float>(base + p1),
                                           static_cast<float>(base + p2), static_cast<float>(base + p3));
            if (in && act) {
                *reinterpret_cast<float4*>(cp + c) = out;
                if ((c + 4) % d.sc_chunks == 0 || c + 4 == d.n_chunks) scp[sc] = out.w;
            }
            run += __shfl(inc, (grp << 3) | 7);
        }
        if (act) {
            for (uint32_t sc = d.n_sc + gl; sc < kMaxSC; sc += 8) scp[sc] = INFINITY;     // never counted
            if (gl == 0) {
                // S~ as the search sees it (the last prefix), the certificate's delta + 2^-21 for the roundings of the stored
                // prefixes (<= 5 of 2^-24 each, relative to the prefix: the same kind of error the budget is made of), Q, an empty memo
                float4* hot = reinterpret_cast<float4*>(d.walk_hot + row * 32);
                hot[0] = make_float4(static_cast<float>(run), hdr.y * 1.000001f + 4.8e-7f, hdr.x, __builtin_bit_cast(float, 0u));
            }
        }
    }
}

// next float above / below (finite x; the roundings of the memo's interval bounds and of the uniform go INWARDS)
__device__ __forceinline__ float f32_up(float x) {
    const uint32_t b = __builtin_bit_cast(uint32_t, x);
    return x == 0.0f ? __builtin_bit_cast(float, 1u) : __builtin_bit_cast(float, x > 0.0f ? b + 1u : b - 1u);
}
__device__ __forceinline__ float f32_down(float x) {
    const uint32_t b = __builtin_bit_cast(uint32_t, x);
    return x == 0.0f ? __builtin_bit_cast(float, 0x80000001u) : __builtin_bit_cast(float, x > 0.0f ? b - 1u : b + 1u);
}

// The float64 pick on sums stored as PREFIXES (k_exact_prefix): the 64-product chunk by counting the prefixes <= u total
// (three ballots instead of three wave scans), then its products walked in product order as exact_pick_wave does.
__device__ __forceinline__ uint32_t exact_pick_pfx(const DevSim& d, const double* pfx, const double* om, double M, double u, int lane) {
    const uint32_t n_cc = d.PT / 64;
    const double total = pfx[n_cc - 1];
    const double target = u * total;
    uint32_t cnt = 0;
    for (uint32_t c0 = 0; c0 < n_cc; c0 += 64) {
        const uint32_t c = c0 + lane;
        cnt += static_cast<uint32_t>(__popcll(__ballot(c < n_cc && pfx[c] <= target)));
    }
    const uint32_t ccstar = min(cnt, n_cc - 1u);            // (u * total rounded up to total: the last chunk)
    double acc = ccstar ? pfx[ccstar - 1] : 0.0;
    uint32_t v = min(ccstar * 64 + 63, d.P - 1);           // if rounding leaves no hit: the chunk's last product
    const uint32_t p = ccstar * 64 + lane;
    const double* g = d.gammaT + p;                        // PT columns: always in range
    double lg = 0.0;
    for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {             // same association as the oracle (k ascending)
        double gv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = g[static_cast<size_t>(min(k0 + j, d.K - 1)) * d.PT];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k0 + j < d.K) lg += gv[j] * om[k0 + j];
    }
    lg = p < d.P ? lg + d.mu_o[p] : -INFINITY;
    const double inc = wave_scan(exp64(lg - M), lane);
    const unsigned long long hit = __ballot(p < d.P && acc + inc > target);
    if (hit) v = ccstar * 64 + static_cast<uint32_t>(__builtin_ctzll(hit));
    return v;
}

// exact_sums rows of the listed users (the float64 batch between rounds 1 and 2 just took them) -> inclusive prefixes, in
// place, in exact_pick_wave's association (a wave scan per block of 64 sums, the blocks in order).  A wave per user.
__global__ void __launch_bounds__(kBlock) k_exact_prefix(DevSim d, uint32_t n_list) {
    const int lane = lane_id();
    const uint32_t n_cc = d.PT / 64;
    const uint32_t waves = gridDim.x * (kBlock / 64);
    if (d.q_count) n_list = static_cast<uint32_t>(*d.q_count);
    for (uint32_t w = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w < n_list; w += waves) {
        const uint32_t slot = d.park_list[d.list_in + w];
        if (slot == 0xFFFFFFFFu) continue;
        double* row = d.exact_sums + static_cast<size_t>(slot) * n_cc;
        double run = 0.0;
        for (uint32_t c0 = 0; c0 < n_cc; c0 += 64) {
            const uint32_t c = c0 + lane;
            const double incl = wave_scan(c < n_cc ? row[c] : 0.0, lane);
            if (c < n_cc) row[c] = run + incl;
            run += __shfl(incl, 63);
        }
    }
}

// (the float64 pick as a noinline CALL freed ~40 registers of the walk's loop, but at four blocks per CU — 128 VGPRs, the loop
// spilling around the call — the kernel no longer reproduced the oracle: measured, dropped; inlined at three blocks per CU
// the loop holds everything in 168 registers)
__device__ __forceinline__ uint32_t exact_pick_call(const DevSim& d, const double* sums, const double* om, double M,
                                                   double u, int lane) {
    return exact_pick_wave(d, sums, om, M, u, 1u, lane);
}

// k_walk2's COMPACT history line (HIST == 2: products < 65 535, hist_cap <= 32 768): the same 128 bytes of LDS per lane hold
// 32 words instead of 16 64-bit entries — word 0 the header (views << 15 | distinct), words 1 .. 31 the 31 smallest products as
// (PREFIX << 16 | product): the running view count up to and including the product in the high half (a user has < 65 536
// events), so the words ascend with the index, an unused word is 0xFFFFFFFF, and the policy's act — first product whose
// cumulative count exceeds u x views — is a COUNT of words below a key: two LDS round trips (the last word of every 8-word
// segment, then the segment) and ~40 vector instructions instead of a 10-instruction step per entry, and all but ~1 % of C3's
// events find their whole history in the line (15 products in 64-bit entries: 11.6 % beyond).  Word w of the line is half
// (w & 1) of the 64-bit LDS entry hl[(w >> 1) * 64]; the user's ROW keeps the (product, count) form every other kernel reads.
constexpr uint32_t kHcLine = 32;     // words of the compact line (header + 31 products)

// Three blocks per CU (168 VGPRs, no spills).  Four (128 VGPRs) were measured in two forms — omega32 re-read from the cache
// row instead of held in registers, and the Gamma rows of the chunk pass in two batches — and did not pay: the extra loads and
// spills cost what the fourth wave brought (C3 walk 130.9 vs 133.8 ms, C2 13.2 vs 12.1 ms: profiles/r3/ab_walk_call3.jsonl).
// The DevSim fields are read from the kernel-argument segment where they are used (as in k_walk); pinning the 25 or 38 of
// the main path in registers instead (104 -> 70 / 60 scalar loads in the code, 200 / 259 scalar registers in VGPR lanes)
// measured the same to 0.3 % (profiles/r3/ab_call11_pinned_fields_shard_sizes.jsonl): the waits are not the scalar loads'.
template <int KH, int HIST>
__global__ void __launch_bounds__(kBlock, (KH <= 10 ? 3 : 2)) k_walk2(DevSim d_arg, uint32_t n_work, int round, uint32_t chunk_rows,
                                                        uint32_t in_base, uint32_t out_base) {
    (void)d_arg;       // read from the kernel-argument segment at the point of use (see k_walk)
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int K2 = 2 * KH;
    constexpr int KC = ((K2 + 3) / 4) * 4;
    constexpr int kEmpty = 3, kPhantom = 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    char* wbase = smem_raw + static_cast<size_t>(wave) * walk2_wave_lds(HIST);
    hent_t* hl = reinterpret_cast<hent_t*>(wbase) + lane;               // HIST: [16][64] entry-major: hl[i * 64]
    float* mboxf = reinterpret_cast<float*>(wbase + (HIST ? 16 * 64 * 8 : 0));   // [64][3]: the search's result per lane
    uint32_t slot = 0, t = 0;
    int st = kEmpty;
    bool hdirty = false;                                               // the history line in LDS is newer than the user's row
    bool pend = false;                                                 // rounds >= 2: the parked draw, to be picked in float64
    float om[KC];                                                      // omega32 of the lane's user
#pragma unroll
    for (int k = 0; k < KC; ++k) om[k] = 0.0f;
    uint32_t res_next = 0, res_end = 0;
    uint64_t row_next = 0, row_end = 0;
    uint32_t park_next = 0, park_end = 0;
    bool exhausted = false;
    // wave-uniform tallies (scalar registers): events are counted by ballots
    uint32_t c_org = 0, c_ban = 0, c_clicks = 0, c_ph = 0, c_pick = 0, c_sweeps = 0, c_maxt = 0, c_limit = 0, c_hit = 0, c_anch = 0;
    if (const unsigned long long* qc = ((const DevSim*)kargs)->q_count) n_work = static_cast<uint32_t>(*qc);   // (pipeline: the list's length is on the device)

    for (;;) {
        asm volatile("" : "+s"(kargs));
        const DevSim& d = *(const DevSim*)kargs;
        const uint32_t n_cc = d.PT / 64;
        // the view history is written back when the lane lets go of the user (stop, park, hand-over) or needs the row
        auto flush_hist = [&](bool c) {
            if (HIST == 2 && c) {
                // (product, count) entries from the prefixes; the pairs that hold entries <= nd (what lies behind them in the row
                // is don't-care)
                ulonglong2* hw = reinterpret_cast<ulonglong2*>(hist_row(d, slot));
                const uint32_t h0 = static_cast<uint32_t>(hl[0]);
                const uint32_t nd = h0 & 0x7FFFu;
                uint32_t prev = 0u;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const hent_t x = hl[i * 64];
                    const uint32_t w0 = static_cast<uint32_t>(x), w1 = static_cast<uint32_t>(x >> 32);
                    hent_t e0, e1;
                    if (i == 0) e0 = (static_cast<hent_t>(h0 >> 15) << 32) | nd;
                    else { e0 = (static_cast<hent_t>(w0 & 0xFFFFu) << 32) | ((w0 >> 16) - prev); prev = w0 >> 16; }
                    e1 = (static_cast<hent_t>(w1 & 0xFFFFu) << 32) | ((w1 >> 16) - prev); prev = w1 >> 16;
                    if (static_cast<uint32_t>(2 * i) <= nd) hw[i] = make_ulonglong2(e0, e1);
                }
            } else
            if (HIST && c) {
                ulonglong2* hw = reinterpret_cast<ulonglong2*>(hist_row(d, slot));
#pragma unroll
                for (int i = 0; i < 8; ++i) hw[i] = make_ulonglong2(hl[(2 * i) * 64], hl[(2 * i + 1) * 64]);
            }
        };
        // the compact line from the user's row (its first 32 entries): running prefixes of the counts, unused words all ones
        auto load_compact = [&](uint32_t s_row) {
            const ulonglong2* hr2 = reinterpret_cast<const ulonglong2*>(hist_row(d, s_row));
            uint32_t run = 0u, nd = 0u;
#pragma unroll
            for (int b = 0; b < 16; b += 8) {           // (two batches of eight 16-byte loads: 32 registers in flight, not 64)
                ulonglong2 x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = hr2[b + i];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uint32_t w0, w1;
                    const bool in0 = static_cast<uint32_t>(2 * (b + i)) <= nd || b + i == 0;
                    if (b + i == 0) { nd = h_cnt(x[0].x); w0 = (h_prod(x[0].x) << 15) | nd; }
                    else { run += in0 ? h_cnt(x[i].x) : 0u; w0 = in0 ? ((run << 16) | h_prod(x[i].x)) : 0xFFFFFFFFu; }
                    const bool in1 = static_cast<uint32_t>(2 * (b + i) + 1) <= nd;
                    run += in1 ? h_cnt(x[i].y) : 0u;
                    w1 = in1 ? ((run << 16) | h_prod(x[i].y)) : 0xFFFFFFFFu;
                    hl[(b + i) * 64] = static_cast<hent_t>(w0) | (static_cast<hent_t>(w1) << 32);
                }
                asm volatile("" ::: "memory");
            }
        };
        // ---- refill the lanes whose user has stopped (or was parked) ----
        {
            unsigned long long dead = __ballot(st == kEmpty);
            if (dead && !exhausted && (static_cast<uint32_t>(__popcll(dead)) >= d.walk_refill || dead == ~0ull)) {
                for (int pass = 0; pass < 2 && dead; ++pass) {
                    if (res_next == res_end) {
                        if (exhausted) break;
                        uint32_t base = 0;
                        if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_ticket, 64ull));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base >= n_work) { exhausted = true; break; }
                        res_next = base; res_end = min(base + 64u, n_work);
                    }
                    const uint32_t take = min(static_cast<uint32_t>(__popcll(dead)), res_end - res_next);
                    const uint32_t r = prefix_in_mask(dead);
                    const bool mine = ((dead >> lane) & 1ull) != 0 && r < take;
                    if (mine) {
                        const uint32_t idx = res_next + r;
                        uint32_t s2 = d.grp_lo + idx;
                        if (round >= 2) s2 = d.park_list[in_base + idx];
                        if (s2 != 0xFFFFFFFFu) {
                            slot = s2; st = RG_STATE_ORGANIC; t = 0u; pend = false; hdirty = false;
                            if (round >= 2) {
                                const uint32_t pt = d.park_t[s2];
                                t = pt & 0xFFFFFFu; st = static_cast<int>((pt >> 24) & 7u); pend = (pt >> 27) & 1u;
                                if (round == 2) d.f64_valid[s2] = 1;         // the batch between the rounds took its sums
                                if (pend) st = kWSlow;                       // its draw goes straight to the float64 pick
                            }
                            {   // omega32 = float(omega), as k_cache_finalize left it in the user's cache row (floats 44 ..)
                                const float4* rp = reinterpret_cast<const float4*>(d.cache_row + static_cast<size_t>(s2) * d.cache_row_f);
#pragma unroll
                                for (int k4 = 0; k4 < K2 / 4; ++k4) {
                                    const float4 x = rp[11 + k4];
                                    om[4 * k4] = x.x; om[4 * k4 + 1] = x.y; om[4 * k4 + 2] = x.z; om[4 * k4 + 3] = x.w;
                                }
#pragma unroll
                                for (int k = (K2 / 4) * 4; k < K2; ++k) om[k] = reinterpret_cast<const float*>(rp)[44 + k];
                            }
                            if (HIST == 2) load_compact(s2);
                            else
                            if (HIST) {
                                const ulonglong2* hr2 = reinterpret_cast<const ulonglong2*>(hist_row(d, s2));
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    const ulonglong2 x = hr2[i];
                                    hl[(2 * i) * 64] = x.x; hl[(2 * i + 1) * 64] = x.y;
                                }
                            }
                        }
                    }
                    // (the float64 batch between the rounds swept for every parked user: counted where round 2 takes them)
                    if (round == 2) c_sweeps += static_cast<uint32_t>(__popcll(__ballot(mine && st != kEmpty && pend)));
                    res_next += take;
                    dead = __ballot(st == kEmpty && !mine);
                }
            }
        }
        const unsigned long long live = __ballot(st != kEmpty);
        if (!live) { if (exhausted) break; else continue; }
        // ---- hand-over (see k_walk) ----
        if (exhausted && round < 3 && d.walk_handover && static_cast<uint32_t>(__popcll(live)) <= d.walk_handover) {
            const bool give = st != kEmpty;
            const unsigned long long gm = __ballot(give);
            const uint32_t np = static_cast<uint32_t>(__popcll(gm));
            if (park_next + np > park_end) {
                for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
                uint32_t base = 0;
                if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_park, 64ull));
                base = __builtin_amdgcn_readfirstlane(base);
                park_next = base; park_end = base + 64;
            }
            flush_hist(give && hdirty);
            if (give) {
                d.park_list[out_base + park_next + prefix_in_mask(gm)] = slot;
                const int st_out = st == kWSlow ? RG_STATE_ORGANIC : st == kWClick ? RG_STATE_BANDIT : st;     // (they restart at the memo check / the act)
                d.park_t[slot] = t | (static_cast<uint32_t>(st_out) << 24) | (pend ? 1u << 27 : 0u);
                if (round == 1) d.exact_ref[slot] = d.cache_row[static_cast<size_t>(slot) * d.cache_row_f + 32];
                st = kEmpty;
            }
            park_next += np;
            break;
        }
        // ---- ONE kind of event per iteration: memo-answered organic draws, searching organic draws, bandit events ----
        const uint32_t n_o = static_cast<uint32_t>(__popcll(__ballot(st == RG_STATE_ORGANIC)));
        const uint32_t n_s = static_cast<uint32_t>(__popcll(__ballot(st == kWSlow)));
        const uint32_t n_b = static_cast<uint32_t>(__popcll(__ballot(st == RG_STATE_BANDIT || st == kPhantom)));
        const uint32_t n_c = static_cast<uint32_t>(__popcll(__ballot(st == kWClick)));
        // (walk_bias == 0: the memo-answered draws AND the bandit events of the wave in the same iteration)
        bool do_org = false, do_srch = false, do_ban = false, do_clk = false;
        if (n_s >= d.walk_search_batch || (n_s && !n_o && !n_b)) do_srch = true;
        else if (n_c && (n_c >= d.walk_click_batch || (!n_o && !n_b))) do_ban = do_clk = true;   // the bandit events that need ctr
        else if (d.walk_bias == 0u) { do_org = n_o != 0u; do_ban = n_b != 0u; }
        else if (n_o && (n_o * d.walk_bias >= n_b * 4u)) do_org = true;
        else if (n_b) do_ban = true;
        else if (n_o) do_org = true;
        else do_srch = true;
        const uint32_t user = static_cast<uint32_t>(d.first_user + slot);
        const rg_u32x4 w = rg_draw(d.seed, user, t, 0, RG_DRAW_EVENT);
        bool have_v = false, parked = false;
        uint32_t v = 0;
        if (do_org) {
            // =========================== organic draw, answered by the user's memo ===========================
            const bool is_o = st == RG_STATE_ORGANIC;
            const size_t row = is_o ? slot : d.n_cap;
            const float4* hp = reinterpret_cast<const float4*>(d.walk_hot + row * 32);
            const float4 h0 = hp[0];
            const uint32_t n_hot = __builtin_bit_cast(uint32_t, h0.w);
            const double u_org = d.u_override ? d.u_override[slot] : rg_uniform(w.w[0], w.w[1]);
            float uf = static_cast<float>(u_org), u_dn = uf, u_up = uf;
            if (static_cast<double>(uf) > u_org) u_dn = f32_down(uf);
            if (static_cast<double>(uf) < u_org) u_up = f32_up(uf);
            bool hit = false;
            if RG_WALK_ABL(23) { hit = true; v = (user + (t & 7u)) % d.P; }     // timing experiment: every draw a memo hit, no row read
            else {
                // the whole line in one round trip (the entries behind n_hot are not looked at), selects only
                float e[28];
#pragma unroll
                for (int i = 1; i < 8; ++i) {
                    const float4 x = hp[i];
                    e[4 * i - 4] = x.x; e[4 * i - 3] = x.y; e[4 * i - 2] = x.z; e[4 * i - 1] = x.w;
                }
#pragma unroll
                for (int j = 0; j < kHotEntries; ++j) {
                    const bool in = static_cast<uint32_t>(j) < n_hot && e[3 * j + 1] < u_dn && u_up < e[3 * j + 2];
                    hit = hit || in;
                    v = in ? __builtin_bit_cast(uint32_t, e[3 * j]) : v;
                }
            }
            have_v = is_o && hit;
            if (is_o && !hit) st = kWSlow;
            c_hit += static_cast<uint32_t>(__popcll(__ballot(have_v)));
        }
        if (do_srch) {
            // =========================== organic draw by the search over the user's prefix sums ===========================
            const bool is_s = st == kWSlow;
            const bool search = is_s;            // (a parked draw too: its chunk is where the anchored certificate starts)
            const size_t row = search ? slot : d.n_cap;
            const float4* hp = reinterpret_cast<const float4*>(d.walk_hot + row * 32);
            const float4 h0 = hp[0];
            const double S = static_cast<double>(h0.x), delta = static_cast<double>(h0.y);
            const float Q = h0.z;
            const uint32_t n_hot = __builtin_bit_cast(uint32_t, h0.w);
            const double u_org = d.u_override ? d.u_override[slot] : rg_uniform(w.w[0], w.w[1]);
            const double tau = u_org * S;
            const float tauf = static_cast<float>(tau);
            // ---- super-chunk: the prefixes <= tau (an unused entry is +inf) ----
            uint32_t sc_star = 0;
            float pbf = 0.0f;
            {
                const float4* sp = reinterpret_cast<const float4*>(d.walk_scp + row * kMaxSC);
#pragma unroll
                for (int i = 0; i < kMaxSC / 4; ++i) {
                    const float4 x = sp[i];
                    const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (xs[q] <= tauf) { sc_star += 1; pbf = fmaxf(pbf, xs[q]); }
                }
            }
            bool found = sc_star < d.n_sc;
            sc_star = min(sc_star, d.n_sc - 1);
            // ---- chunk inside it ----
            uint32_t c_star;
            {
                const uint32_t c0 = sc_star * d.sc_chunks, c1 = min(c0 + d.sc_chunks, d.n_chunks);
                const float* cp = d.cache_chunk + row * d.n_chunks;
                uint32_t cnt = 0;
                for (uint32_t cb = c0; cb < c1; cb += 16) {
                    float4 w4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        w4[i] = cb + 4 * i < c1 ? *reinterpret_cast<const float4*>(cp + cb + 4 * i) : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xs[4] = {w4[i].x, w4[i].y, w4[i].z, w4[i].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (xs[q] <= tauf) { cnt += 1; pbf = fmaxf(pbf, xs[q]); }
                    }
                }
                found = found && cnt < c1 - c0;
                c_star = min(c0 + cnt, c1 - 1);
            }
            const double pb = static_cast<double>(pbf);
            const float rem = static_cast<float>(tau - pb);
            // ---- the 32 products of a chunk: eight users per pass, eight lanes per user, four products per lane.  For every
            // lane that wants it: mboxf[lane] = {index in the chunk of the first product whose fp32 prefix exceeds rem_f, the
            // prefix before it, the prefix with it}, or {-1, chunk total, chunk total} ----
            auto chunk_pass = [&](bool want, uint32_t chunk, float rem_f) {
                const int grp = lane >> 3, gl = lane & 7;
                unsigned long long todo = __ballot(want);
                while (todo) {
                    int src = -1;
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int bit = todo ? __builtin_ctzll(todo) : -1;
                        if (g == grp) src = bit;
                        if (todo) todo &= todo - 1;
                    }
                    const bool has = src >= 0;
                    const int s2 = has ? src : 0;
                    const uint32_t cs = static_cast<uint32_t>(__shfl(static_cast<int>(chunk), s2));
                    const float Qs = __shfl(Q, s2);
                    const float rems = __shfl(rem_f, s2);                // what is left of u S~ at the chunk's start
                    const float4* gp = reinterpret_cast<const float4*>(d.gamma32t + (static_cast<size_t>(cs) * K2) * 32) + gl;
                    float4 l = *(reinterpret_cast<const float4*>(d.mu32 + cs * 32) + gl);
#pragma unroll
                    for (int kh = 0; kh < K2; kh += KH) {
                        float4 gk[KH];
#pragma unroll
                        for (int k = 0; k < KH; ++k) gk[k] = gp[(kh + k) * 8];
#pragma unroll
                        for (int k = 0; k < KH; ++k) {
                            const float wk = __shfl(om[kh + k], s2);
                            l.x = fmaf(gk[k].x, wk, l.x); l.y = fmaf(gk[k].y, wk, l.y);
                            l.z = fmaf(gk[k].z, wk, l.z); l.w = fmaf(gk[k].w, wk, l.w);
                        }
                        asm volatile("" : "+v"(l.x), "+v"(l.y), "+v"(l.z), "+v"(l.w));
                    }
                    const float e0 = __builtin_amdgcn_exp2f(fmaf(l.x, kLog2e, -Qs)), e1 = __builtin_amdgcn_exp2f(fmaf(l.y, kLog2e, -Qs));
                    const float e2 = __builtin_amdgcn_exp2f(fmaf(l.z, kLog2e, -Qs)), e3 = __builtin_amdgcn_exp2f(fmaf(l.w, kLog2e, -Qs));
                    const float q0 = e0, q1 = q0 + e1, q2 = q1 + e2, q3 = q2 + e3;
                    float inc = q3;
#pragma unroll
                    for (int o2 = 1; o2 < 8; o2 <<= 1) {
                        const float y = __shfl_up(inc, o2, 8);
                        if (gl >= o2) inc += y;
                    }
                    float ex = __shfl_up(inc, 1, 8);
                    if (gl == 0) ex = 0.0f;
                    // the product in fp32 (which product is only a proposal: the certificate is taken in float64 from the two
                    // prefixes around it and rejects a wrong one)
                    const float x0 = ex + q0, x1 = ex + q1, x2 = ex + q2, x3 = ex + q3;
                    const int j0 = x0 > rems ? 0 : x1 > rems ? 1 : x2 > rems ? 2 : x3 > rems ? 3 : -1;
                    const unsigned long long hits = __ballot(has && j0 >= 0);
                    const uint32_t gmask = static_cast<uint32_t>(hits >> (8 * grp)) & 0xFFu;
                    if (has) {
                        if (gmask) {
                            if (gl == __builtin_ctz(gmask)) {
                                mboxf[src * 3] = static_cast<float>(4 * gl + j0);
                                mboxf[src * 3 + 1] = j0 == 0 ? ex : j0 == 1 ? x0 : j0 == 2 ? x1 : x2;
                                mboxf[src * 3 + 2] = j0 == 0 ? x0 : j0 == 1 ? x1 : j0 == 2 ? x2 : x3;
                            }
                        } else if (gl == 7) { mboxf[src * 3] = -1.0f; mboxf[src * 3 + 1] = inc; mboxf[src * 3 + 2] = inc; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            };
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            chunk_pass(search, c_star, rem);
            bool ok = false;
            if (search) {
                const int idx = static_cast<int>(mboxf[lane * 3]);
                const CertLin ct = cert_correlated(S, pb, static_cast<double>(mboxf[lane * 3 + 1]), static_cast<double>(mboxf[lane * 3 + 2]), delta);
                v = c_star * 32 + static_cast<uint32_t>(max(idx, 0));
                const bool lo_ok = v == 0 || u_org * ct.den_lo > ct.num_lo;
                const bool hi_ok = v == d.P - 1 || u_org * ct.den_hi < ct.num_hi;
                ok = found && idx >= 0 && v < d.P && ct.valid && lo_ok && hi_ok;
                if (ok && n_hot < static_cast<uint32_t>(kHotEntries)) {
                    // memoise the certified u-interval of v, rounded inwards (and a hair more for the float64 roundings of
                    // the inequality above): u in (lo, hi) implies both conditions, whatever u
                    float lo = -1.0f, hi = 2.0f;
                    if (v != 0) {
                        const double x = ct.num_lo / ct.den_lo * (1.0 + 1e-14);
                        lo = static_cast<float>(x);
                        if (static_cast<double>(lo) < x) lo = f32_up(lo);
                    }
                    if (v != d.P - 1) {
                        const double x = ct.num_hi / ct.den_hi * (1.0 - 1e-14);
                        hi = static_cast<float>(x);
                        if (static_cast<double>(hi) > x) hi = f32_down(hi);
                    }
                    float* hf = d.walk_hot + static_cast<size_t>(slot) * 32;
                    hf[4 + 3 * n_hot] = __builtin_bit_cast(float, v);
                    hf[5 + 3 * n_hot] = lo;
                    hf[6 + 3 * n_hot] = hi;
                    reinterpret_cast<uint32_t*>(hf)[3] = n_hot + 1u;
                }
            }
            // ---- uncertified: with the user's float64 sums, or park the user until they exist ----
            const bool need64 = is_s && !ok;
            const bool have64 = need64 && d.f64_valid[slot] != 0;
            parked = need64 && !have64;
            // ANCHORED certificate.  The float64 sums of a user (k_exact_prefix left them as prefixes at the end of every 64
            // products, on the same reference Q as the fp32 exps) pin the prefix at the start of the draw's 64-product chunk to
            // ~1e-13 S; only the part INSIDE the chunk is fp32, so the same test with delta applied to that part alone — and
            // 1e-12 S of slack for the anchors' own roundings — certifies all but ~2.5 % of the draws the plain certificate
            // rejected (those need a heavy product earlier in the same chunk).  Lane-parallel, like the search: what is left
            // for the wave-serial float64 pick below is ~0.1 % of the organic draws instead of 3 %.
            bool got64 = false;
            if (__ballot(have64)) {
                const double* pfx = d.exact_sums + static_cast<size_t>(have64 ? slot : 0u) * n_cc;
                const double S64 = pfx[n_cc - 1];
                const double target = u_org * S64;
                uint32_t cc = min(c_star >> 1, n_cc - 1u);               // the fp32 search's chunk is (nearly always) the float64 one
                double hi64 = pfx[cc], lo64 = cc ? pfx[cc - 1] : 0.0;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    if (!(lo64 <= target) && cc > 0u) { --cc; hi64 = lo64; lo64 = cc ? pfx[cc - 1] : 0.0; }
                    else if (!(target < hi64) && cc + 1u < n_cc) { ++cc; lo64 = hi64; hi64 = pfx[cc]; }
                }
                const bool anchored = have64 && lo64 <= target && target < hi64;
                const float rem1 = static_cast<float>(target - lo64);
                chunk_pass(anchored, 2u * cc, rem1);
                const int idx1 = anchored ? static_cast<int>(mboxf[lane * 3]) : 0;
                const float a1 = mboxf[lane * 3 + 1], b1 = mboxf[lane * 3 + 2];
                const bool in2 = anchored && idx1 < 0;
                __builtin_amdgcn_wave_barrier();
                chunk_pass(in2, 2u * cc + 1u, rem1 - a1);
                if (anchored) {
                    int ix = idx1;
                    float fa = a1, fb = b1;
                    uint32_t va = 64u * cc + static_cast<uint32_t>(max(idx1, 0));
                    if (in2) {
                        ix = static_cast<int>(mboxf[lane * 3]);
                        fa = a1 + mboxf[lane * 3 + 1]; fb = a1 + mboxf[lane * 3 + 2];
                        va = 64u * cc + 32u + static_cast<uint32_t>(max(ix, 0));
                    }
                    // (a computed in-chunk prefix s = e (1 + eps), |eps| <= delta: the true e is at most s / (1 - delta) <=
                    // s (1 + dp), dp = delta (1 + 2 delta) as in cert_correlated, and at least s / (1 + delta) >= s (1 - delta))
                    const double slack = 1.0e-12 * S64;
                    const double dp = delta * (1.0 + 2.0 * delta);
                    const bool lo_ok = va == 0u || lo64 + static_cast<double>(fa) * (1.0 + dp) + slack < target;
                    const bool hi_ok = va == d.P - 1 || target + slack < lo64 + static_cast<double>(fb) * (1.0 - delta);
                    got64 = ix >= 0 && va < d.P && lo_ok && hi_ok;
                    if (got64) v = va;
                }
            }
            c_anch += static_cast<uint32_t>(__popcll(__ballot(got64)));
            c_pick += static_cast<uint32_t>(__popcll(__ballot(have64)));     // resolved with float64 sums: anchored or picked
            unsigned long long picks = __ballot(have64 && !got64);
            while (picks) {
                const int L = __builtin_ctzll(picks);
                picks &= picks - 1;
                const uint32_t s_slot = static_cast<uint32_t>(__shfl(static_cast<int>(slot), L));
                const double s_u = __shfl(u_org, L);
                const double M = static_cast<double>(d.exact_ref[s_slot]) * 0.69314718055994530942;
                const uint32_t pv = exact_pick_pfx(d, d.exact_sums + static_cast<size_t>(s_slot) * n_cc,
                                                   d.omega + static_cast<size_t>(s_slot) * d.OMS, M, s_u, lane);
                if (lane == L) v = pv;
                __builtin_amdgcn_wave_barrier();
            }
            if (parked) {
                d.park_t[slot] = t | (static_cast<uint32_t>(RG_STATE_ORGANIC) << 24) | (1u << 27);
                d.exact_ref[slot] = Q;
            }
            have_v = is_s && !parked;
            if (have_v) pend = false;
        }
        // ---- park list entries for the users parked in this step ----
        const unsigned long long pmask = __ballot(parked);
        if (pmask) {
            const uint32_t np = static_cast<uint32_t>(__popcll(pmask));
            if (park_next + np > park_end) {
                for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
                uint32_t base = 0;
                if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_park, 64ull));
                base = __builtin_amdgcn_readfirstlane(base);
                park_next = base; park_end = base + 64;
            }
            flush_hist(parked && hdirty);
            if (parked) { d.park_list[out_base + park_next + prefix_in_mask(pmask)] = slot; st = kEmpty; }
            park_next += np;
        }
        // =========================== bandit event: the policy's act and the click ===========================
        bool is_ban = do_ban && (do_clk ? st == kWClick : st == RG_STATE_BANDIT);
        const bool is_ph = do_ban && !do_clk && st == kPhantom;
        double ps = 1.0;
        uint32_t a = 0;
        bool click = false, click_known = false;
        double ctr = 0.0;
        if (do_ban) {
            if ((is_ban || is_ph) && RG_WALK_ABL(28)) { a = user % d.P; ps = 1.0; }      // timing experiment: no policy act
            else
            if (is_ban || is_ph) {
                if (HIST == 2) {
                    // the same act on the COMPACT line (prefix form): the first product whose cumulative count exceeds u x views
                    // = the number of words below the key (Thi + 1) << 16 — the last word of each 8-word segment, then the segment
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, t, 0, RG_DRAW_POLICY);
                    const double u1 = rg_uniform(pw.w[2], pw.w[3]);
                    const hent_t* hr = hist_row(d, slot);
                    const uint32_t* hw32 = reinterpret_cast<const uint32_t*>(hl);      // word w: hw32[(w >> 1) * 128 + (w & 1)]
                    const uint32_t h0 = hw32[0];
                    const uint32_t p7 = hw32[3 * 128 + 1], p15 = hw32[7 * 128 + 1], p23 = hw32[11 * 128 + 1], p31 = hw32[15 * 128 + 1];
                    const uint32_t nd = h0 & 0x7FFFu;
                    const double sum = static_cast<double>(h0 >> 15);
                    const double T = u1 * sum;
                    const uint32_t Thi = static_cast<uint32_t>(fmin(floor(T * (1.0 + 0x1p-36)), 4294967295.0));
                    const uint32_t Tlo = static_cast<uint32_t>(fmin(ceil(T * (1.0 - 0x1p-36)), 4294967295.0));
                    const bool over = Thi >= 65535u;                                     // (u x views at the top of the range: no entry exceeds it)
                    const uint32_t khi = over ? 0u : (Thi + 1u) << 16;                   // prefix <= Thi  <=>  word < khi
                    const uint32_t seg = (p7 < khi ? 1u : 0u) + (p15 < khi ? 1u : 0u) + (p23 < khi ? 1u : 0u);
                    uint32_t x[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const hent_t y = hl[(seg * 4 + j) * 64];
                        x[2 * j] = static_cast<uint32_t>(y); x[2 * j + 1] = static_cast<uint32_t>(y >> 32);
                    }
                    if (seg == 0u) x[0] = 0u;                                             // (the header: counted, prefix 0)
                    uint32_t in_seg = 0u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) in_seg += x[j] < khi ? 1u : 0u;
                    const uint32_t idx = seg * 8u + in_seg;                              // first entry with prefix > Thi (32: none in the line)
                    // its word and the one before it (the entry before a segment's first: the segment end read above)
                    uint32_t w_at = 0xFFFFFFFFu, w_prev = seg == 0u ? 0u : (seg == 1u ? p7 : (seg == 2u ? p15 : p23));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        w_at = in_seg == static_cast<uint32_t>(j) ? x[j] : w_at;
                        w_prev = in_seg == static_cast<uint32_t>(j + 1) ? x[j] : w_prev;
                    }
                    bool found = !over && idx <= nd && idx < kHcLine;
                    // an entry inside the 2^-36 band of u x views: only the last one at or below Thi can be (prefixes ascend)
                    bool amb = idx >= 2u && (w_prev >> 16) >= Tlo;
                    uint32_t c_f = (w_at >> 16) - (idx >= 2u ? (w_prev >> 16) : 0u);
                    a = w_at & 0xFFFFu;
                    uint32_t C = p31 >> 16;                                              // (nd >= 31: the line's last prefix)
                    for (uint32_t base = kHcLine; base <= nd && !found; base += kHistRegs) {      // longer histories: from the row
                        hent_t f[kHistRegs];
                        hist_load_line(hr + base, f);
#pragma unroll
                        for (int i = 0; i < kHistRegs; ++i)
                            if (base + i <= nd && !found) {
                                C += h_cnt(f[i]);
                                if (C > Thi) { found = true; a = h_prod(f[i]); c_f = h_cnt(f[i]); }
                                else if (C >= Tlo) amb = true;
                            }
                    }
                    if (found && !amb) ps = static_cast<double>(c_f) / sum;
                    else {
                        // inside the band (~1e-10 of the acts): numpy's arithmetic over the viewed products, as below
                        auto ent = [&](uint32_t i, uint32_t* prev) -> hent_t {       // (product, count) of entry i, walked in order
                            if (i >= kHcLine) return hr[i];
                            const uint32_t w = hw32[(i >> 1) * 128 + (i & 1u)];
                            const uint32_t cnt = (w >> 16) - *prev;
                            *prev = w >> 16;
                            return (static_cast<hent_t>(w & 0xFFFFu) << 32) | cnt;
                        };
                        double last = 0.0;
                        uint32_t pv = 0u;
                        for (uint32_t i = 1; i <= nd; ++i) last += static_cast<double>(h_cnt(ent(i, &pv))) / sum;
                        double acc = 0.0, pa = 0.0;
                        a = d.P - 1;
                        bool fnd = false;
                        pv = 0u;
                        for (uint32_t i = 1; i <= nd && !fnd; ++i) {
                            const hent_t y = ent(i, &pv);
                            const double p = static_cast<double>(h_cnt(y)) / sum;
                            acc += p;
                            if (!(acc / last <= u1)) { a = h_prod(y); pa = p; fnd = true; }
                        }
                        ps = pa;
                    }
                } else
                if (HIST) {
                    // OrganicUserEventCounterModel.act (organic_user_count.py:45-96; exploit_explore, epsilon = 0,
                    // select_randomly: the host instantiates HIST = 1 for this form only) on the history line in LDS: decided
                    // by integer prefix counts outside a 2^-36 band (see policy_act), by the float64 cdf walk inside it
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, t, 0, RG_DRAW_POLICY);
                    const double u1 = rg_uniform(pw.w[2], pw.w[3]);
                    const hent_t h0 = hl[0];
                    const uint32_t nd = h_cnt(h0);
                    const double sum = static_cast<double>(h_prod(h0));
                    const hent_t* hr = hist_row(d, slot);
                    const double T = u1 * sum;
                    const uint32_t Thi = static_cast<uint32_t>(fmin(floor(T * (1.0 + 0x1p-36)), 4294967295.0));
                    const uint32_t Tlo = static_cast<uint32_t>(fmin(ceil(T * (1.0 - 0x1p-36)), 4294967295.0));
                    uint32_t C = 0, c_f = 0;
                    bool found = false, amb = false;
                    {
                        // the 15 entries of the line at once (one LDS round trip), then selects only: an entry-by-entry loop
                        // with its two exits compiled to 15 dependent round trips and 30 branches
                        hent_t e[16];
#pragma unroll
                        for (int i = 1; i < 16; ++i) e[i] = hl[i * 64];
#pragma unroll
                        for (int i = 1; i < 16; ++i) {
                            const bool in = static_cast<uint32_t>(i) <= nd;
                            const uint32_t cnt = in ? h_cnt(e[i]) : 0u;
                            C += cnt;
                            const bool take = in && !found && C > Thi;
                            amb = amb || (in && !found && !take && C >= Tlo);
                            a = take ? h_prod(e[i]) : a;
                            c_f = take ? cnt : c_f;
                            found = found || take;
                        }
                    }
                    for (uint32_t base = 16; base <= nd && !found; base += kHistRegs) {      // longer histories: from the row
                        hent_t f[kHistRegs];
                        hist_load_line(hr + base, f);
#pragma unroll
                        for (int i = 0; i < kHistRegs; ++i)
                            if (base + i <= nd && !found) {
                                C += h_cnt(f[i]);
                                if (C > Thi) { found = true; a = h_prod(f[i]); c_f = h_cnt(f[i]); }
                                else if (C >= Tlo) amb = true;
                            }
                    }
                    if (found && !amb) ps = static_cast<double>(c_f) / sum;
                    else {
                        // inside the band (~1e-10 of the acts): numpy's arithmetic — p_i = count_i / sum, cdf = cumsum(p) / last,
                        // first index with cdf > u1 — over the viewed products (zero entries add exactly 0.0)
                        double last = 0.0;
                        for (uint32_t i = 1; i <= nd; ++i) last += static_cast<double>(h_cnt(i < 16 ? hl[i * 64] : hr[i])) / sum;
                        double acc = 0.0, pa = 0.0;
                        a = d.P - 1;
                        bool fnd = false;
                        for (uint32_t i = 1; i <= nd && !fnd; ++i) {
                            const hent_t x = i < 16 ? hl[i * 64] : hr[i];
                            const double p = static_cast<double>(h_cnt(x)) / sum;
                            acc += p;
                            if (!(acc / last <= u1)) { a = h_prod(x); pa = p; fnd = true; }
                        }
                        ps = pa;
                    }
                } else if (d.policy == RG_POLICY_LAST_VIEW_TABLE) {
                    const uint32_t p = d.lpv[slot];
                    ps = d.pol_ps ? static_cast<double>(d.pol_ps[p]) : 1.0;
                    a = static_cast<uint32_t>(d.pol_table[p]);
                } else {        // agent = None / RandomAgent: uniform over P from the env / the agent stream
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, t, 0, RG_DRAW_POLICY);
                    ps = 1.0 / static_cast<double>(d.P);
                    a = rg_bounded(pw.w[0], pw.w[1], d.P);
                }
            }
            if (is_ph) {       // final step_offline(done = True): the act above, reward 0 (abstract.py:223-233,311-316)
                rg_event e;
                e.u = user; e.t = t; e.code = RG_EV_BANDIT | RG_EV_PHANTOM | a;
                e.ps = static_cast<float>(ps);
                d.phantom[slot] = e;
                d.phantom_ps[slot] = ps;
                d.has_phantom[slot] = 1;
                st = kEmpty;
            }
            flush_hist(is_ph && hdirty);
            c_ph += static_cast<uint32_t>(__popcll(__ballot(is_ph)));
            if (is_ban && RG_WALK_ABL(24)) { click = false; click_known = true; }        // timing experiment: no beta row
            else
            if (is_ban && !d.aux_pclick && !do_clk) {
                // no click below kNoClickBelow; the 3 % above it wait (kWClick) until walk_click_batch lanes of the wave do: the
                // beta row is a memory round trip the whole wave would otherwise sit out in every bandit iteration
                if (rg_uniform(w.w[0], w.w[1]) < kNoClickBelow) { click = false; click_known = true; }
                else if (d.walk_click_batch) { st = kWClick; is_ban = false; }
            }
            if (is_ban && !d.aux_pclick && !click_known) {
                const int dec = click_decide32<KC>(d.beta32 + static_cast<size_t>(a) * d.KB4, [&](int k) { return om[k]; }, d.K, d.KB4,
                                                   static_cast<float>(d.mu_b[a]), rg_uniform(w.w[0], w.w[1]));
                if (dec >= 0) { click = dec != 0; click_known = true; }
            }
            if (is_ban && !click_known) {
                const double* b = d.beta + static_cast<size_t>(a) * d.K;
                const double* omd = d.omega + static_cast<size_t>(slot) * d.OMS;
                double x = 0.0;
                for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {
                    double wv[8], bv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t k = min(k0 + i, d.K - 1);
                        wv[i] = omd[k];
                        bv[i] = b[k];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (k0 + i < d.K) x += bv[i] * wv[i];
                }
                ctr = ff64(x + d.mu_b[a]);
                const double p0 = 1.0 - ctr;
                click = (p0 / (p0 + ctr)) <= rg_uniform(w.w[0], w.w[1]);
            }
        }
        // =========================== the event's row, the view, the transition ===========================
        const bool ev = have_v || is_ban;
        const unsigned long long rowm = __ballot(ev);
        if (rowm) {
            const uint32_t nrow = static_cast<uint32_t>(__popcll(rowm));
            if (row_next + nrow > row_end) {
                for (uint64_t r = row_next + lane; r < row_end; r += 64)
                    if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&d.counters[kCntTailRows], static_cast<unsigned long long>(chunk_rows));
                base = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base >> 32))) << 32) |
                       __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base));
                row_next = base; row_end = base + chunk_rows;
            }
            const uint64_t my_row = row_next + prefix_in_mask(rowm);
            row_next += nrow;
            if (ev && d.log && my_row < d.log_cap && !RG_WALK_ABL(25)) {
                rg_event e;
                e.u = user; e.t = t;
                e.code = have_v ? v : (RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a);
                e.ps = have_v ? __builtin_nanf("") : static_cast<float>(ps);
                d.log[my_row] = e;
                if (is_ban && d.aux_ps) d.aux_ps[my_row] = ps;
                if (is_ban && d.aux_pclick) d.aux_pclick[my_row] = ctr;
            }
            c_org += static_cast<uint32_t>(__popcll(__ballot(have_v)));
            c_ban += static_cast<uint32_t>(__popcll(__ballot(is_ban)));
            c_clicks += static_cast<uint32_t>(__popcll(__ballot(is_ban && click)));
            if (have_v) {
                if (d.lpv) d.lpv[slot] = v;
                if (HIST == 2 && !RG_WALK_ABL(27)) {
                    // ViewsFeaturesProvider.observe (agents/abstract.py:347-358) on the COMPACT line in LDS: position and hit of v,
                    // then the line with the prefixes from there on raised by the view (and shifted by the new product)
                    hent_t* hr = hist_row(d, slot);
                    uint32_t e[kHcLine];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const hent_t y = hl[i * 64];
                        e[2 * i] = static_cast<uint32_t>(y); e[2 * i + 1] = static_cast<uint32_t>(y >> 32);
                    }
                    const uint32_t h0 = e[0];
                    const uint32_t nd = h0 & 0x7FFFu;
                    uint32_t pos = 1;                       // first entry with product >= v (min(nd, 31) + 1 if none)
                    bool hit = false;
#pragma unroll
                    for (int i = 1; i < static_cast<int>(kHcLine); ++i) {
                        const uint32_t pl = e[i] & 0xFFFFu;                   // (an unused word: 0xFFFF, above every product)
                        pos += pl < v ? 1u : 0u;
                        hit = hit || pl == v;
                    }
                    uint32_t* hw32 = reinterpret_cast<uint32_t*>(hl);           // word w of this lane's line: hw32[(w >> 1) * 128 + (w & 1)]
                    const bool room = nd < kHcLine - 1u;
                    if (hit || room) {
                        const bool full = !hit && nd + 1 >= d.hist_cap;
                        if (full) atomicAdd(&d.counters[RG_CNT_HIST_OVERFLOW], 1ull);
                        // new word i: below pos unchanged; at pos the product's own (raised, or new: the prefix before it + 1);
                        // above it the old word (hit) or the old word below (new product), raised by the view
                        uint32_t f[kHcLine];
                        f[0] = full ? h0 : h0 + (1u << 15) + (hit ? 0u : 1u);
                        const uint32_t last = hit ? nd : nd + 1u;              // entries in use after the view
#pragma unroll
                        for (int i = 1; i < static_cast<int>(kHcLine); ++i) {
                            const uint32_t ui = static_cast<uint32_t>(i);
                            const uint32_t below = i == 1 ? 0u : e[i - 1];
                            const uint32_t src = hit ? e[i] : (ui == pos ? ((below & 0xFFFF0000u) | v) : below);
                            const uint32_t raised = src + 0x10000u;
                            f[i] = (full || ui < pos) ? e[i] : (ui <= last ? raised : 0xFFFFFFFFu);
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) hl[i * 64] = static_cast<hent_t>(f[2 * i]) | (static_cast<hent_t>(f[2 * i + 1]) << 32);
                        hdirty = true;
                    } else if (v > (e[kHcLine - 1] & 0xFFFFu)) {
                        // a longer history, v behind the line's 31 products: entries >= 32 of the row (always current), the
                        // header in LDS — one round trip, nothing to read back
                        if (nd + 1 >= d.hist_cap) {
                            flush_hist(hdirty);
                            hdirty = false;
                            history_add(d, slot, v);
                            const hent_t hh = hr[0];
                            hw32[0] = (h_prod(hh) << 15) | h_cnt(hh);
                        } else {
                            const uint32_t fresh = history_tail_add(hr, nd, v, kHcLine);
                            hw32[0] = h0 + (1u << 15) + fresh;
                            hdirty = true;
                        }
                    } else {
                        // a new product inside the line of a longer history (its last product moves to the row): the general
                        // insertion on the row, then the line again
                        flush_hist(hdirty);
                        hdirty = false;
                        history_add(d, slot, v);
                        load_compact(slot);
                    }
                } else
                if (HIST && !RG_WALK_ABL(27)) {
                    // ViewsFeaturesProvider.observe (agents/abstract.py:347-358) on the line in LDS, written through to the row
                    hent_t* hr = hist_row(d, slot);
                    const hent_t h0 = hl[0];
                    const uint32_t nd = h_cnt(h0);
                    const hent_t key = static_cast<hent_t>(v) << 32;
                    // the line in registers (one LDS round trip): position of v, whether it is there
                    hent_t e[17];
                    e[0] = h0; e[16] = 0ull;
#pragma unroll
                    for (int i = 1; i < 16; ++i) e[i] = hl[i * 64];
                    uint32_t pos = 1;                       // first entry with product >= v (min(nd, 15) + 1 if none)
                    bool hit = false;
#pragma unroll
                    for (int i = 1; i < 16; ++i) {
                        const bool in = static_cast<uint32_t>(i) <= nd;
                        pos += (in && e[i] < key) ? 1u : 0u;
                        hit = hit || (in && h_prod(e[i]) == v);
                    }
                    if (nd < 15u && RG_WALK_ABL(29)) {}                        // timing experiment: no insertion into the line
                    else if (nd >= 15u && RG_WALK_ABL(26)) {}                  // timing experiment: no insertion into a longer history
                    else
                    if (nd < 15u || hit) {
                        // the new line by selects, written back whole to LDS (the row gets it when the lane lets go of the
                        // user): no data-dependent branch, no dependent loads.  A longer history whose line holds v: the same
                        const bool full = !hit && nd + 1 >= d.hist_cap;
                        if (full) atomicAdd(&d.counters[RG_CNT_HIST_OVERFLOW], 1ull);
                        hent_t f[16];
                        f[0] = full ? h0 : h0 + (1ull << 32) + (hit ? 0ull : 1ull);
#pragma unroll
                        for (int i = 1; i < 16; ++i) {
                            const uint32_t ui = static_cast<uint32_t>(i);
                            const hent_t shifted = ui < pos ? e[i] : (ui == pos ? (key | 1ull) : e[i - 1]);
                            const hent_t bumped = ui == pos ? e[i] + 1ull : e[i];
                            f[i] = full ? e[i] : (hit ? bumped : shifted);
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) hl[i * 64] = f[i];
                        hdirty = true;                                         // (written back when the lane lets go of the user)
                    } else if (v > h_prod(e[15])) {
                        // a longer history, v behind the line's 15 products: entries >= 16 of the row (always current), the
                        // header in LDS — one round trip, nothing to read back
                        if (nd + 1 >= d.hist_cap) {
                            // (a new product would not fit: the general insertion decides and counts the overflow)
                            flush_hist(hdirty);
                            hdirty = false;
                            history_add(d, slot, v);
                            hl[0] = hr[0];
                        } else {
                            const uint32_t fresh = history_tail_add(hr, nd, v);
                            hl[0] = h0 + (1ull << 32) + fresh;
                            hdirty = true;
                        }
                    } else {
                        // a new product inside the line of a longer history (its last product moves to the row): the general
                        // insertion on the row, then the line again
                        flush_hist(hdirty);
                        hdirty = false;
                        history_add(d, slot, v);
                        const ulonglong2* hr2 = reinterpret_cast<const ulonglong2*>(hr);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const ulonglong2 x = hr2[i];
                            hl[(2 * i) * 64] = x.x; hl[(2 * i + 1) * 64] = x.y;
                        }
                    }
                }
            }
            if (ev) {
                const double u_trans = rg_uniform(w.w[2], w.w[3]);
                const double c0 = have_v ? d.cdf_o0 : d.cdf_b0, c1 = have_v ? d.cdf_o1 : d.cdf_b1;
                int ns = (c0 <= u_trans) + (c1 <= u_trans);
                if (click) ns = RG_STATE_ORGANIC;                  // abstract.py:180-181 (sigma_omega == 0: no drift to apply)
                const bool organic_only = (d.first_user + slot) < d.organic_only_below;
                bool limit = false;
                if (organic_only && ns != RG_STATE_ORGANIC) {
                    ns = RG_STATE_STOP;
                    d.n_events[slot] = t + 1;
                } else if (ns == RG_STATE_STOP) {
                    d.n_events[slot] = t + 1;
                    ns = kPhantom;                                 // the phantom row's act: this lane's next bandit iteration
                } else if (t + 2 >= kMaxSteps) {
                    ns = RG_STATE_STOP;
                    d.n_events[slot] = t + 1;
                    limit = true;
                }
                const unsigned long long endm = __ballot(ns == RG_STATE_STOP || ns == kPhantom);
                (void)endm;
                if (ns == RG_STATE_STOP || ns == kPhantom) {
                    // (maximum over the wave taken once at the end: a per-lane maximum in one register)
                    c_maxt = max(c_maxt, t + 1);
                }
                if (limit) c_limit += 1;
                flush_hist(ns == RG_STATE_STOP && hdirty);
                if (ns == RG_STATE_STOP) st = kEmpty;
                else { st = ns; t += 1; }
            }
        }
    }
    // ---- leftovers of the reserved chunks, counters ----
    {
        const DevSim& d = *(const DevSim*)kargs;
        for (uint64_t r = row_next + lane; r < row_end; r += 64)
            if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
        for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
        for (int o = 32; o > 0; o >>= 1) {
            c_maxt = max(c_maxt, static_cast<uint32_t>(__shfl_xor(static_cast<int>(c_maxt), o)));
            c_limit += static_cast<uint32_t>(__shfl_xor(static_cast<int>(c_limit), o));
        }
        if (lane == 0) {
            if (c_org) atomicAdd(&d.counters[kCntTailOrganic], static_cast<unsigned long long>(c_org));
            if (c_ban) atomicAdd(&d.counters[kCntTailBandit], static_cast<unsigned long long>(c_ban));
            if (c_clicks) atomicAdd(&d.counters[RG_CNT_CLICKS], static_cast<unsigned long long>(c_clicks));
            if (c_ph) atomicAdd(&d.counters[RG_CNT_PHANTOM], static_cast<unsigned long long>(c_ph));
            if (c_pick) atomicAdd(&d.counters[RG_CNT_EXACT_DRAWS], static_cast<unsigned long long>(c_pick));
            if (c_sweeps) atomicAdd(&d.counters[RG_CNT_EXACT_SWEEPS], static_cast<unsigned long long>(c_sweeps));
            if (c_hit) atomicAdd(&d.counters[kCntWalkHits], static_cast<unsigned long long>(c_hit));
            if (c_anch) atomicAdd(&d.counters[RG_CNT_ANCHORED], static_cast<unsigned long long>(c_anch));
            if (c_maxt) atomicMax(&d.counters[kCntTailMaxT], static_cast<unsigned long long>(c_maxt));
            if (c_limit) atomicAdd(&d.counters[kCntTailLimit], static_cast<unsigned long long>(c_limit));
        }
    }
}

#endif
// ------------------------------------------------------------------------------------------
// k_walk_solo — the LAST round of the user-major walk: a WAVE per user, a LANE per consecutive event.
//
// What is left for the last round are the users the draining waves of the earlier rounds handed over: few (some 10^4 of
// 10^7) and long-lived (the longest trajectory of a 10 M-user run has ~1 600 events).  Walked a lane per user, an event per
// wave iteration, their round costs (events of the longest user) x (latency of an iteration, ~7 us) whatever the GPU could
// do meanwhile — a third of the walk on a 2 M-user shard.  But between two organic events nothing a user does depends on
// its own earlier events of the RUN it is in:
//   * organic run: the state chain of organic events is decided by their transition uniforms alone (addressed draws), so
//     the run's length is known up front and its product draws (memo / search / float64 pick) are independent;
//   * bandit run: omega and the view history are fixed, so the policy's act, the click and the transition of the next 64
//     events are evaluated at once and committed up to the first one that leaves the run (a click, a transition).
// A lane takes event t + lane of the user's current run; the wave commits the run's prefix, moves the user past it and
// goes on: ~11 iterations per 100 events instead of 100.  Rows, counters, view history, phantom row: as k_walk2 (the sorted
// log cannot tell the difference; the raw order differs, like between any two forms).  Every listed user has its float64
// sums (the batch between rounds 1 and 2 took them).  Needs hist_cap <= 256 (the user's whole history lives in LDS).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kSoloHist = 256;

// k_walk_solo's view history: the user's whole row in LDS (hs[0] header, hs[1 ..] the entries), every lane an event.
// The OrganicUserEventCounter act of ONE lane's event (its own uniform u1) on the wave's shared history — the integer
// prefix walk of policy_act, the float64 cdf walk inside the 2^-36 band.
__device__ __forceinline__ uint32_t solo_ouc_act(const DevSim& d, const hent_t* hs, double u1, double* ps_out) {
    const hent_t h0 = hs[0];
    const uint32_t nd = h_cnt(h0);
    const double sum = static_cast<double>(h_prod(h0));
    const double T = u1 * sum;
    const uint32_t Thi = static_cast<uint32_t>(fmin(floor(T * (1.0 + 0x1p-36)), 4294967295.0));
    const uint32_t Tlo = static_cast<uint32_t>(fmin(ceil(T * (1.0 - 0x1p-36)), 4294967295.0));
    uint32_t C = 0, a = 0, c_f = 0;
    bool found = false, amb = false;
    for (uint32_t i = 1; i <= nd; ++i) {                 // (wave-uniform trip count, broadcast reads)
        const hent_t x = hs[i];
        C += h_cnt(x);
        const bool take = !found && C > Thi;
        amb = amb || (!found && !take && C >= Tlo);
        a = take ? h_prod(x) : a;
        c_f = take ? h_cnt(x) : c_f;
        found = found || take;
    }
    if (found && !amb) { *ps_out = static_cast<double>(c_f) / sum; return a; }
    double last = 0.0;
    for (uint32_t i = 1; i <= nd; ++i) last += static_cast<double>(h_cnt(hs[i])) / sum;
    double acc = 0.0, pa = 0.0;
    a = d.P - 1;
    bool fnd = false;
    for (uint32_t i = 1; i <= nd && !fnd; ++i) {
        const hent_t x = hs[i];
        const double p = static_cast<double>(h_cnt(x)) / sum;
        acc += p;
        if (!(acc / last <= u1)) { a = h_prod(x); pa = p; fnd = true; }
    }
    *ps_out = pa;
    return a;
}
// ViewsFeaturesProvider.observe (agents/abstract.py:347-358) by the whole wave on the history in LDS, written through to
// the row: position and hit by ballots over the entries (four per lane: nd < 256), the shift by every lane moving its own.
__device__ __forceinline__ void solo_hist_add(const DevSim& d, hent_t* hs, hent_t* hr, uint32_t v, int lane) {
    const hent_t h0 = hs[0];
    const uint32_t nd = h_cnt(h0);
    const hent_t key = static_cast<hent_t>(v) << 32;
    hent_t mine[4];
    uint32_t below = 0;
    bool hit = false;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t j = 1u + static_cast<uint32_t>(lane) + 64u * b;
        mine[b] = j <= nd ? hs[j] : ~0ull;
        below += static_cast<uint32_t>(__popcll(__ballot(j <= nd && mine[b] < key)));
        hit = hit || __ballot(j <= nd && h_prod(mine[b]) == v) != 0ull;
    }
    const uint32_t pos = 1u + below;                      // first entry with product >= v (nd + 1 if none)
    __builtin_amdgcn_wave_barrier();
    if (hit) {
        if (lane == 0) {
            const hent_t x = hs[pos] + 1ull;
            hs[pos] = x; hr[pos] = x;
            hs[0] = h0 + (1ull << 32); hr[0] = h0 + (1ull << 32);
        }
    } else if (nd + 1 >= d.hist_cap || nd + 2 > kSoloHist) {
        if (lane == 0) atomicAdd(&d.counters[RG_CNT_HIST_OVERFLOW], 1ull);
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t j = 1u + static_cast<uint32_t>(lane) + 64u * b;
            if (j >= pos && j <= nd) { hs[j + 1] = mine[b]; hr[j + 1] = mine[b]; }
        }
        if (lane == 0) {
            hs[pos] = key | 1ull; hr[pos] = key | 1ull;
            hs[0] = h0 + (1ull << 32) + 1ull; hr[0] = h0 + (1ull << 32) + 1ull;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

#if RG_HAS(7)
template <int KH, int HIST>
__global__ void __launch_bounds__(kBlock) k_walk_solo(DevSim d_arg, uint32_t n_work, uint32_t chunk_rows, uint32_t in_base) {
    (void)d_arg;       // read where it lies, in the kernel-argument segment (the float64 pick is a call that takes its address)
    const DevSim& d = *(const DevSim*)(const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int K2 = 2 * KH;
    constexpr int KC = ((K2 + 3) / 4) * 4;
    constexpr int kEmpty = 3, kPhantom = 4;
    __shared__ hent_t s_hist[kBlock / 64][HIST ? kSoloHist : 1];      // the user's history row: [0] header, then the entries
    __shared__ float s_mbox[kBlock / 64][64 * 3];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    hent_t* hs = s_hist[wave];
    float* mboxf = s_mbox[wave];
    const uint32_t n_cc = d.PT / 64;
    uint64_t row_next = 0, row_end = 0;
    uint32_t c_org = 0, c_ban = 0, c_clicks = 0, c_ph = 0, c_pick = 0, c_maxt = 0, c_limit = 0, c_hit = 0;
    if (d.q_count) n_work = static_cast<uint32_t>(*d.q_count);
    for (;;) {
        uint32_t idx = 0;
        if (lane == 0) idx = static_cast<uint32_t>(atomicAdd(d.q_ticket, 1ull));
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= n_work) break;
        const uint32_t slot = __builtin_amdgcn_readfirstlane(d.park_list[in_base + idx]);
        if (slot == 0xFFFFFFFFu) continue;
        const uint32_t pt = __builtin_amdgcn_readfirstlane(d.park_t[slot]);
        uint32_t t = pt & 0xFFFFFFu;
        int st = static_cast<int>((pt >> 24) & 7u);
        bool pend = (pt >> 27) & 1u;
        const uint32_t user = static_cast<uint32_t>(d.first_user + slot);
        const bool organic_only = (d.first_user + slot) < d.organic_only_below;
        float om[KC];
        {
            const float4* rp = reinterpret_cast<const float4*>(d.cache_row + static_cast<size_t>(slot) * d.cache_row_f);
#pragma unroll
            for (int k4 = 0; k4 < K2 / 4; ++k4) {
                const float4 x = rp[11 + k4];
                om[4 * k4] = x.x; om[4 * k4 + 1] = x.y; om[4 * k4 + 2] = x.z; om[4 * k4 + 3] = x.w;
            }
#pragma unroll
            for (int k = (K2 / 4) * 4; k < K2; ++k) om[k] = reinterpret_cast<const float*>(rp)[44 + k];
#pragma unroll
            for (int k = K2; k < KC; ++k) om[k] = 0.0f;
        }
        hent_t* hr = HIST ? hist_row(d, slot) : nullptr;
        if (HIST) {
            const uint32_t nd0 = h_cnt(hr[0]);
            for (uint32_t j = lane; j <= nd0; j += 64) hs[j] = hr[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t lastv = d.lpv ? d.lpv[slot] : 0u;
        // hot row of the user (no memo entries are added here: the lanes of a run would race for the row)
        const float4* hp = reinterpret_cast<const float4*>(d.walk_hot + static_cast<size_t>(slot) * 32);
        while (st != kEmpty) {
            const uint32_t te = t + static_cast<uint32_t>(lane);             // this lane's event
            const rg_u32x4 w = rg_draw(d.seed, user, te, 0, RG_DRAW_EVENT);
            const double u_trans = rg_uniform(w.w[2], w.w[3]);
            if (st == kPhantom) {
                // final step_offline(done = True): one more act, reward 0 (abstract.py:223-233,311-316) — lane 0's
                double ps = 1.0;
                uint32_t a = 0;
                if (HIST) {
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, t, 0, RG_DRAW_POLICY);
                    a = solo_ouc_act(d, hs, rg_uniform(pw.w[2], pw.w[3]), &ps);
                } else if (d.policy == RG_POLICY_LAST_VIEW_TABLE) {
                    ps = d.pol_ps ? static_cast<double>(d.pol_ps[lastv]) : 1.0;
                    a = static_cast<uint32_t>(d.pol_table[lastv]);
                } else {
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, t, 0, RG_DRAW_POLICY);
                    ps = 1.0 / static_cast<double>(d.P);
                    a = rg_bounded(pw.w[0], pw.w[1], d.P);
                }
                if (lane == 0) {
                    rg_event e;
                    e.u = user; e.t = t; e.code = RG_EV_BANDIT | RG_EV_PHANTOM | a;
                    e.ps = static_cast<float>(ps);
                    d.phantom[slot] = e;
                    d.phantom_ps[slot] = ps;
                    d.has_phantom[slot] = 1;
                }
                c_ph += 1;
                st = kEmpty;
                break;
            }
            const bool org = st == RG_STATE_ORGANIC;
            // ---- bandit run: act and click of every lane's event (they decide where the run ends) ----
            double ps = 1.0, ctr = 0.0;
            uint32_t a = 0;
            bool click = false;
            if (!org) {
                if (HIST) {
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, te, 0, RG_DRAW_POLICY);
                    a = solo_ouc_act(d, hs, rg_uniform(pw.w[2], pw.w[3]), &ps);
                } else if (d.policy == RG_POLICY_LAST_VIEW_TABLE) {
                    ps = d.pol_ps ? static_cast<double>(d.pol_ps[lastv]) : 1.0;
                    a = static_cast<uint32_t>(d.pol_table[lastv]);
                } else {
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, te, 0, RG_DRAW_POLICY);
                    ps = 1.0 / static_cast<double>(d.P);
                    a = rg_bounded(pw.w[0], pw.w[1], d.P);
                }
                int dec = -1;
                if (!d.aux_pclick) {
                    if (rg_uniform(w.w[0], w.w[1]) < kNoClickBelow) dec = 0;
                    else dec = click_decide32<KC>(d.beta32 + static_cast<size_t>(a) * d.KB4, [&](int k) { return om[k]; }, d.K, d.KB4,
                                                  static_cast<float>(d.mu_b[a]), rg_uniform(w.w[0], w.w[1]));
                }
                if (dec >= 0) click = dec != 0;
                else {
                    const double* b = d.beta + static_cast<size_t>(a) * d.K;
                    const double* omd = d.omega + static_cast<size_t>(slot) * d.OMS;
                    double x = 0.0;
                    for (uint32_t k = 0; k < d.K; ++k) x += b[k] * omd[k];
                    ctr = ff64(x + d.mu_b[a]);
                    const double p0 = 1.0 - ctr;
                    click = (p0 / (p0 + ctr)) <= rg_uniform(w.w[0], w.w[1]);
                }
            }
            // ---- the state after every lane's event, had the run reached it (abstract.py:123-197) ----
            const double c0 = org ? d.cdf_o0 : d.cdf_b0, c1 = org ? d.cdf_o1 : d.cdf_b1;
            int ns = (c0 <= u_trans) + (c1 <= u_trans);
            if (click) ns = RG_STATE_ORGANIC;
            bool limit = false;
            if (organic_only && ns != RG_STATE_ORGANIC) ns = RG_STATE_STOP;
            else if (ns == RG_STATE_STOP) ns = kPhantom;
            else if (te + 2 >= kMaxSteps) { ns = RG_STATE_STOP; limit = true; }
            const unsigned long long leave = __ballot(ns != st);
            const int last = leave ? __builtin_ctzll(leave) : 63;            // the run's events of this pass: lanes 0 .. last
            const bool mine = lane <= last;
            const int ns_last = __shfl(ns, last);
            // ---- organic run: the product of every event of the run ----
            uint32_t v = 0;
            if (org) {
                const float4 h0 = hp[0];
                const double S = static_cast<double>(h0.x), delta = static_cast<double>(h0.y);
                const float Q = h0.z;
                const uint32_t n_hot = __builtin_bit_cast(uint32_t, h0.w);
                const double u_org = d.u_override ? d.u_override[slot] : rg_uniform(w.w[0], w.w[1]);   // (test hook)
                float uf = static_cast<float>(u_org), u_dn = uf, u_up = uf;
                if (static_cast<double>(uf) > u_org) u_dn = f32_down(uf);
                if (static_cast<double>(uf) < u_org) u_up = f32_up(uf);
                bool hit = false;
                {
                    float e[28];
#pragma unroll
                    for (int i = 1; i < 8; ++i) {
                        const float4 x = hp[i];
                        e[4 * i - 4] = x.x; e[4 * i - 3] = x.y; e[4 * i - 2] = x.z; e[4 * i - 1] = x.w;
                    }
#pragma unroll
                    for (int j = 0; j < kHotEntries; ++j) {
                        const bool in = static_cast<uint32_t>(j) < n_hot && e[3 * j + 1] < u_dn && u_up < e[3 * j + 2];
                        hit = hit || in;
                        v = in ? __builtin_bit_cast(uint32_t, e[3 * j]) : v;
                    }
                }
                const bool first_pend = pend && lane == 0;                   // the parked draw: float64, whatever the memo says
                hit = hit && !first_pend;
                c_hit += static_cast<uint32_t>(__popcll(__ballot(mine && hit)));
                const bool search = mine && !hit && !first_pend;
                bool ok = false;
                if (__ballot(search)) {
                    const double tau = u_org * S;
                    const float tauf = static_cast<float>(tau);
                    uint32_t sc_star = 0;
                    float pbf = 0.0f;
                    {
                        const float4* sp = reinterpret_cast<const float4*>(d.walk_scp + static_cast<size_t>(slot) * kMaxSC);
#pragma unroll
                        for (int i = 0; i < kMaxSC / 4; ++i) {
                            const float4 x = sp[i];
                            const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (xs[q] <= tauf) { sc_star += 1; pbf = fmaxf(pbf, xs[q]); }
                        }
                    }
                    bool found = sc_star < d.n_sc;
                    sc_star = min(sc_star, d.n_sc - 1);
                    uint32_t c_star;
                    {
                        const uint32_t cc0 = sc_star * d.sc_chunks, cc1 = min(cc0 + d.sc_chunks, d.n_chunks);
                        const float* cp = d.cache_chunk + static_cast<size_t>(slot) * d.n_chunks;
                        uint32_t cnt = 0;
                        for (uint32_t cb = cc0; cb < cc1; cb += 16) {
                            float4 w4[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                w4[i] = cb + 4 * i < cc1 ? *reinterpret_cast<const float4*>(cp + cb + 4 * i) : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float xs[4] = {w4[i].x, w4[i].y, w4[i].z, w4[i].w};
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    if (xs[q] <= tauf) { cnt += 1; pbf = fmaxf(pbf, xs[q]); }
                            }
                        }
                        found = found && cnt < cc1 - cc0;
                        c_star = min(cc0 + cnt, cc1 - 1);
                    }
                    const double pb = static_cast<double>(pbf);
                    const float rem = static_cast<float>(tau - pb);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    {
                        const int grp = lane >> 3, gl = lane & 7;
                        unsigned long long todo = __ballot(search);
                        while (todo) {
                            int src = -1;
#pragma unroll
                            for (int g = 0; g < 8; ++g) {
                                const int bit = todo ? __builtin_ctzll(todo) : -1;
                                if (g == grp) src = bit;
                                if (todo) todo &= todo - 1;
                            }
                            const bool has = src >= 0;
                            const int s2 = has ? src : 0;
                            const uint32_t cs = static_cast<uint32_t>(__shfl(static_cast<int>(c_star), s2));
                            const float rems = __shfl(rem, s2);
                            const float4* gp = reinterpret_cast<const float4*>(d.gamma32t + (static_cast<size_t>(cs) * K2) * 32) + gl;
                            float4 l = *(reinterpret_cast<const float4*>(d.mu32 + cs * 32) + gl);
#pragma unroll
                            for (int kh = 0; kh < K2; kh += KH) {
                                float4 gk[KH];
#pragma unroll
                                for (int k = 0; k < KH; ++k) gk[k] = gp[(kh + k) * 8];
#pragma unroll
                                for (int k = 0; k < KH; ++k) {
                                    const float wk = om[kh + k];                 // (every lane holds THE user's omega32)
                                    l.x = fmaf(gk[k].x, wk, l.x); l.y = fmaf(gk[k].y, wk, l.y);
                                    l.z = fmaf(gk[k].z, wk, l.z); l.w = fmaf(gk[k].w, wk, l.w);
                                }
                                asm volatile("" : "+v"(l.x), "+v"(l.y), "+v"(l.z), "+v"(l.w));
                            }
                            const float e0 = __builtin_amdgcn_exp2f(fmaf(l.x, kLog2e, -Q)), e1 = __builtin_amdgcn_exp2f(fmaf(l.y, kLog2e, -Q));
                            const float e2 = __builtin_amdgcn_exp2f(fmaf(l.z, kLog2e, -Q)), e3 = __builtin_amdgcn_exp2f(fmaf(l.w, kLog2e, -Q));
                            const float q0 = e0, q1 = q0 + e1, q2 = q1 + e2, q3 = q2 + e3;
                            float inc = q3;
#pragma unroll
                            for (int o2 = 1; o2 < 8; o2 <<= 1) {
                                const float y = __shfl_up(inc, o2, 8);
                                if (gl >= o2) inc += y;
                            }
                            float ex = __shfl_up(inc, 1, 8);
                            if (gl == 0) ex = 0.0f;
                            const float x0 = ex + q0, x1 = ex + q1, x2 = ex + q2, x3 = ex + q3;
                            const int j0 = x0 > rems ? 0 : x1 > rems ? 1 : x2 > rems ? 2 : x3 > rems ? 3 : -1;
                            const unsigned long long hits = __ballot(has && j0 >= 0);
                            const uint32_t gmask = static_cast<uint32_t>(hits >> (8 * grp)) & 0xFFu;
                            if (has) {
                                if (gmask) {
                                    if (gl == __builtin_ctz(gmask)) {
                                        mboxf[src * 3] = static_cast<float>(4 * gl + j0);
                                        mboxf[src * 3 + 1] = j0 == 0 ? ex : j0 == 1 ? x0 : j0 == 2 ? x1 : x2;
                                        mboxf[src * 3 + 2] = j0 == 0 ? x0 : j0 == 1 ? x1 : j0 == 2 ? x2 : x3;
                                    }
                                } else if (gl == 0) { mboxf[src * 3] = -1.0f; mboxf[src * 3 + 1] = 0.0f; mboxf[src * 3 + 2] = 0.0f; }
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    if (search) {
                        const int ix = static_cast<int>(mboxf[lane * 3]);
                        const CertLin ct = cert_correlated(S, pb, static_cast<double>(mboxf[lane * 3 + 1]), static_cast<double>(mboxf[lane * 3 + 2]), delta);
                        v = c_star * 32 + static_cast<uint32_t>(max(ix, 0));
                        const bool lo_ok = v == 0 || u_org * ct.den_lo > ct.num_lo;
                        const bool hi_ok = v == d.P - 1 || u_org * ct.den_hi < ct.num_hi;
                        ok = found && ix >= 0 && v < d.P && ct.valid && lo_ok && hi_ok;
                    }
                }
                // uncertified draws (and the parked one): float64 picks from the user's stored sums, one after the other
                unsigned long long picks = __ballot(mine && !hit && !ok);
                c_pick += static_cast<uint32_t>(__popcll(picks));
                while (picks) {
                    const int L = __builtin_ctzll(picks);
                    picks &= picks - 1;
                    const double s_u = __shfl(u_org, L);
                    const double M = static_cast<double>(d.exact_ref[slot]) * 0.69314718055994530942;
                    const uint32_t pv = exact_pick_pfx(d, d.exact_sums + static_cast<size_t>(slot) * n_cc,
                                                       d.omega + static_cast<size_t>(slot) * d.OMS, M, s_u, lane);
                    if (lane == L) v = pv;
                    __builtin_amdgcn_wave_barrier();
                }
                pend = false;
            }
            // ---- rows of the run's events ----
            const uint32_t n_ev = static_cast<uint32_t>(last) + 1u;
            if (row_next + n_ev > row_end) {
                for (uint64_t r = row_next + lane; r < row_end; r += 64)
                    if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&d.counters[kCntTailRows], static_cast<unsigned long long>(chunk_rows));
                base = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base >> 32))) << 32) |
                       __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base));
                row_next = base; row_end = base + chunk_rows;
            }
            const uint64_t my_row = row_next + static_cast<uint32_t>(lane);
            row_next += n_ev;
            if (mine && d.log && my_row < d.log_cap) {
                rg_event e;
                e.u = user; e.t = te;
                e.code = org ? v : (RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a);
                e.ps = org ? __builtin_nanf("") : static_cast<float>(ps);
                d.log[my_row] = e;
                if (!org && d.aux_ps) d.aux_ps[my_row] = ps;
                if (!org && d.aux_pclick) d.aux_pclick[my_row] = ctr;
            }
            if (org) {
                c_org += n_ev;
                lastv = static_cast<uint32_t>(__shfl(static_cast<int>(v), last));
                if (d.lpv && lane == 0) d.lpv[slot] = lastv;
                if (HIST)
                    for (int i = 0; i <= last; ++i) solo_hist_add(d, hs, hr, static_cast<uint32_t>(__shfl(static_cast<int>(v), i)), lane);
            } else {
                c_ban += n_ev;
                c_clicks += static_cast<uint32_t>(__popcll(__ballot(mine && click)));
            }
            // ---- the user after the run ----
            t += n_ev;
            if (ns_last == RG_STATE_STOP || ns_last == kPhantom) {
                if (lane == 0) d.n_events[slot] = t;
                c_maxt = max(c_maxt, t);
                c_limit += static_cast<uint32_t>(__shfl(static_cast<int>(limit), last));
            }
            st = ns_last == RG_STATE_STOP ? kEmpty : ns_last;
        }
    }
    for (uint64_t r = row_next + lane; r < row_end; r += 64)
        if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
    if (lane == 0) {
        if (c_org) atomicAdd(&d.counters[kCntTailOrganic], static_cast<unsigned long long>(c_org));
        if (c_ban) atomicAdd(&d.counters[kCntTailBandit], static_cast<unsigned long long>(c_ban));
        if (c_clicks) atomicAdd(&d.counters[RG_CNT_CLICKS], static_cast<unsigned long long>(c_clicks));
        if (c_ph) atomicAdd(&d.counters[RG_CNT_PHANTOM], static_cast<unsigned long long>(c_ph));
        if (c_pick) atomicAdd(&d.counters[RG_CNT_EXACT_DRAWS], static_cast<unsigned long long>(c_pick));
        if (c_hit) atomicAdd(&d.counters[kCntWalkHits], static_cast<unsigned long long>(c_hit));
        if (c_maxt) atomicMax(&d.counters[kCntTailMaxT], static_cast<unsigned long long>(c_maxt));
        if (c_limit) atomicAdd(&d.counters[kCntTailLimit], static_cast<unsigned long long>(c_limit));
    }
}
solo_kernel_t solo_kernel_for(const DevSim& d) {
    const bool ouc = d.policy == RG_POLICY_ORGANIC_USER_COUNT;
    if (walk2_kernel_for(d, 3) == nullptr || (ouc && d.hist_cap > kSoloHist)) return nullptr;
#ifdef RG_W2_ONLY
    return k_walk_solo<10, 1>;
#else
    switch (d.KH) {
        case 4: return ouc ? k_walk_solo<4, 1> : k_walk_solo<4, 0>;
        case 10: return ouc ? k_walk_solo<10, 1> : k_walk_solo<10, 0>;
        default: return ouc ? k_walk_solo<16, 1> : k_walk_solo<16, 0>;
    }
#endif
}
walk_kernel_t walk2_kernel_for(const DevSim& d, int occ) {
    // the forms k_walk2 is instantiated for: K <= 32, no group sums, the policies without a view history or the
    // OrganicUserEventCounter default (exploit_explore, epsilon = 0, select_randomly)
    const bool ouc = d.policy == RG_POLICY_ORGANIC_USER_COUNT;
    if (d.KH > 16 || !d.walk_hot) return nullptr;
    if (ouc && !(d.ouc_exploit_explore && d.ouc_epsilon == 0.0 && d.ouc_select_randomly)) return nullptr;
    if (d.policy != RG_POLICY_UNIFORM_ENV && d.policy != RG_POLICY_RANDOM_AGENT && d.policy != RG_POLICY_LAST_VIEW_TABLE && !ouc) return nullptr;
#ifdef RG_W2_ONLY     // kernel work: one instantiation, seconds to compile (never a shipped build)
    return k_walk2<10, 2>;
#else
    (void)occ;
    // the view-history line in LDS: compact (31 products per line) where a product fits 16 bits (RECOGYM_WALK_HIST=1: the
    // 64-bit line of 15 products, A/B)
    const bool compact = ouc && d.P <= 65535u && d.hist_cap >= 32u && d.hist_cap <= 32768u && !d.walk_line64;
    switch (d.KH) {
        case 4: return ouc ? (compact ? k_walk2<4, 2> : k_walk2<4, 1>) : k_walk2<4, 0>;
        case 10: return ouc ? (compact ? k_walk2<10, 2> : k_walk2<10, 1>) : k_walk2<10, 0>;
        default: return ouc ? (compact ? k_walk2<16, 2> : k_walk2<16, 1>) : k_walk2<16, 0>;
    }
#endif
}
void (*cache_prefix_kernel())(DevSim, int) { return k_cache_prefix; }
void (*exact_prefix_kernel())(DevSim, uint32_t) { return k_exact_prefix; }
#endif

// closes the books of a walked run: no lock-step step holds events; step 1 exists, is empty and starts after the raw rows
#if RG_HAS(1)
__global__ void k_walk_finish(DevSim d) {
    d.step_cnt[0] = 0; d.step_cnt[1] = 0; d.step_cnt[2] = 0; d.step_cnt[3] = 0;
    d.log_base[0] = 0;
    d.log_base[1] = d.counters[kCntTailRows];
}
#endif

#if RG_HAS(7)
// blocks per CU the kernel is compiled for (register budget 512 / OCC per lane): KH <= 16 at 2, 3 or 4, KH = 32 at 1
walk_kernel_t walk_kernel_for(const DevSim& d, int occ) {
#ifdef RG_W2_ONLY
    return nullptr;
#else
    // the O(P) forms of the OrganicUserEventCounter policy are compiled in only where the configuration can reach them
    const bool dense = d.policy == RG_POLICY_ORGANIC_USER_COUNT && !(d.ouc_exploit_explore && d.ouc_epsilon == 0.0);
#define RG_W(kh, o) (dense ? k_walk<kh, o, true> : k_walk<kh, o, false>)
    switch (d.KH) {
        // (three blocks per CU at K <= 32 — two and four were measured in round 2: 358 / 328 against 301 ms — one at K <= 64)
        case 4: return RG_W(4, 3);
        case 10: return RG_W(10, 3);
        case 16: return RG_W(16, 3);
        default: return RG_W(32, 1);
    }
#undef RG_W
#endif
}
#endif

// totals that are sums over the per-step counts
#if RG_HAS(1)
__global__ void k_totals(DevSim d, uint32_t t_now) {
    __shared__ unsigned long long so[kBlock], sb[kBlock];
    unsigned long long o = 0, b = 0;
    for (uint32_t t = threadIdx.x; t < t_now; t += kBlock) { o += d.step_cnt[2 * t]; b += d.step_cnt[2 * t + 1]; }
    so[threadIdx.x] = o; sb[threadIdx.x] = b;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { so[threadIdx.x] += so[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        d.counters[RG_CNT_ORGANIC] = so[0] + d.counters[kCntTailOrganic];
        d.counters[RG_CNT_BANDIT] = sb[0] + d.counters[kCntTailBandit];
        d.counters[RG_CNT_LIVE] = static_cast<unsigned long long>(d.step_cnt[2 * t_now]) + d.step_cnt[2 * t_now + 1];
        d.counters[RG_CNT_STEP] = max(static_cast<unsigned long long>(t_now), d.counters[kCntTailMaxT]);
        const unsigned long long rows = d.log_base[t_now];
        d.counters[RG_CNT_LOG_ROWS] = d.log ? (rows < d.log_cap ? rows : d.log_cap) : 0ull;
        d.counters[RG_CNT_LOG_DROPPED] = d.log ? (rows > d.log_cap ? rows - d.log_cap : 0ull) : 0ull;
    }
}
#endif

#if RG_HAS(1)
// rg_sim_step_user: what step t of a ONE-user simulator produced, packed for one read-back — the row it emitted (first row of
// the step; a stopping user's phantom row is not part of the step), the user's state and clock after it
__global__ void k_step_user_pack(DevSim d, uint32_t t) {
    rg_step_result* out = reinterpret_cast<rg_step_result*>(d.step1_buf + 8);
    const uint64_t row = d.log_base[t];
    rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f;
    const bool has = d.log && row < d.log_cap && d.log_base[t + 1] > row;
    if (has) e = d.log[row];
    out->row = e;
    out->state = d.step_cnt[2 * (t + 1)] ? RG_STATE_ORGANIC : (d.step_cnt[2 * (t + 1) + 1] ? RG_STATE_BANDIT : RG_STATE_STOP);
    out->has_row = has ? 1 : 0;
    out->time = d.time_mode ? d.utime[0] : static_cast<double>(t + 1);
    out->ps = (has && d.aux_ps) ? d.aux_ps[row] : static_cast<double>(e.ps);
    out->p_click = (has && d.aux_pclick) ? d.aux_pclick[row] : 0.0;
}

__global__ void __launch_bounds__(kBlock) k_export_state(DevSim d, uint32_t t, int8_t* state) {
    const uint32_t n_o = d.step_cnt[2 * t], n_b = d.step_cnt[2 * t + 1];
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_o + n_b; i += gridDim.x * kBlock) {
        if (i < n_o) state[d.uid[list_ptr(d, t & 1, 0)[i]]] = RG_STATE_ORGANIC;
        else state[d.uid[list_ptr(d, t & 1, 1)[i - n_o]]] = RG_STATE_BANDIT;
    }
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_export_omega(DevSim d, double* out) {
    const size_t n = static_cast<size_t>(d.n_users) * d.K;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t u = i / d.K, k = i % d.K;
        out[i] = d.omega[u * d.OMS + k];
    }
}
#endif

// test hooks
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_debug_set_omega(DevSim d, const double* in) {
    const size_t n = static_cast<size_t>(d.n_users) * d.K;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t u = i / d.K, k = i % d.K;
        d.omega[u * d.OMS + k] = in[i];
    }
}
#endif
#if RG_HAS(1)
// rg_sim_debug_click_decisions: click_decide32 (k_walk's fp32 decision) beside the float64 decision, per user index
__global__ void __launch_bounds__(kBlock) k_debug_click(DevSim d, const int32_t* actions, const double* u, uint8_t* out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        const uint32_t a = static_cast<uint32_t>(actions[i]);
        const double* om = d.omega + static_cast<size_t>(i) * d.OMS;
        const int dec = click_decide32<64>(d.beta32 + static_cast<size_t>(a) * d.KB4, [&](int k) { return static_cast<float>(om[k]); },
                                           d.K, d.KB4, static_cast<float>(d.mu_b[a]), u[i]);
        const double* b = d.beta + static_cast<size_t>(a) * d.K;
        double x = 0.0;
        for (uint32_t k = 0; k < d.K; ++k) x += b[k] * om[k];
        const double ctr = ff64(x + d.mu_b[a]);
        const double p0 = 1.0 - ctr;
        const bool click64 = (p0 / (p0 + ctr)) <= u[i];
        out[i] = static_cast<uint8_t>((dec >= 0 ? 1u : 0u) | (dec == 1 ? 2u : 0u) | (click64 ? 4u : 0u));
    }
}
// rg_sim_debug_set_history: view histories of the reset range from (distinct count, products ascending, counts)
__global__ void __launch_bounds__(kBlock) k_debug_set_history(DevSim d, const uint32_t* nd, const uint32_t* prod, const uint32_t* cnt,
                                                               uint32_t stride) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        hent_t* hr = hist_row(d, i);
        unsigned long long views = 0;
        for (uint32_t j = 0; j < nd[i]; ++j) {
            const uint32_t c = cnt[static_cast<size_t>(i) * stride + j];
            hr[1 + j] = (static_cast<hent_t>(prod[static_cast<size_t>(i) * stride + j]) << 32) | c;
            views += c;
        }
        hr[0] = (views << 32) | nd[i];
    }
}
// rg_sim_debug_ouc_acts: policy_act (OrganicUserEventCounter) with a caller-chosen second uniform, per user index
__global__ void __launch_bounds__(kBlock) k_debug_ouc_acts(DevSim d, const double* u1, int32_t* action, double* ps, uint8_t* flags) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        double p = 0.0;
        int fl = 0;
        const uint32_t a = policy_act<true, true>(d, i, static_cast<uint32_t>(d.first_user + i), 0u, &p, u1[i], &fl);
        action[i] = static_cast<int32_t>(a); ps[i] = p; flags[i] = static_cast<uint8_t>(fl);
    }
}
__global__ void __launch_bounds__(kBlock) k_debug_fate_round2(DevSim d, uint8_t* flags) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) flags[i] = d.f64_valid[i] ? 1 : 0;
}
__global__ void __launch_bounds__(kBlock) k_debug_fate_last(DevSim d, uint8_t* flags, uint32_t base, const unsigned long long* count) {
    const uint32_t n = static_cast<uint32_t>(*count);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t slot = d.park_list[base + i];
        if (slot != 0xFFFFFFFFu) flags[slot] |= 2;
    }
}
__global__ void __launch_bounds__(kBlock) k_debug_uncertified(DevSim d, uint32_t t_prev, uint8_t* flags) {
    const uint32_t n_a = d.exact_cnt[t_prev], n = n_a + (d.use_cache ? d.exact_cnt_b[t_prev] : 0u);
    const uint32_t* lst = list_ptr(d, t_prev & 1, RG_STATE_ORGANIC);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        flags[d.uid[lst[d.exact_list[i < n_a ? i : d.n_cap - 1u - (i - n_a)]]]] = 1;
}
#endif

// live users only (after a repack the slots of users that left are gone); `out` is zero-filled first
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_export_omega_live(DevSim d, uint32_t t, double* out) {
    const uint32_t n_o = d.step_cnt[2 * t], n_b = d.step_cnt[2 * t + 1];
    const size_t n = static_cast<size_t>(n_o + n_b) * d.K;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const uint32_t li = static_cast<uint32_t>(i / d.K), k = static_cast<uint32_t>(i % d.K);
        const uint32_t slot = li < n_o ? list_ptr(d, t & 1, 0)[li] : list_ptr(d, t & 1, 1)[li - n_o];
        out[static_cast<size_t>(d.uid[slot]) * d.K + k] = d.omega[static_cast<size_t>(slot) * d.OMS + k];
    }
}
#endif

// ------------------------------------------------------------------------------------------
// repack: the live lists lose their order step by step (the block that reserves first writes
// first) and thin out as users leave, so the per-user gathers of omega / the view history turn
// into scattered single-line fetches (measured: k_advance 0.22 -> 0.57 ns/event between steps
// 0-20 and 220-240 of the 10 M-user run).  Every few steps the state of the users still alive is
// therefore copied into the second buffer in list order — new slot = position in [organic |
// bandit] — and the lists become the identity.  Pure relabelling: user ids travel in uid[].
// ------------------------------------------------------------------------------------------
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_repack_copy(DevSim d, uint32_t t) {
    const uint32_t n_o = d.step_cnt[2 * t], n_b = d.step_cnt[2 * t + 1], n = n_o + n_b;
    const uint32_t* cur_o = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const uint32_t* cur_b = list_ptr(d, t & 1, RG_STATE_BANDIT);
    const uint32_t sub = threadIdx.x & 31;                       // 32 lanes move one user
    const uint32_t groups = gridDim.x * (kBlock / 32);
    for (uint32_t i = blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5); i < n; i += groups) {
        const uint32_t old = i < n_o ? cur_o[i] : cur_b[i - n_o];
        for (uint32_t k = sub; k < d.OMS; k += 32)
            d.omega_alt[static_cast<size_t>(i) * d.OMS + k] = d.omega[static_cast<size_t>(old) * d.OMS + k];
        if (sub == 0) {
            d.uid_alt[i] = d.uid[old];
            if (d.lpv) d.lpv_alt[i] = d.lpv[old];
        }
        if (d.hist_cap) {
            const hent_t* src = d.hist + static_cast<size_t>(old) * d.hist_cap;
            hent_t* dst = d.hist_alt + static_cast<size_t>(i) * d.hist_cap;
            const uint32_t hn = h_cnt(src[0]) + 1u;               // header + products
            for (uint32_t e = sub; e < hn; e += 32) dst[e] = src[e];
        }
    }
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_repack_lists(DevSim d, uint32_t t) {
    const uint32_t n_o = d.step_cnt[2 * t], n_b = d.step_cnt[2 * t + 1];
    uint32_t* cur_o = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    uint32_t* cur_b = list_ptr(d, t & 1, RG_STATE_BANDIT);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_o + n_b; i += gridDim.x * kBlock) {
        if (i < n_o) cur_o[i] = i;
        else cur_b[i - n_o] = i;
    }
}
#endif

// ------------------------------------------------------------------------------------------
// log reordering: rows of user u occupy [off[u], off[u] + n_events[u] + has_phantom[u])
// ------------------------------------------------------------------------------------------
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_rows_per_user(DevSim d, int64_t* rows) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock)
        rows[i] = static_cast<int64_t>(d.n_events[i]) + d.has_phantom[i];
}
#endif

// exclusive scan, three phases (block sums -> scan of sums by one block -> add)
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_scan_block(const int64_t* in, int64_t* out, int64_t* block_sums, uint32_t n) {
    __shared__ int64_t s[kBlock];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const int64_t x = i < n ? in[i] : 0;
    s[threadIdx.x] = x;
    __syncthreads();
    for (int o = 1; o < kBlock; o <<= 1) {
        const int64_t y = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += y;
        __syncthreads();
    }
    if (i < n) out[i] = s[threadIdx.x] - x;
    if (threadIdx.x == kBlock - 1) block_sums[blockIdx.x] = s[threadIdx.x];
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_scan_sums(int64_t* block_sums, uint32_t nb, int64_t* total) {
    __shared__ int64_t s[kBlock];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        const int64_t x = i < nb ? block_sums[i] : 0;
        s[threadIdx.x] = x;
        __syncthreads();
        for (int o = 1; o < kBlock; o <<= 1) {
            const int64_t y = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += y;
            __syncthreads();
        }
        if (i < nb) block_sums[i] = carry + s[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 0) carry += s[kBlock - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_scan_add(int64_t* out, const int64_t* block_sums, uint32_t n) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] += block_sums[blockIdx.x];
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_scatter_rows(DevSim d, uint64_t n_rows, const int64_t* off,
                                                       rg_event* out, uint64_t out_cap) {
    for (uint64_t r = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; r < n_rows;
         r += static_cast<uint64_t>(gridDim.x) * kBlock) {
        const rg_event e = d.log[r];
        if (e.code == kHoleCode) continue;                           // unused entry of a k_walk row chunk
        const uint64_t dst = static_cast<uint64_t>(off[e.u - d.first_user]) + e.t;
        if (dst < out_cap) out[dst] = e;
    }
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_scatter_phantom(DevSim d, const int64_t* off, rg_event* out,
                                                          uint64_t out_cap) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        if (!d.has_phantom[i]) continue;
        const uint64_t dst = static_cast<uint64_t>(off[i]) + d.n_events[i];
        if (dst < out_cap) out[dst] = d.phantom[i];
    }
}
#endif

// the float64 side arrays in the same order: NaN where the reference's column is NaN (organic rows; p_click of
// the phantom row, which is never drawn)
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_scatter_aux(DevSim d, uint64_t n_rows, const int64_t* off,
                                                      double* out_ps, double* out_pc, uint64_t out_cap) {
    const double nan = __builtin_nan("");
    for (uint64_t r = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; r < n_rows;
         r += static_cast<uint64_t>(gridDim.x) * kBlock) {
        const rg_event e = d.log[r];
        if (e.code == kHoleCode) continue;
        const uint64_t dst = static_cast<uint64_t>(off[e.u - d.first_user]) + e.t;
        if (dst >= out_cap) continue;
        const bool is_b = (e.code & RG_EV_BANDIT) != 0;
        if (out_ps) out_ps[dst] = (is_b && d.aux_ps) ? d.aux_ps[r] : nan;
        if (out_pc) out_pc[dst] = (is_b && d.aux_pclick) ? d.aux_pclick[r] : nan;
    }
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_scatter_aux_phantom(DevSim d, const int64_t* off, double* out_ps,
                                                              double* out_pc, uint64_t out_cap) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        if (!d.has_phantom[i]) continue;
        const uint64_t dst = static_cast<uint64_t>(off[i]) + d.n_events[i];
        if (dst >= out_cap) continue;
        if (out_ps) out_ps[dst] = d.phantom_ps[i];
        if (out_pc) out_pc[dst] = __builtin_nan("");
    }
}
#endif

#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_scatter_time(DevSim d, uint64_t n_rows, const int64_t* off, double* out, uint64_t out_cap) {
    for (uint64_t r = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; r < n_rows;
         r += static_cast<uint64_t>(gridDim.x) * kBlock) {
        const rg_event e = d.log[r];
        if (e.code == kHoleCode) continue;
        const uint64_t dst = static_cast<uint64_t>(off[e.u - d.first_user]) + e.t;
        if (dst < out_cap) out[dst] = d.aux_time ? d.aux_time[r] : static_cast<double>(e.t);
    }
}
#endif
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_scatter_time_phantom(DevSim d, const int64_t* off, double* out, uint64_t out_cap) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        if (!d.has_phantom[i]) continue;
        const uint64_t dst = static_cast<uint64_t>(off[i]) + d.n_events[i];
        if (dst < out_cap) out[dst] = d.time_mode ? d.phantom_time[i] : static_cast<double>(d.n_events[i]);
    }
}
#endif
#if RG_HAS(1)
__global__ void __launch_bounds__(kBlock) k_export_time(DevSim d, uint32_t t, double* out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock)
        out[i] = d.time_mode ? d.utime[i] : static_cast<double>(d.n_events[i] ? d.n_events[i] : t);
}
#endif

#if RG_HAS(1)   // host code (to the end of the namespace)
inline int grid_for(uint64_t n, int per_block = kBlock) {
    uint64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return static_cast<int>(g);
}

int prof_mark(rg_sim* sim, hipStream_t st) {
    if (!sim->profiling) return RG_OK;
    if (sim->prof_used == sim->prof_events.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        sim->prof_events.push_back(e);
    }
    HIP_TRY(hipEventRecord(sim->prof_events[sim->prof_used++], st));
    return RG_OK;
}

// float64 draw of this step: from_list = 1 resolves the users the MFMA kernel could not certify
// (est = expected count), from_list = 0 serves every organic user (pure float64 mode)
void launch_exact(rg_sim* sim, uint32_t t, int from_list, uint64_t est, hipStream_t st) {
    const uint32_t n_chunks = sim->d.PT / 64;
    // Without the per-user cache the float64 chunk sums of a step's uncertified draws go through a scratch of
    // exact_rows rows: one batch where the step cannot have more draws than that, else two (covers 25 % of the live
    // users uncertified; beyond that the run reports RG_CNT_EXACT_OVERFLOW instead of dropping draws)
    const bool batched = from_list == 1 && !sim->d.use_cache;
    const int n_batches = (batched && sim->live_upper > sim->d.exact_rows) ? 2 : 1;
    for (int b = 0; b < n_batches; ++b) {
        DevSim d = sim->d;
        d.exact_base = batched ? static_cast<uint32_t>(b) * d.exact_rows : 0u;
        d.exact_last = b + 1 == n_batches ? 1u : 0u;
        if (batched && est > d.exact_rows) est = d.exact_rows;
        if (exact_m_kernel_t km = (sim->opt.exact_tile ? nullptr : exact_m_kernel_for(d.XKB))) {
            if (!from_list) {
                launch_exact_m(km, d, t, 0, 0, est, st);
                hipLaunchKernelGGL(exact_ref_kernel(), dim3(grid_for(est, kBlock / 64)), dim3(kBlock), 0, st, d, t, 1u);
            }
            launch_exact_m(km, d, t, from_list, 1, est, st);
            hipLaunchKernelGGL(exact_pick_kernel(), dim3(grid_for(est, kBlock / 64)), dim3(kBlock),
                               sizeof(double) * d.K * (kBlock / 64), st, d, t, from_list, 1u);
            continue;
        }
        const uint64_t groups = (est + kExactUsers - 1) / kExactUsers;
        uint32_t S = static_cast<uint32_t>(2048 / (groups ? groups : 1));
        if (S > (n_chunks + 7) / 8) S = (n_chunks + 7) / 8;
        if (S < 1) S = 1;
        const int grid = grid_for(groups * S, 1);
        const size_t smem = sizeof(double) * (static_cast<size_t>(d.K) * 64 + 64 + kExactUsers * d.K);
        if (!from_list) {
            hipLaunchKernelGGL(exact_tile_kernel(), dim3(grid), dim3(kBlock), smem, st, d, t, 0, 0, S);
            hipLaunchKernelGGL(exact_ref_kernel(), dim3(grid_for(est, kBlock / 64)), dim3(kBlock), 0, st, d, t, 8u);
        }
        hipLaunchKernelGGL(exact_tile_kernel(), dim3(grid), dim3(kBlock), smem, st, d, t, from_list, 1, S);
        hipLaunchKernelGGL(exact_pick_kernel(), dim3(grid_for(est, kBlock / 64)), dim3(kBlock),
                           sizeof(double) * d.K * (kBlock / 64), st, d, t, from_list, 8u);
    }
}

int device_cus(rg_sim* sim) {
    if (!sim->n_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            sim->n_cus = prop.multiProcessorCount;
        else sim->n_cus = 256;
    }
    return sim->n_cus;
}

// Grid of a sweep kernel.  The per-wave scratch (chunk sums + super-chunk records, ~40 KB per wave at C3) is indexed
// by BLOCK in the fused form.  Capping the grid at the blocks the device holds at once (RECOGYM_RESIDENT_GRID=1) keeps
// that scratch an ~80 MB working set instead of ~650 MB, but the memory-side counters (FETCH_SIZE / WRITE_SIZE sit at
// the L2 <-> fabric boundary and include Infinity-Cache hits) were identical and the kernel 3 % slower: not the default.
int sweep_grid(rg_sim* sim, uint64_t work_items, uint32_t S) {
    int grid = grid_for(work_items, 1);
    const int resident = device_cus(sim) * (sim->draw_users == 256 ? 1 : 2);
    if (S == 1 && grid > resident && sim->opt.resident_grid) grid = resident;
    if (sim->draw_users == 256 && grid > kMaxGrid / 2) grid = kMaxGrid / 2;     // 8 groups per block share the per-wave scratch
    return grid;
}

int launch_step(rg_sim* sim, const int32_t* d_actions, hipStream_t st) {
    if (sim->t >= kMaxSteps) return fail(RG_ELIMIT, "more than %u steps", kMaxSteps);
    const DevSim& d = sim->d;
    const uint32_t t = sim->t;
    const uint32_t upper = sim->live_upper;
    if (sim->repack_every && t && t % sim->repack_every == 0 && sim->d.n_cap >= sim->opt.repack_min && upper >= sim->opt.repack_min / 4) {
        DevSim& m = sim->d;
        hipLaunchKernelGGL(k_repack_copy, dim3(grid_for(upper, kBlock / 32)), dim3(kBlock), 0, st, m, t);
        hipLaunchKernelGGL(k_repack_lists, dim3(grid_for(upper)), dim3(kBlock), 0, st, m, t);
        std::swap(m.omega, m.omega_alt); std::swap(m.hist, m.hist_alt); std::swap(m.uid, m.uid_alt);
        if (m.lpv) std::swap(m.lpv, m.lpv_alt);
        sim->repacked = true;
        if (sim->opt.debug) fprintf(stderr, "[recogym] repack at t=%u (upper %u)\n", t, upper);
    }
    if (int rc = prof_mark(sim, st)) return rc;
    // 1. organic product draws of this step (read omega before the transition drifts it)
    if (d.use_mfma == 2 && d.use_cache && t > 0) {
        // sigma_omega == 0, after step 0: every live user's exp-sums are in the per-user cache — search only
        if (int rc = prof_mark(sim, st)) return rc;
        hipLaunchKernelGGL(cached_kernel_for(d), dim3(grid_for(upper, kBlock)), dim3(kBlock),
                           sizeof(float) * (kBlock / 64) * 64 * 2 * d.KH, st, d, t);
        if (int rc = prof_mark(sim, st)) return rc;
        launch_exact(sim, t, 1, upper / 100 + 16, st);
    } else if (d.use_mfma == 2) {
        // few user tiles: slice the products so that the step's latency is a slice, not a sweep
        const uint32_t tiles_up = (upper + sim->draw_users - 1) / sim->draw_users;
        uint32_t S = tiles_up >= 131072u / sim->draw_users ? 1u : (262144u / sim->draw_users) / (tiles_up ? tiles_up : 1u);
        if (sim->opt.slices >= 0) S = static_cast<uint32_t>(sim->opt.slices);   // tests: force either form
        if (S > d.n_sc) S = d.n_sc;
        if (S < 1) S = 1;
        const int grid = sweep_grid(sim, static_cast<uint64_t>(tiles_up) * S, S);
        // (the search stays at the end of every user tile of the sweep: as its own kernel over the whole step — scratch slot
        // per user tile — the sweep got 15 % shorter and the step 6 % longer: profiles/r3/ab_call26_*, ab_call27_*)
        hipLaunchKernelGGL(sim->bf16_kernel, dim3(grid), dim3(sim->draw_threads), sim->bf16_smem, st, d, t, S);
        if (int rc = prof_mark(sim, st)) return rc;
        if (S > 1)
            hipLaunchKernelGGL(search_kernel_for(d), dim3(grid_for(upper, 128)), dim3(kBlock),
                               sizeof(float) * 4 * 32 * 2 * d.KH, st, d, t);
        if (d.use_cache)       // step 0 of a sigma_omega == 0 run: the rows every later draw starts from
            hipLaunchKernelGGL(finalize_kernel_for(d), dim3(grid_for(d.n_users)), dim3(kBlock), 0, st, d);
        if (int rc = prof_mark(sim, st)) return rc;
        launch_exact(sim, t, 1, upper / 100 + 16, st);
    } else if (d.use_mfma) {
        const int grid = grid_for(upper, 128);
        const size_t smem = sim->mfma_smem;
        hipLaunchKernelGGL(mfma_kernel_for(d.KH), dim3(grid), dim3(kBlock), smem, st, d, t);
        if (int rc = prof_mark(sim, st)) return rc;
        if (int rc = prof_mark(sim, st)) return rc;
        // draws the fp32 path could not certify -> float64 (a few percent of the organic users)
        launch_exact(sim, t, 1, upper / 100 + 16, st);
    } else {
        if (int rc = prof_mark(sim, st)) return rc;
        if (int rc = prof_mark(sim, st)) return rc;
        launch_exact(sim, t, 0, upper, st);
    }
    if (int rc = prof_mark(sim, st)) return rc;
    if (d.policy == RG_POLICY_LOGREG_FROZEN) {
        // acts of the users whose view history changed since their last one (frozen LogReg at scale)
        hipLaunchKernelGGL(logreg_select_kernel(), dim3(grid_for(upper)), dim3(kBlock), 0, st, d, t);
        if (d.lr_coef16_t) {       // screen (a wave per act and class range), then decide (a wave per act)
            hipLaunchKernelGGL(logreg_screen_kernel(), dim3(grid_for((static_cast<uint64_t>(upper) / 4 + 64) * kLrSplit, kBlock / 64)),
                               dim3(kBlock), 0, st, d, t);
            hipLaunchKernelGGL(logreg_decide_kernel(), dim3(grid_for(static_cast<uint64_t>(upper) / 4 + 64, kBlock / 64)), dim3(kBlock), 0, st, d, t);
        } else
            hipLaunchKernelGGL(logreg_acts_kernel(), dim3(grid_for(static_cast<uint64_t>(upper) / 4 + 64, kBlock / 64)), dim3(kBlock), 0, st, d, t);
    }
    if (int rc = prof_mark(sim, st)) return rc;
    // 2. click draws, transitions, drift, next lists, bandit + phantom rows
    hipLaunchKernelGGL(advance_kernel(), dim3(grid_for(upper, kAdvBlock)), dim3(kAdvBlock), 0, st, d, t, d_actions);
    if (d.sigma_omega != 0.0)
        hipLaunchKernelGGL(drift_kernel(), dim3(grid_for(static_cast<uint64_t>(upper) * ((d.K + 1) / 2))), dim3(kBlock), 0, st, d, t);
    HIP_TRY(hipGetLastError());
    if (int rc = prof_mark(sim, st)) return rc;
    sim->t = t + 1;
    return RG_OK;
}

// fold the recorded events into per-kernel totals (synchronises on the last event)
int prof_collect(rg_sim* sim) {
    if (!sim->prof_used) return RG_OK;
    HIP_TRY(hipEventSynchronize(sim->prof_events[sim->prof_used - 1]));
    for (size_t i = 0; i + 5 < sim->prof_used; i += 6) {
        for (int k = 0; k < 5; ++k) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, sim->prof_events[i + k], sim->prof_events[i + k + 1]));
            sim->prof_ms[k] += ms;
        }
        sim->prof_launches += 1;
    }
    sim->prof_used = 0;
    return RG_OK;
}

// rg_sim_run "to the end" of a sigma_omega == 0 run: sweep (fills the per-user cache) -> k_walk round 1 ->
// float64 sums of the parked users in one batch -> k_walk round 2.  Five launches and one host read-back.
int run_walk(rg_sim* sim, hipStream_t st) {
    const DevSim& d = sim->d;
    (void)device_cus(sim);
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    auto mark = [&](int i) -> int {
        if (!sim->profiling) return RG_OK;
        HIP_TRY(hipEventCreate(&ev[i]));
        HIP_TRY(hipEventRecord(ev[i], st));
        return RG_OK;
    };
    if (int rc = mark(0)) return rc;
    sim->fate_count = nullptr;
    bool fused_prefix = false;
    // 1. every user's first product sweep: only the per-user sums are kept (no search, no rows)
    {
        DevSim ds = d;
        const uint32_t tiles_up = (d.n_users + sim->draw_users - 1) / sim->draw_users;
        uint32_t S = tiles_up >= 131072u / sim->draw_users ? 1u : (262144u / sim->draw_users) / (tiles_up ? tiles_up : 1u);
        if (sim->opt.slices >= 0) S = static_cast<uint32_t>(sim->opt.slices);
        if (S > d.n_sc) S = d.n_sc;
        if (S < 1) S = 1;
        // k_walk2 behind the fused (unsliced) form of the pipelined fp16 sweep of K <= 21: the sweep stores the sums in the
        // walk's prefix form itself (no conversion pass over the 1.3 KB of chunk sums per user)
        fused_prefix = sim->walk2 && S == 1 && sim->bf16_kernel == bf16p_kernel_for(d) && d.f16 && !d.wide && !sim->opt.sweep_prefix_off;
        ds.sweep_only = fused_prefix ? 2u : 1u;
        const int grid = sweep_grid(sim, static_cast<uint64_t>(tiles_up) * S, S);
        hipLaunchKernelGGL(sim->bf16_kernel, dim3(grid), dim3(sim->draw_threads), sim->bf16_smem, st, ds, 0u, S);
    }
    if (int rc = mark(1)) return rc;
    hipLaunchKernelGGL(finalize_kernel_for(d), dim3(grid_for(d.n_users)), dim3(kBlock), 0, st, d);
    if (sim->walk2)      // the sums in prefix form, the memo rows emptied
        hipLaunchKernelGGL(cache_prefix_kernel(), dim3(grid_for((static_cast<uint64_t>(d.n_users) + 7) / 8, kBlock / 64)), dim3(kBlock), 0, st, d,
                           fused_prefix ? 1 : 0);
    if (int rc = mark(2)) return rc;
    // 2. round 1: every user from t = 0 to its end or to its first uncertified draw
    const size_t smem = sim->walk2 ? (kBlock / 64) * walk2_wave_lds(d.policy == RG_POLICY_ORGANIC_USER_COUNT)
                                   : (kBlock / 64) * walk_wave_lds(d.KH);
    const walk_kernel_t wk = sim->walk2 ? walk2_kernel_for(d, sim->walk_occ) : walk_kernel_for(d, d.KH <= 16 ? sim->walk_occ : 1);
    auto launch_walk = [&](uint32_t n_work, int round, uint32_t in_base, uint32_t out_base) {
        const int occ = d.KH <= 16 ? sim->walk_occ : 1;
        const int blocks_cap = sim->n_cus * occ;
        int blocks = static_cast<int>((static_cast<uint64_t>(n_work) + kBlock - 1) / kBlock);
        if (blocks > blocks_cap) blocks = blocks_cap;
        if (blocks < 1) blocks = 1;
        // rows are reserved per wave in chunks: ~1/32 of what a wave will emit, within [256, 4096] (unused entries:
        // < 64 per chunk and the rest of every wave's last chunk — a few percent of the raw log)
        uint64_t chunk = static_cast<uint64_t>(n_work) * 100 / (static_cast<uint64_t>(blocks) * 4 * 32);
        chunk = chunk / 64 * 64;
        if (chunk < 256) chunk = 256;
        if (chunk > 4096) chunk = 4096;
        if (smem > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wk), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
        hipLaunchKernelGGL(wk, dim3(blocks), dim3(kBlock), smem, st, d, n_work, round, static_cast<uint32_t>(chunk), in_base, out_base);
    };
    launch_walk(d.n_users, 1, 0u, 0u);
    if (int rc = mark(3)) return rc;
    // 3. the users parked at an uncertified draw: float64 sums in one batch, then their round
    unsigned long long* h64 = reinterpret_cast<unsigned long long*>(sim->h_pinned);
    HIP_TRY(hipMemcpyAsync(h64, d.counters + kCntParkCnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t n_park = static_cast<uint32_t>(*h64);
    if (sim->opt.debug) {
        unsigned long long ev2[2] = {0, 0};
        HIP_TRY(hipMemcpy(ev2, d.counters + kCntTailOrganic, sizeof(ev2), hipMemcpyDeviceToHost));
        fprintf(stderr, "[recogym] walk round 1: %llu organic + %llu bandit events, %u users parked of %u\n", ev2[0], ev2[1], n_park, d.n_users);
    }
    // round 2 over the parked (and handed-over) users; what IT hands over is appended behind them for round 3
    auto later_rounds = [&](uint32_t n_list) -> int {
        const uint32_t base3 = (n_list + 63u) & ~63u;
        if (sim->walk2)      // the listed users' float64 sums as prefixes (anchored certificate, prefix pick)
            hipLaunchKernelGGL(exact_prefix_kernel(), dim3(grid_for(n_list, kBlock / 64)), dim3(kBlock), 0, st, d, n_list);
        HIP_TRY(hipMemsetAsync(d.counters + kCntWalkTicket, 0, sizeof(unsigned long long), st));
        HIP_TRY(hipMemsetAsync(d.counters + kCntParkCnt, 0, sizeof(unsigned long long), st));
        launch_walk(n_list, 2, 0u, base3);
        if (!d.walk_handover) return RG_OK;
        sim->fate_base = base3; sim->fate_count = d.counters + kCntParkCnt;
        HIP_TRY(hipMemcpyAsync(h64, d.counters + kCntParkCnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const uint32_t n_left = static_cast<uint32_t>(*h64);
        if (n_left) {
            HIP_TRY(hipMemsetAsync(d.counters + kCntWalkTicket, 0, sizeof(unsigned long long), st));
            const solo_kernel_t sk = (sim->walk2 && sim->walk_solo) ? solo_kernel_for(d) : nullptr;
            if (sk) {      // a wave per user, a lane per consecutive event
                // >= 4 listed users per wave; rows reserved per wave in chunks of ~1/8 of what it will emit (a commit is <= 64
                // rows; what a wave leaves of its last chunk are holes in the raw log: a few percent of this round's rows)
                uint32_t blocks = (n_left + 15u) / 16u;
                const uint32_t cap = static_cast<uint32_t>(sim->n_cus) * 8u;
                if (blocks > cap) blocks = cap;
                uint64_t chunk = static_cast<uint64_t>(n_left) * 150 / (static_cast<uint64_t>(blocks) * 4 * 8);
                chunk = chunk / 64 * 64;
                if (chunk < 64) chunk = 64;
                if (chunk > 1024) chunk = 1024;
                hipLaunchKernelGGL(sk, dim3(blocks), dim3(kBlock), 0, st, d, n_left, static_cast<uint32_t>(chunk), base3);
            } else launch_walk(n_left, 3, base3, base3);
        }
        return RG_OK;
    };
    if (n_park) {
        const uint32_t mfma_of_8 = static_cast<uint32_t>(sim->opt.exact_mix);     // groups of every 8 that take the matrix form (8 = all)
        exact_h_kernel_t kh = mfma_of_8 < 8 ? exact_h_kernel_for(d.XKB) : nullptr;
        if (kh) {
            HIP_TRY(hipMemsetAsync(d.counters + kCntWalkTicket, 0, sizeof(unsigned long long), st));
            const uint32_t groups = (n_park + 255u) / 256u;
            const uint32_t grid = groups < 1024u ? groups : 1024u;
            hipLaunchKernelGGL(kh, dim3(grid), dim3(kBlock), exact_m_lds(d.XKB), st, d, n_park, mfma_of_8);
            if (int rc = mark(4)) return rc;
            if (int rc = later_rounds(n_park)) return rc;
            goto walked;
        }
        if (exact_m_kernel_t km = exact_m_kernel_for(d.XKB)) {
            launch_exact_m(km, d, n_park, 2, 1, n_park, st);
            if (int rc = mark(4)) return rc;
            if (int rc = later_rounds(n_park)) return rc;
            goto walked;
        }
        return fail(RG_ESTATE, "no float64 batch kernel for K = %u", d.K);
    } else if (int rc = mark(4)) return rc;
walked:
    if (int rc = mark(5)) return rc;
    hipLaunchKernelGGL(k_walk_finish, dim3(1), dim3(1), 0, st, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h64, d.counters + kCntTailLimit, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (sim->profiling) {
        float ms[5];
        for (int i = 0; i < 5; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
        sim->prof_ms[0] += ms[0]; sim->prof_ms[1] += ms[1]; sim->prof_ms[2] += ms[3];
        sim->prof_walk_ms[0] += ms[2]; sim->prof_walk_ms[1] += ms[4];
        sim->prof_tail_ms += ms[2] + ms[4];
        sim->prof_launches += 1;
        for (int i = 0; i < 6; ++i) (void)hipEventDestroy(ev[i]);
    }
    sim->t = 1;
    sim->live_upper = 0;
    if (*h64) return fail(RG_ELIMIT, "more than %u steps", kMaxSteps);
    return RG_OK;
}

// The same run as a PIPELINE over user groups: the reset range is cut into G groups of equal size; per group
//   sweep -> finalize -> round 1          (k_draw_bf16p sweep_only = 2, k_cache_finalize + k_cache_prefix, k_walk2)
//   float64 batch -> prefixes -> round 2  (k_exact_sums_h, k_exact_prefix, k_walk2) on the users round 1 parked
// and one last round (k_walk_solo) over what the rounds 2 handed over.  The second chain of group g runs on a second stream
// while the first chain of group g + 1 runs on the caller's: the float64 batch is bound by the float64 pipes, the walk by its
// chains of dependent loads (half of its wave cycles are waits), so they share the compute units instead of taking turns.
// Every list length stays on the device (q_count): no host read-back between the launches, one at the end (the step limit).
// Results are those of run_walk bit for bit: every draw is addressed by (seed, user, t), a user's events are walked by one
// lane at a time, and the sorted log does not depend on the raw order.
int run_walk_pipe(rg_sim* sim, hipStream_t st) {
    const DevSim& d = sim->d;
    const int n_cus = device_cus(sim);
    const uint32_t n = d.n_users;
    // groups: equal sizes, multiples of 256 users, each large enough for the unsliced sweep (>= 1024 user tiles)
    uint32_t G = static_cast<uint32_t>(sim->pipe_groups);
    if (G > kMaxWalkGroups) G = kMaxWalkGroups;
    while (G > 1 && n / G < sim->pipe_min_users) --G;
    const uint32_t gsz = (((n + G - 1) / G) + 255u) & ~255u;
    G = (n + gsz - 1) / gsz;
    const int mode = G > 1 ? sim->pipe_mode : 0;
    if (mode >= 1 && !sim->pipe_streams[0]) {
        HIP_TRY(hipStreamCreateWithFlags(&sim->pipe_streams[0], hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&sim->pipe_streams[1], hipStreamNonBlocking));
    }
    const size_t n_ev = 3 * static_cast<size_t>(kMaxWalkGroups) + 2;
    while (sim->pipe_events.size() < n_ev) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        sim->pipe_events.push_back(e);
    }
    hipStream_t sA = st, sB = mode >= 1 ? sim->pipe_streams[0] : st, sS = mode >= 2 ? sim->pipe_streams[1] : st;
    // profiling: a pair of timing events around every launch group, on the stream it is launched on
    struct Span { int cls; hipEvent_t a, b; };
    std::vector<Span> spans;
    auto span_begin = [&](int cls, hipStream_t s) -> int {
        if (!sim->profiling) return RG_OK;
        Span sp{cls, nullptr, nullptr};
        HIP_TRY(hipEventCreate(&sp.a)); HIP_TRY(hipEventCreate(&sp.b));
        HIP_TRY(hipEventRecord(sp.a, s));
        spans.push_back(sp);
        return RG_OK;
    };
    auto span_end = [&](hipStream_t s) -> int {
        if (!sim->profiling) return RG_OK;
        HIP_TRY(hipEventRecord(spans.back().b, s));
        return RG_OK;
    };
    hipEvent_t wall[2] = {nullptr, nullptr};
    if (sim->profiling) {
        HIP_TRY(hipEventCreate(&wall[0])); HIP_TRY(hipEventCreate(&wall[1]));
        HIP_TRY(hipEventRecord(wall[0], st));
    }
    HIP_TRY(hipMemsetAsync(d.walk_ctl, 0, sizeof(unsigned long long) * kWalkCtlWords, st));
    hipEvent_t ev_start = sim->pipe_events[3 * kMaxWalkGroups];
    if (sB != st || sS != st) {
        HIP_TRY(hipEventRecord(ev_start, st));
        if (sB != st) HIP_TRY(hipStreamWaitEvent(sB, ev_start, 0));
        if (sS != st) HIP_TRY(hipStreamWaitEvent(sS, ev_start, 0));
    }
    const bool hist = d.policy == RG_POLICY_ORGANIC_USER_COUNT;
    const size_t smem = (kBlock / 64) * walk2_wave_lds(hist);
    const walk_kernel_t wk = walk2_kernel_for(d, sim->walk_occ);
    const solo_kernel_t sk = solo_kernel_for(d);
    const exact_h_kernel_t kh = exact_h_kernel_for(d.XKB);
    if (!wk || !sk || !kh) return fail(RG_ESTATE, "run_walk_pipe: no kernel for this configuration");
    const uint32_t mfma_of_8 = static_cast<uint32_t>(sim->opt.exact_mix);
    auto walk_chunk = [&](uint64_t n_work, int blocks) {
        uint64_t chunk = n_work * 100 / (static_cast<uint64_t>(blocks) * 4 * 32);
        chunk = chunk / 64 * 64;
        if (chunk < 256) chunk = 256;
        if (chunk > 4096) chunk = 4096;
        return static_cast<uint32_t>(chunk);
    };
    unsigned long long* ctl_last = d.walk_ctl + 8 * kMaxWalkGroups;
    const uint32_t base_solo = ((n + 63u) & ~63u) + kMaxWalkGroups * kParkSlack;    // behind every group's region
    for (uint32_t g = 0; g < G; ++g) {
        DevSim dg = d;
        dg.grp_lo = g * gsz;
        dg.grp_n = n - dg.grp_lo < gsz ? n - dg.grp_lo : gsz;
        unsigned long long* ctl = d.walk_ctl + 8 * g;
        const uint32_t region = dg.grp_lo + g * kParkSlack;
        // ---- sweep, finalize ----
        {
            DevSim ds = dg;
            ds.sweep_only = 2u;
            ds.fin_in_sweep = dg.fin_in_sweep = sim->fin_in_sweep ? 1u : 0u;
            const uint32_t tiles_up = (dg.grp_n + sim->draw_users - 1) / sim->draw_users;
            if (int rc = span_begin(0, sS)) return rc;
            hipLaunchKernelGGL(sim->bf16_kernel, dim3(sweep_grid(sim, tiles_up, 1)), dim3(sim->draw_threads), sim->bf16_smem, sS, ds, 0u, 1u);
            if (int rc = span_end(sS)) return rc;
            if (int rc = span_begin(1, sS)) return rc;
            hipLaunchKernelGGL(finalize_kernel_for(d), dim3(grid_for(dg.grp_n)), dim3(kBlock), 0, sS, dg);
            hipLaunchKernelGGL(cache_prefix_kernel(), dim3(grid_for((static_cast<uint64_t>(dg.grp_n) + 7) / 8, kBlock / 64)), dim3(kBlock), 0, sS, dg, 1);
            if (int rc = span_end(sS)) return rc;
            if (sS != sA) {
                HIP_TRY(hipEventRecord(sim->pipe_events[3 * g], sS));
                HIP_TRY(hipStreamWaitEvent(sA, sim->pipe_events[3 * g], 0));
            }
        }
        // ---- round 1 ----
        {
            DevSim dw = dg;
            dw.q_ticket = ctl + 0; dw.q_park = ctl + 1; dw.q_count = nullptr;
            int blocks = static_cast<int>((static_cast<uint64_t>(dg.grp_n) + kBlock - 1) / kBlock);
            if (blocks > n_cus * sim->pipe_occ1) blocks = n_cus * sim->pipe_occ1;
            if (blocks > static_cast<int>(kMaxWalkWaves / 4)) blocks = kMaxWalkWaves / 4;
            if (int rc = span_begin(2, sA)) return rc;
            hipLaunchKernelGGL(wk, dim3(blocks), dim3(kBlock), smem, sA, dw, dg.grp_n, 1, walk_chunk(dg.grp_n, blocks), 0u, region);
            if (int rc = span_end(sA)) return rc;
            if (sB != sA) {
                HIP_TRY(hipEventRecord(sim->pipe_events[3 * g + 1], sA));
                HIP_TRY(hipStreamWaitEvent(sB, sim->pipe_events[3 * g + 1], 0));
            }
        }
        // ---- the users it parked: float64 sums, prefixes, round 2 (what it hands over: the last round's list) ----
        {
            DevSim dx = dg;
            dx.q_ticket = ctl + 2; dx.q_count = ctl + 1; dx.list_in = region;
            const uint32_t est = dg.grp_n / 3 + 4096u;                  // launch shapes only: the lengths are read on the device
            uint32_t xgrid = (dg.grp_n + 255u) / 256u;
            if (xgrid > static_cast<uint32_t>(sim->pipe_xblocks)) xgrid = static_cast<uint32_t>(sim->pipe_xblocks);
            if (int rc = span_begin(3, sB)) return rc;
            hipLaunchKernelGGL(kh, dim3(xgrid), dim3(kBlock), exact_m_lds(d.XKB), sB, dx, dg.grp_n, mfma_of_8);
            hipLaunchKernelGGL(exact_prefix_kernel(), dim3(grid_for(est, kBlock / 64)), dim3(kBlock), 0, sB, dx, dg.grp_n);
            if (int rc = span_end(sB)) return rc;
            DevSim dr = dg;
            dr.q_ticket = ctl + 3; dr.q_park = ctl_last + 0; dr.q_count = ctl + 1;
            int blocks = static_cast<int>((static_cast<uint64_t>(est) + kBlock - 1) / kBlock);
            if (blocks > n_cus * sim->pipe_occ2) blocks = n_cus * sim->pipe_occ2;
            if (blocks > static_cast<int>(kMaxWalkWaves / 4)) blocks = kMaxWalkWaves / 4;
            if (int rc = span_begin(4, sB)) return rc;
            hipLaunchKernelGGL(wk, dim3(blocks), dim3(kBlock), smem, sB, dr, dg.grp_n, 2, walk_chunk(est, blocks), region, base_solo);
            if (int rc = span_end(sB)) return rc;
            if (sB != sA && g + 1 == G) {
                HIP_TRY(hipEventRecord(sim->pipe_events[3 * g + 2], sB));
                HIP_TRY(hipStreamWaitEvent(sA, sim->pipe_events[3 * g + 2], 0));
            }
        }
    }
    // ---- last round: a wave per user (k_walk_solo) over what the rounds 2 handed over ----
    sim->fate_base = base_solo; sim->fate_count = ctl_last + 0;
    if (d.walk_handover) {
        DevSim dl = d;
        dl.q_ticket = ctl_last + 1; dl.q_count = ctl_last + 0;
        const uint32_t est = n / 256u + 1024u;
        uint32_t blocks = (est + 15u) / 16u;
        const uint32_t cap = static_cast<uint32_t>(n_cus) * 8u;
        if (blocks > cap) blocks = cap;
        uint64_t chunk = static_cast<uint64_t>(est) * 150 / (static_cast<uint64_t>(blocks) * 4 * 8);
        chunk = chunk / 64 * 64;
        if (chunk < 64) chunk = 64;
        if (chunk > 1024) chunk = 1024;
        if (int rc = span_begin(4, sA)) return rc;
        hipLaunchKernelGGL(sk, dim3(blocks), dim3(kBlock), 0, sA, dl, n, static_cast<uint32_t>(chunk), base_solo);
        if (int rc = span_end(sA)) return rc;
    }
    hipLaunchKernelGGL(k_walk_finish, dim3(1), dim3(1), 0, st, d);
    HIP_TRY(hipGetLastError());
    if (sim->profiling) HIP_TRY(hipEventRecord(wall[1], st));
    unsigned long long* h64 = reinterpret_cast<unsigned long long*>(sim->h_pinned);
    HIP_TRY(hipMemcpyAsync(h64, d.counters + kCntTailLimit, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (sim->profiling) {
        for (const Span& sp : spans) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, sp.a, sp.b));
            if (sp.cls == 0) sim->prof_ms[0] += ms;
            else if (sp.cls == 1) sim->prof_ms[1] += ms;
            else if (sp.cls == 3) sim->prof_ms[2] += ms;
            else { sim->prof_walk_ms[sp.cls == 2 ? 0 : 1] += ms; sim->prof_tail_ms += ms; }
            (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b);
        }
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, wall[0], wall[1]));
        sim->prof_pipe_ms += ms;
        (void)hipEventDestroy(wall[0]); (void)hipEventDestroy(wall[1]);
        sim->prof_launches += 1;
    }
    sim->t = 1;
    sim->live_upper = 0;
    if (*h64) return fail(RG_ELIMIT, "more than %u steps", kMaxSteps);
    return RG_OK;
}

#endif  // RG_HAS(1): host code
}  // namespace rgk

#if RG_HAS(1)
// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

const char* rg_last_error(void) { return g_err; }
int rg_abi_version(void) { return RG_ABI_VERSION; }

int rg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t rg_sim_workspace_bytes(const rg_config* cfg, uint64_t n_users) {
    if (validate(cfg, n_users) != RG_OK) return 0;
    return carve_all(*cfg, n_users, nullptr, nullptr);
}

int rg_sim_create(rg_sim** out, const rg_config* cfg, uint64_t n_users, void* d_workspace,
                  size_t workspace_bytes) {
    if (!out) return fail(RG_EINVAL, "out is NULL");
    *out = nullptr;
    if (int rc = validate(cfg, n_users)) return rc;
    if (!d_workspace) return fail(RG_EINVAL, "workspace is NULL");
    const size_t need = carve_all(*cfg, n_users, nullptr, nullptr);
    if (workspace_bytes < need)
        return fail(RG_ENOMEM, "workspace has %zu bytes, %zu needed", workspace_bytes, need);
    if ((reinterpret_cast<uintptr_t>(d_workspace) & 255u) != 0)
        return fail(RG_EINVAL, "workspace must be 256-byte aligned");
    rg_sim* s = new (std::nothrow) rg_sim();
    if (!s) return fail(RG_ENOMEM, "host allocation failed");
    s->cfg = *cfg;
    s->workspace = d_workspace;
    s->workspace_bytes = workspace_bytes;
    DevSim& d = s->d;
    memset(&d, 0, sizeof(d));
    carve_all(*cfg, n_users, d_workspace, &d);
    d.P = cfg->num_products; d.K = cfg->K;
    d.seed = cfg->seed; d.policy_seed = cfg->policy_seed;
    d.cdf_o0 = cfg->trans_cdf[0][0]; d.cdf_o1 = cfg->trans_cdf[0][1];
    d.cdf_b0 = cfg->trans_cdf[1][0]; d.cdf_b1 = cfg->trans_cdf[1][1];
    d.sigma0 = cfg->sigma_omega_initial; d.sigma_omega = cfg->sigma_omega;
    d.change_omega_for_bandits = cfg->change_omega_for_bandits;
    d.policy = cfg->policy;
    d.ouc_select_randomly = cfg->ouc_select_randomly;
    d.ouc_exploit_explore = cfg->ouc_exploit_explore;
    d.ouc_reverse_pop = cfg->ouc_reverse_pop;
    d.ouc_epsilon = cfg->ouc_epsilon;
    d.time_mode = cfg->time_mode; d.time_mu = cfg->time_mu; d.time_sigma = cfg->time_sigma;
    d.n_users = d.n_cap = static_cast<uint32_t>(n_users);
    s->h_pinned = nullptr; s->h_step = nullptr;
    {   // run-path options from the environment, once
        RunOpts& o = s->opt;
        o.exact_tile = getenv("RECOGYM_EXACT_TILE") ? 1 : 0;
        o.exact_mix = 5;
        if (const char* e = getenv("RECOGYM_EXACT_MIX")) o.exact_mix = atoi(e);
        o.resident_grid = getenv("RECOGYM_RESIDENT_GRID") ? 1 : 0;
        o.slices = -1;
        if (const char* e = getenv("RECOGYM_SLICES")) o.slices = atoi(e);
        o.sweep_prefix_off = getenv("RECOGYM_SWEEP_PREFIX_OFF") ? 1 : 0;
        o.debug = getenv("RECOGYM_DEBUG") ? 1 : 0;
        o.repack_min = repack_min_users();
        const char* e_h = getenv("RECOGYM_WALK_HIST");
        d.walk_line64 = (e_h && e_h[0] == '1') ? 1u : 0u;
    }
    s->profiling = false; s->prof_used = 0; s->prof_launches = 0;
    s->prof_ms[0] = s->prof_ms[1] = s->prof_ms[2] = s->prof_ms[3] = s->prof_ms[4] = 0.0;
    s->mfma_smem = d.use_mfma ? mfma_smem_bytes(geom_of(*cfg)) : 0;
    // kernel choice: split-bf16 MFMA when a class exists for K, else fp32 MFMA; RECOGYM_DRAW=f64|fp32|bf16 overrides
    s->bf16_kernel = nullptr; s->bf16_smem = 0;
    s->draw_threads = kBlock; s->draw_users = 128;
    if (d.use_mfma && d.N1) {
        s->bf16_kernel = d.f16 ? nullptr : bf16_kernel_for(d);
        // the pipelined form (two chunks in flight, exp-sum and operand loads inside the MFMA stream)
        // where its ~200 VGPRs fit; RECOGYM_BF16=lean keeps the one-accumulator kernel (A/B tests)
        const char* lean = getenv("RECOGYM_BF16");
        if (!(lean && !strcmp(lean, "lean")))
            if (static_cast<size_t>(d.P_pad) * d.RS < (1ull << 31))     // its DMA uses 31-bit buffer offsets
                if (draw_kernel_t kp = bf16p_kernel_for(d)) s->bf16_kernel = kp;
        s->bf16_smem = bf16_smem_bytes(geom_of(*cfg), 2 * d.KH, s->bf16_kernel == bf16p_kernel_for(d) ? 3u : 2u);
        if (d.wide) {
            s->bf16_kernel = static_cast<size_t>(d.P_pad) * d.RS < (1ull << 31) ? f16w_kernel_for(d) : nullptr;
            s->bf16_smem = 3 * (64 * static_cast<size_t>(d.RS) + 256) + 8 * 32 * 2 * static_cast<size_t>(d.KH) * 4;
            s->draw_threads = 512 / f16w_ug(); s->draw_users = 256;
        }
        // the larger classes still spill registers; the fp32 kernel is faster there for now
        if (s->bf16_kernel && (d.f16 || (d.N1 <= 4 && d.KH <= 10))) d.use_mfma = 2;
    }
    if (const char* e = getenv("RECOGYM_DRAW")) {
        if (!strcmp(e, "f64")) d.use_mfma = 0;
        else if (!strcmp(e, "fp32") && d.KH) d.use_mfma = 1;
        else if ((!strcmp(e, "bf16") || !strcmp(e, "f16")) && s->bf16_kernel) d.use_mfma = 2;
    }
    if (const char* e = getenv("RECOGYM_FORCE_EXACT")) if (e[0] == '1') d.use_mfma = 0;   // A/B switch for tests
    // the per-user sum cache is written by the pipelined 16-bit kernel only
    if (!(d.use_mfma == 2 && s->bf16_kernel &&
          (s->bf16_kernel == bf16p_kernel_for(d) || (d.wide && s->bf16_kernel == f16w_kernel_for(d))))) d.use_cache = 0;
    if (s->bf16_kernel && s->bf16_smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s->bf16_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(s->bf16_smem));
    d.ablate = 0;
    s->repack_every = 16;
    s->repacked = false;
    {   // a per-user draw streams the whole Gamma table through one CU: the population at which the tail
        // kernel beats the latency floor of the lock-step steps shrinks with P * K (4096 users at 10^4 x 20)
        const double scale = 2.0e5 / (static_cast<double>(d.P) * static_cast<double>(d.K));
        // (only where the lock-step draw kernel slices products for small populations; the fp32 and float64
        // kernels sweep all P per step, so for them the tail kernel wins much earlier)
        const double tb = 4096.0 * ((scale < 1.0 && d.use_mfma == 2) ? scale : 1.0);
        s->tail_below = tb < 64.0 ? 64u : static_cast<uint32_t>(tb);
        // the tail kernel runs a policy on one thread: fine for a history walk, not for n_classes x views
        // score loops — the frozen LogReg policy stays in lock-step (wave-cooperative acts) to the end
        if (d.policy == RG_POLICY_LOGREG_FROZEN) s->tail_below = 0;
    }
    s->prof_tail_ms = 0.0;
    // user-major walk: wherever the per-user cache exists and the policy acts lane by lane (the frozen LogReg
    // policy acts wave-cooperatively: lock-step); RECOGYM_WALK=0 keeps the lock-step loop (A/B tests)
    s->walk = d.use_cache && (d.policy == RG_POLICY_UNIFORM_ENV || d.policy == RG_POLICY_RANDOM_AGENT ||
                              d.policy == RG_POLICY_ORGANIC_USER_COUNT || d.policy == RG_POLICY_LAST_VIEW_TABLE);
    if (const char* e = getenv("RECOGYM_WALK")) if (e[0] == '0') s->walk = false;
    if (d.time_mode) { s->walk = false; s->tail_below = 0; }      // per-user clocks: the lock-step kernels only
    s->n_cus = 0;
    s->walk_occ = 3;
    d.walk_bias = 8;
    d.walk_refill = 8;
    d.walk_handover = 32;
    if (const char* e = getenv("RECOGYM_WALK_HANDOVER")) d.walk_handover = static_cast<uint32_t>(atoi(e));
    d.walk_click_batch = 8;
    if (const char* e = getenv("RECOGYM_WALK_CLICK_BATCH")) d.walk_click_batch = static_cast<uint32_t>(atoi(e));
    d.walk_search_batch = 16;
    if (const char* e = getenv("RECOGYM_WALK_SEARCH_BATCH")) d.walk_search_batch = static_cast<uint32_t>(atoi(e)) ? static_cast<uint32_t>(atoi(e)) : 1u;
    if (const char* e = getenv("RECOGYM_WALK_REFILL")) d.walk_refill = static_cast<uint32_t>(atoi(e));
    if (const char* e = getenv("RECOGYM_WALK_BIAS")) d.walk_bias = static_cast<uint32_t>(atoi(e));
    // k_walk2 where it is instantiated for the configuration (RECOGYM_WALK=1: k_walk), four blocks per CU at K <= 20
    s->walk2 = s->walk && walk2_kernel_for(d, 4) != nullptr;
    if (const char* e = getenv("RECOGYM_WALK")) if (e[0] == '1') s->walk2 = false;
    // (k_walk2: 4 since the act is a count on the compact history line — the bandit iteration got shorter, so the organic kind
    // waits for more lanes: profiles/r4/ab_call6_walk_bias.jsonl; k_walk keeps round 2's 8)
    if (s->walk2 && !getenv("RECOGYM_WALK_BIAS")) d.walk_bias = 4;
    if (s->walk2) s->walk_occ = d.KH <= 10 ? 3 : 2;     // (what k_walk2 is compiled for: K <= 20 three blocks per CU, K <= 32 two)
    s->walk_solo = true;
    if (const char* e = getenv("RECOGYM_WALK_SOLO")) s->walk_solo = e[0] != '0';
    s->prof_walk_ms[0] = s->prof_walk_ms[1] = 0.0;
    // the walked run as a pipeline over user groups (run_walk_pipe).  RECOGYM_PIPE=G (0: run_walk, host-side list lengths),
    // RECOGYM_PIPE_MODE=0|1|2, RECOGYM_PIPE_OCC1 / _OCC2 (blocks per CU of the rounds' grids), RECOGYM_PIPE_XBLOCKS: A/B tests
    // Default: ONE group (the serial chain, every list length read on the device: no host read-back between the launches).
    // More groups on two or three streams were measured on C3 and do not pay (profiles/r4/ab_call1_pipe_forms.jsonl, DESIGN.md
    // 3a): the walk's three waves per SIMD fill the register file, so nothing co-resides with it, and every group adds a
    // drain tail to both walk rounds and a partial last wave of blocks to the float64 batch.
    s->pipe_groups = 1; s->pipe_mode = 1;
    s->pipe_occ1 = s->pipe_occ2 = s->walk_occ;
    s->pipe_xblocks = 1024;
    s->pipe_streams[0] = s->pipe_streams[1] = nullptr;
    s->fate_base = 0; s->fate_count = nullptr;
    s->prof_pipe_ms = 0.0;
    if (const char* e = getenv("RECOGYM_PIPE")) s->pipe_groups = atoi(e);
    if (const char* e = getenv("RECOGYM_PIPE_MODE")) s->pipe_mode = atoi(e);
    if (const char* e = getenv("RECOGYM_PIPE_OCC1")) { const int o = atoi(e); if (o >= 1 && o <= s->walk_occ) s->pipe_occ1 = o; }
    if (const char* e = getenv("RECOGYM_PIPE_OCC2")) { const int o = atoi(e); if (o >= 1 && o <= s->walk_occ) s->pipe_occ2 = o; }
    if (const char* e = getenv("RECOGYM_PIPE_XBLOCKS")) { const int o = atoi(e); if (o >= 1) s->pipe_xblocks = o; }
    s->fin_in_sweep = true;
    if (const char* e = getenv("RECOGYM_FIN_IN_SWEEP")) s->fin_in_sweep = e[0] != '0';
    d.fin_in_sweep = 0;
    s->pipe_min_users = 1u << 17;
    if (const char* e = getenv("RECOGYM_PIPE_MIN")) { const int o = atoi(e); if (o >= 256) s->pipe_min_users = static_cast<uint32_t>(o); }
    d.grp_lo = 0; d.grp_n = d.n_users; d.list_in = 0;
    d.q_ticket = d.counters + kCntWalkTicket; d.q_park = d.counters + kCntParkCnt; d.q_count = nullptr;
    if (const char* e = getenv("RECOGYM_TAIL")) s->tail_below = static_cast<uint32_t>(atoi(e));
    if (const char* e = getenv("RECOGYM_REPACK")) s->repack_every = static_cast<uint32_t>(atoi(e));
    if (const char* e = getenv("RECOGYM_ABLATE")) d.ablate = static_cast<uint32_t>(atoi(e));
    if (const char* e = getenv("RECOGYM_LDS_PAD")) s->mfma_smem += static_cast<size_t>(atoi(e));
    if (s->opt.debug && d.use_mfma && rg_device_count() > 0) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(mfma_kernel_for(10)), kBlock, s->mfma_smem);
        fprintf(stderr, "[recogym] k_draw_mfma<10>: dynamic LDS %zu B, occupancy API %d blocks/CU\n", s->mfma_smem, nb);
    }
    if (s->mfma_smem > 64 * 1024) {
        // more than 64 KiB of dynamic LDS needs an explicit opt-in per kernel instantiation
        const int bytes = static_cast<int>(s->mfma_smem);
        for (uint32_t kh : {4u, 10u, 16u, 32u, 64u})
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel_for(kh)), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
    *out = s;
    return RG_OK;
}

int rg_sim_destroy(rg_sim* sim) {
    if (!sim) return RG_OK;
    if (sim->h_pinned) (void)hipHostFree(sim->h_pinned);
    if (sim->h_step) (void)hipHostFree(sim->h_step);
    for (hipEvent_t e : sim->prof_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : sim->pipe_events) (void)hipEventDestroy(e);
    for (hipStream_t ps : sim->pipe_streams) if (ps) (void)hipStreamDestroy(ps);
    delete sim;
    return RG_OK;
}

// name -> the field it sets; every entry is a run-path tuning knob (none changes the workspace layout or a result)
namespace {
int* opt_int(rg_sim* s, const char* n) {
    if (!strcmp(n, "pipe_groups")) return &s->pipe_groups;
    if (!strcmp(n, "pipe_mode")) return &s->pipe_mode;
    if (!strcmp(n, "pipe_occ1")) return &s->pipe_occ1;
    if (!strcmp(n, "pipe_occ2")) return &s->pipe_occ2;
    if (!strcmp(n, "pipe_xblocks")) return &s->pipe_xblocks;
    if (!strcmp(n, "exact_mix")) return &s->opt.exact_mix;
    if (!strcmp(n, "exact_tile")) return &s->opt.exact_tile;
    if (!strcmp(n, "resident_grid")) return &s->opt.resident_grid;
    if (!strcmp(n, "slices")) return &s->opt.slices;
    if (!strcmp(n, "sweep_prefix_off")) return &s->opt.sweep_prefix_off;
    if (!strcmp(n, "debug")) return &s->opt.debug;
    return nullptr;
}
uint32_t* opt_u32(rg_sim* s, const char* n) {
    if (!strcmp(n, "walk_bias")) return &s->d.walk_bias;
    if (!strcmp(n, "walk_refill")) return &s->d.walk_refill;
    if (!strcmp(n, "walk_handover")) return &s->d.walk_handover;
    if (!strcmp(n, "walk_click_batch")) return &s->d.walk_click_batch;
    if (!strcmp(n, "walk_search_batch")) return &s->d.walk_search_batch;
    if (!strcmp(n, "walk_line64")) return &s->d.walk_line64;
    if (!strcmp(n, "pipe_min_users")) return &s->pipe_min_users;
    if (!strcmp(n, "tail_below")) return &s->tail_below;
    if (!strcmp(n, "repack_every")) return &s->repack_every;
    return nullptr;
}
}  // namespace

int rg_sim_set_option(rg_sim* sim, const char* name, int64_t value) {
    if (!sim || !name) return fail(RG_EINVAL, "NULL argument");
    if (int* p = opt_int(sim, name)) {
        if ((!strcmp(name, "pipe_occ1") || !strcmp(name, "pipe_occ2")) && (value < 1 || value > sim->walk_occ))
            return fail(RG_EINVAL, "%s must be in [1, %d]", name, sim->walk_occ);
        if (!strcmp(name, "pipe_xblocks") && value < 1) return fail(RG_EINVAL, "pipe_xblocks must be >= 1");
        if (!strcmp(name, "exact_mix") && (value < 0 || value > 8)) return fail(RG_EINVAL, "exact_mix must be in [0, 8]");
        *p = static_cast<int>(value);
        return RG_OK;
    }
    if (uint32_t* p = opt_u32(sim, name)) {
        if (value < 0) return fail(RG_EINVAL, "%s must be >= 0", name);
        if (!strcmp(name, "walk_search_batch") && value < 1) value = 1;
        if (!strcmp(name, "pipe_min_users") && value < 256) return fail(RG_EINVAL, "pipe_min_users must be >= 256");
        *p = static_cast<uint32_t>(value);
        return RG_OK;
    }
    return fail(RG_EINVAL, "unknown option '%s'", name);
}

int rg_sim_get_option(rg_sim* sim, const char* name, int64_t* value) {
    if (!sim || !name || !value) return fail(RG_EINVAL, "NULL argument");
    if (int* p = opt_int(sim, name)) { *value = *p; return RG_OK; }
    if (uint32_t* p = opt_u32(sim, name)) { *value = *p; return RG_OK; }
    return fail(RG_EINVAL, "unknown option '%s'", name);
}

int rg_sim_set_tables(rg_sim* sim, const double* d_gamma, const double* d_mu_organic,
                      const double* d_beta, const double* d_mu_bandit, void* stream) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (!d_gamma || !d_mu_organic || !d_beta || !d_mu_bandit) return fail(RG_EINVAL, "table pointer is NULL");
    if (rg_device_count() <= 0) return fail(RG_ENODEV, "no HIP device");
    sim->d.gamma = d_gamma; sim->d.mu_o = d_mu_organic; sim->d.beta = d_beta; sim->d.mu_b = d_mu_bandit;
    hipLaunchKernelGGL(k_make_gammaT, dim3(grid_for(static_cast<size_t>(sim->d.K) * sim->d.PT)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), sim->d);
    if (sim->d.XKB)
        hipLaunchKernelGGL(k_make_gamma_rm, dim3(grid_for(static_cast<size_t>(sim->d.PT) * (4 * sim->d.XKB + 4))), dim3(kBlock), 0,
                           static_cast<hipStream_t>(stream), sim->d);
    if (sim->d.beta32)
        hipLaunchKernelGGL(k_make_beta32, dim3(grid_for(static_cast<size_t>(sim->d.P) * sim->d.KB4)), dim3(kBlock), 0,
                           static_cast<hipStream_t>(stream), sim->d);
    if (sim->d.use_mfma) {
        const size_t n = static_cast<size_t>(sim->d.P_pad) * sim->d.KS;
        hipLaunchKernelGGL(k_make_fp32_tables, dim3(grid_for(n)), dim3(kBlock), 0,
                           static_cast<hipStream_t>(stream), sim->d);
        hipLaunchKernelGGL(k_table_stats, dim3(2 * sim->d.KH + 2 + kAhatGrid), dim3(kBlock), 0,
                           static_cast<hipStream_t>(stream), sim->d);
        if (sim->d.N1)
            hipLaunchKernelGGL(k_make_split_table, dim3(grid_for(static_cast<size_t>(sim->d.P_pad) * (sim->d.RS / 2))),
                               dim3(kBlock), 0, static_cast<hipStream_t>(stream), sim->d);
    }
    HIP_TRY(hipGetLastError());
    sim->tables_set = true;
    return RG_OK;
}

int rg_sim_set_policy_table(rg_sim* sim, const int32_t* d_action, const float* d_ps) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (sim->d.policy != RG_POLICY_LAST_VIEW_TABLE) return fail(RG_ESTATE, "policy is not RG_POLICY_LAST_VIEW_TABLE");
    if (!d_action) return fail(RG_EINVAL, "action table is NULL");
    sim->d.pol_table = d_action;
    sim->d.pol_ps = d_ps;
    return RG_OK;
}

int rg_sim_set_logreg(rg_sim* sim, const double* d_coef_t, const double* d_intercept,
                      const int32_t* d_classes, uint32_t n_classes) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (sim->d.policy != RG_POLICY_LOGREG_FROZEN) return fail(RG_ESTATE, "policy is not RG_POLICY_LOGREG_FROZEN");
    if (!d_coef_t || !d_intercept || !d_classes || n_classes == 0) return fail(RG_EINVAL, "NULL model array or no classes");
    sim->d.lr_coef_t = d_coef_t; sim->d.lr_intercept = d_intercept; sim->d.lr_classes = d_classes;
    sim->d.lr_n = n_classes;
    // a new model invalidates the optional copies of the old one (their shapes and bounds belong to it): set them again
    sim->d.lr_coef32_t = nullptr; sim->d.lr_intercept32 = nullptr; sim->d.lr_wmax = nullptr; sim->d.lr_bmax = 0.0f;
    sim->d.lr_coef16_t = nullptr;
    return RG_OK;
}

int rg_sim_set_logreg_fp32(rg_sim* sim, const float* d_coef32_t, const float* d_intercept32, const float* d_wmax, float bmax) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (sim->d.policy != RG_POLICY_LOGREG_FROZEN) return fail(RG_ESTATE, "policy is not RG_POLICY_LOGREG_FROZEN");
    if (!sim->d.lr_coef_t) return fail(RG_ESTATE, "rg_sim_set_logreg must be called first");
    if ((d_coef32_t || d_intercept32 || d_wmax) && !(d_coef32_t && d_intercept32 && d_wmax)) return fail(RG_EINVAL, "all three arrays or none");
    if (!(bmax >= 0.0f)) return fail(RG_EINVAL, "bmax must be >= 0");
    sim->d.lr_coef32_t = d_coef32_t; sim->d.lr_intercept32 = d_intercept32; sim->d.lr_wmax = d_wmax; sim->d.lr_bmax = bmax;
    sim->d.lr_coef16_t = nullptr;      // the screening pass reads intercept32 / wmax / bmax: attach it again after this call
    return RG_OK;
}

int rg_sim_set_logreg_fp16(rg_sim* sim, const uint16_t* d_coef16_t) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (sim->d.policy != RG_POLICY_LOGREG_FROZEN) return fail(RG_ESTATE, "policy is not RG_POLICY_LOGREG_FROZEN");
    if (d_coef16_t && !sim->d.lr_coef32_t) return fail(RG_ESTATE, "rg_sim_set_logreg_fp32 must be called first (intercept32, wmax, bmax)");
    if (d_coef16_t && sim->d.lr_n % 8u) return fail(RG_EINVAL, "the fp16 screening pass needs n_classes %% 8 == 0 (have %u)", sim->d.lr_n);
    sim->d.lr_coef16_t = d_coef16_t;
    return RG_OK;
}

int rg_sim_set_log(rg_sim* sim, rg_event* d_log, uint64_t capacity) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    sim->d.log = capacity ? d_log : nullptr;
    sim->d.log_cap = d_log ? capacity : 0;
    sim->d.aux_ps = nullptr; sim->d.aux_pclick = nullptr; sim->d.aux_time = nullptr;     // side arrays are sized with the log: re-attach
    return RG_OK;
}

int rg_sim_set_log_aux(rg_sim* sim, double* d_ps, double* d_p_click) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if ((d_ps || d_p_click) && !sim->d.log) return fail(RG_ESTATE, "attach a log buffer first (rg_sim_set_log)");
    sim->d.aux_ps = d_ps; sim->d.aux_pclick = d_p_click;
    return RG_OK;
}

int rg_sim_reseed(rg_sim* sim, uint64_t seed, uint64_t policy_seed) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    sim->cfg.seed = sim->d.seed = seed;
    sim->cfg.policy_seed = sim->d.policy_seed = policy_seed;
    return RG_OK;
}

int rg_sim_reset_users(rg_sim* sim, uint64_t first_user_id, uint64_t n, uint64_t organic_only_below,
                       void* stream) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (!sim->tables_set) return fail(RG_ESTATE, "rg_sim_set_tables must be called first");
    if (sim->d.policy == RG_POLICY_LAST_VIEW_TABLE && !sim->d.pol_table)
        return fail(RG_ESTATE, "rg_sim_set_policy_table must be called first");
    if (sim->d.policy == RG_POLICY_LOGREG_FROZEN && !sim->d.lr_coef_t)
        return fail(RG_ESTATE, "rg_sim_set_logreg must be called first");
    if (n == 0 || n > sim->d.n_cap) return fail(RG_EINVAL, "n %llu exceeds the %u users the workspace was sized for",
                                                 (unsigned long long)n, sim->d.n_cap);
    if (first_user_id + n > (1ull << 32)) return fail(RG_EINVAL, "user ids must fit 32 bits");
    if (rg_device_count() <= 0) return fail(RG_ENODEV, "no HIP device");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DevSim& d = sim->d;
    d.first_user = first_user_id;
    d.organic_only_below = organic_only_below;
    d.n_users = static_cast<uint32_t>(n);      // lists stay strided by the carve-time n_cap
    d.grp_lo = 0; d.grp_n = d.n_users;
    HIP_TRY(hipMemsetAsync(d.step_cnt, 0, sizeof(uint32_t) * 2 * (kMaxSteps + 2), st));
    HIP_TRY(hipMemsetAsync(d.exact_cnt, 0, sizeof(uint32_t) * (kMaxSteps + 2), st));
    HIP_TRY(hipMemsetAsync(d.exact_cnt_b, 0, sizeof(uint32_t) * (kMaxSteps + 2), st));
    if (d.lr_dirty) HIP_TRY(hipMemsetAsync(d.lr_cnt, 0, sizeof(uint32_t) * (kMaxSteps + 2), st));
    if (d.sigma_omega != 0.0) HIP_TRY(hipMemsetAsync(d.drift_cnt, 0, sizeof(uint32_t) * (kMaxSteps + 2), st));
    HIP_TRY(hipMemsetAsync(d.counters, 0, sizeof(unsigned long long) * RG_CNT_N, st));
    hipLaunchKernelGGL(k_reset_users, dim3(grid_for(n)), dim3(kBlock), 0, st, d);
    HIP_TRY(hipGetLastError());
    sim->t = 0;
    sim->live_upper = static_cast<uint32_t>(n);
    sim->users_reset = true;
    sim->repacked = false;
    return RG_OK;
}

#ifdef RG_F16W_TIMING
void rg_debug_f16w_timing(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f16w_t), sizeof(unsigned long long) * 8);
}
#endif

int rg_sim_step(rg_sim* sim, const int32_t* d_actions, void* stream) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (!sim->users_reset) return fail(RG_ESTATE, "rg_sim_reset_users must be called first");
    if (sim->d.policy == RG_POLICY_EXTERNAL && !d_actions) return fail(RG_EINVAL, "external policy needs d_actions");
    return launch_step(sim, d_actions, static_cast<hipStream_t>(stream));
}

int rg_sim_step_user(rg_sim* sim, int32_t action, rg_step_result* out, void* stream) {
    if (!sim || !out) return fail(RG_EINVAL, "NULL argument");
    if (!sim->users_reset) return fail(RG_ESTATE, "rg_sim_reset_users must be called first");
    if (sim->d.policy != RG_POLICY_EXTERNAL || sim->d.n_users != 1) return fail(RG_ESTATE, "rg_sim_step_user needs RG_POLICY_EXTERNAL and a one-user reset range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!sim->h_step) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sim->h_step), 128));
    int32_t* h_act = reinterpret_cast<int32_t*>(sim->h_step);
    *h_act = action;
    HIP_TRY(hipMemcpyAsync(sim->d.step1_buf, h_act, sizeof(int32_t), hipMemcpyHostToDevice, st));
    const uint32_t t = sim->t;
    if (int rc = launch_step(sim, reinterpret_cast<const int32_t*>(sim->d.step1_buf), st)) return rc;
    hipLaunchKernelGGL(k_step_user_pack, dim3(1), dim3(1), 0, st, sim->d, t);
    HIP_TRY(hipGetLastError());
    rg_step_result* h_res = reinterpret_cast<rg_step_result*>(sim->h_step + 64);
    HIP_TRY(hipMemcpyAsync(h_res, sim->d.step1_buf + 8, sizeof(rg_step_result), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *out = *h_res;
    return RG_OK;
}

int rg_sim_run(rg_sim* sim, uint32_t max_steps, void* stream) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (!sim->users_reset) return fail(RG_ESTATE, "rg_sim_reset_users must be called first");
    if (sim->d.policy == RG_POLICY_EXTERNAL) return fail(RG_ESTATE, "rg_sim_run needs a device policy");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!sim->h_pinned) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sim->h_pinned), 4 * sizeof(uint32_t)));
    if (sim->walk && sim->t == 0 && max_steps >= kMaxSteps) {
        // the pipelined form where its kernels exist (k_walk2 behind the fused-prefix fp16 sweep, k_walk_solo, the mixed float64
        // batch) and the reset range fills an unsliced sweep; else the serial chain with its list lengths read back by the host
        const int mix = sim->opt.exact_mix;
        const bool pipe = sim->pipe_groups >= 1 && sim->walk2 && sim->walk_solo && solo_kernel_for(sim->d) && sim->d.walk_handover &&
                          sim->bf16_kernel == bf16p_kernel_for(sim->d) && sim->d.f16 && !sim->d.wide && sim->d.n_users >= sim->pipe_min_users &&
                          exact_h_kernel_for(sim->d.XKB) && mix < 8 && sim->opt.slices < 0 &&
                          !sim->opt.sweep_prefix_off;
        return pipe ? run_walk_pipe(sim, st) : run_walk(sim, st);
    }
    uint32_t done_steps = 0;
    const uint32_t chunk = 16;
    while (done_steps < max_steps) {
        const uint32_t todo = (max_steps - done_steps) < chunk ? (max_steps - done_steps) : chunk;
        for (uint32_t i = 0; i < todo; ++i)
            if (int rc = launch_step(sim, nullptr, st)) return rc;
        done_steps += todo;
        HIP_TRY(hipMemcpyAsync(sim->h_pinned, sim->d.step_cnt + 2 * sim->t, 2 * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (int rc = prof_collect(sim)) return rc;
        const uint64_t live = static_cast<uint64_t>(sim->h_pinned[0]) + sim->h_pinned[1];
        sim->live_upper = static_cast<uint32_t>(live);
        if (live == 0) break;
        // few users left: finish them user by user (k_tail) instead of ~1 000 more latency-bound steps
        const size_t tail_smem = sizeof(double) * (((sim->d.K + 1) & ~1u) + ((sim->d.PT / 64 + 3) & ~3u));
        if (live <= sim->tail_below && max_steps >= kMaxSteps && tail_smem <= 48 * 1024 && sim->t + 2 < kMaxSteps) {
            hipEvent_t ev[2] = {nullptr, nullptr};
            if (sim->profiling) {
                HIP_TRY(hipEventCreate(&ev[0])); HIP_TRY(hipEventCreate(&ev[1]));
                HIP_TRY(hipEventRecord(ev[0], st));
            }
            const int grid = static_cast<int>(live < 2048 ? live : 2048);
            hipLaunchKernelGGL(tail_kernel(), dim3(grid), dim3(kBlock), tail_smem, st, sim->d, sim->t);
            hipLaunchKernelGGL(k_tail_finish, dim3(1), dim3(1), 0, st, sim->d, sim->t);
            HIP_TRY(hipGetLastError());
            if (sim->profiling) HIP_TRY(hipEventRecord(ev[1], st));
            unsigned long long* h64 = reinterpret_cast<unsigned long long*>(sim->h_pinned);
            HIP_TRY(hipMemcpyAsync(h64, sim->d.counters + kCntTailLimit, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (sim->profiling) {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
                sim->prof_tail_ms += ms;
                (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
            }
            sim->t += 1;
            sim->live_upper = 0;
            if (*h64) return fail(RG_ELIMIT, "more than %u steps", kMaxSteps);
            break;
        }
    }
    {   // an incomplete run is an error, not a counter to remember to look at
        unsigned long long* h64 = reinterpret_cast<unsigned long long*>(sim->h_pinned);
        HIP_TRY(hipMemcpyAsync(h64, sim->d.counters + RG_CNT_EXACT_OVERFLOW, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (*h64) return fail(RG_ELIMIT, "%llu uncertified organic draws exceeded the float64 resolve scratch: the run is incomplete", *h64);
    }
    return RG_OK;
}

int rg_sim_read_counters(rg_sim* sim, int64_t* out, void* stream) {
    if (!sim || !out) return fail(RG_EINVAL, "NULL argument");
    if (rg_device_count() <= 0) return fail(RG_ENODEV, "no HIP device");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_totals, dim3(1), dim3(kBlock), 0, st, sim->d, sim->t);
    HIP_TRY(hipGetLastError());
    unsigned long long h[RG_CNT_N];
    HIP_TRY(hipMemcpyAsync(h, sim->d.counters, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < RG_CNT_N; ++i) out[i] = static_cast<int64_t>(h[i]);
    sim->live_upper = static_cast<uint32_t>(h[RG_CNT_LIVE]);
    return RG_OK;
}

int rg_sim_set_profiling(rg_sim* sim, int on) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    sim->profiling = on != 0;
    sim->prof_used = 0; sim->prof_launches = 0;
    sim->prof_ms[0] = sim->prof_ms[1] = sim->prof_ms[2] = sim->prof_ms[3] = sim->prof_ms[4] = 0.0;
    sim->prof_tail_ms = 0.0;
    sim->prof_walk_ms[0] = sim->prof_walk_ms[1] = 0.0;
    sim->prof_pipe_ms = 0.0;
    return RG_OK;
}

int rg_sim_get_profile(rg_sim* sim, double* out) {
    if (!sim || !out) return fail(RG_EINVAL, "NULL argument");
    if (int rc = prof_collect(sim)) return rc;
    out[0] = sim->prof_ms[0]; out[1] = sim->prof_ms[1]; out[2] = sim->prof_ms[2]; out[3] = sim->prof_ms[4];
    out[4] = static_cast<double>(sim->prof_launches);
    out[5] = sim->prof_tail_ms;
    out[6] = sim->prof_walk_ms[0]; out[7] = sim->prof_walk_ms[1];
    out[8] = sim->prof_ms[3]; out[9] = sim->prof_pipe_ms;
    return RG_OK;
}

int rg_sim_export_state(rg_sim* sim, int8_t* d_state, void* stream) {
    if (!sim || !d_state) return fail(RG_EINVAL, "NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(d_state, RG_STATE_STOP, sim->d.n_users, st));
    hipLaunchKernelGGL(k_export_state, dim3(grid_for(sim->live_upper)), dim3(kBlock), 0, st, sim->d, sim->t, d_state);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_export_omega(rg_sim* sim, double* d_omega, void* stream) {
    if (!sim || !d_omega) return fail(RG_EINVAL, "NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (sim->repacked) {     // slots of users that left are gone: their rows read 0
        HIP_TRY(hipMemsetAsync(d_omega, 0, sizeof(double) * sim->d.n_users * sim->d.K, st));
        hipLaunchKernelGGL(k_export_omega_live, dim3(grid_for(static_cast<uint64_t>(sim->live_upper) * sim->d.K)),
                           dim3(kBlock), 0, st, sim->d, sim->t, d_omega);
    } else
        hipLaunchKernelGGL(k_export_omega, dim3(grid_for(static_cast<uint64_t>(sim->d.n_users) * sim->d.K)),
                           dim3(kBlock), 0, st, sim->d, d_omega);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_set_log_time(rg_sim* sim, double* d_time) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (d_time && !sim->d.log) return fail(RG_ESTATE, "attach a log buffer first (rg_sim_set_log)");
    sim->d.aux_time = d_time;
    return RG_OK;
}

int rg_sim_sort_log_time(rg_sim* sim, const int64_t* d_row_offsets, double* d_sorted_time, uint64_t sorted_capacity, void* stream) {
    if (!sim || !d_row_offsets || !d_sorted_time) return fail(RG_EINVAL, "NULL argument");
    if (!sim->d.log) return fail(RG_ESTATE, "no log buffer is attached");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const DevSim& d = sim->d;
    uint64_t n_rows = 0;
    HIP_TRY(hipMemcpyAsync(&n_rows, d.log_base + sim->t, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_rows > d.log_cap) return fail(RG_ELIMIT, "log overflow: %llu rows emitted, capacity %llu",
                                        (unsigned long long)n_rows, (unsigned long long)d.log_cap);
    hipLaunchKernelGGL(k_scatter_time, dim3(grid_for(n_rows)), dim3(kBlock), 0, st, d, n_rows, d_row_offsets, d_sorted_time, sorted_capacity);
    hipLaunchKernelGGL(k_scatter_time_phantom, dim3(grid_for(d.n_users)), dim3(kBlock), 0, st, d, d_row_offsets, d_sorted_time, sorted_capacity);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_export_time(rg_sim* sim, double* d_time, void* stream) {
    if (!sim || !d_time) return fail(RG_EINVAL, "NULL argument");
    hipLaunchKernelGGL(k_export_time, dim3(grid_for(sim->d.n_users)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), sim->d, sim->t, d_time);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_debug_set_uniforms(rg_sim* sim, const double* d_u) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    sim->d.u_override = d_u;
    return RG_OK;
}

int rg_sim_debug_set_omega(rg_sim* sim, const double* d_omega, void* stream) {
    if (!sim || !d_omega) return fail(RG_EINVAL, "NULL argument");
    if (!sim->users_reset || sim->t != 0) return fail(RG_ESTATE, "only right after rg_sim_reset_users");
    hipLaunchKernelGGL(k_debug_set_omega, dim3(grid_for(static_cast<uint64_t>(sim->d.n_users) * sim->d.K)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), sim->d, d_omega);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_debug_click_decisions(rg_sim* sim, const int32_t* d_actions, const double* d_u, uint8_t* d_out, void* stream) {
    if (!sim || !d_actions || !d_u || !d_out) return fail(RG_EINVAL, "NULL argument");
    if (!sim->users_reset || !sim->tables_set) return fail(RG_ESTATE, "needs tables and a reset range");
    if (!sim->d.beta32) return fail(RG_ESTATE, "the fp32 click decision exists where k_walk runs (sigma_omega == 0 with the per-user cache)");
    if (sim->d.K > 64) return fail(RG_EINVAL, "K > 64");
    hipLaunchKernelGGL(k_debug_click, dim3(grid_for(sim->d.n_users)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), sim->d,
                       d_actions, d_u, d_out);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_debug_set_history(rg_sim* sim, const uint32_t* d_nd, const uint32_t* d_products, const uint32_t* d_counts,
                             uint32_t stride, void* stream) {
    if (!sim || !d_nd || !d_products || !d_counts) return fail(RG_EINVAL, "NULL argument");
    if (!sim->users_reset || sim->t != 0) return fail(RG_ESTATE, "only right after rg_sim_reset_users");
    if (!sim->d.hist_cap) return fail(RG_ESTATE, "the policy keeps no view history");
    if (stride + 1 > sim->d.hist_cap) return fail(RG_EINVAL, "stride %u exceeds the history capacity %u", stride, sim->d.hist_cap - 1);
    hipLaunchKernelGGL(k_debug_set_history, dim3(grid_for(sim->d.n_users)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), sim->d,
                       d_nd, d_products, d_counts, stride);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_debug_ouc_acts(rg_sim* sim, const double* d_u1, int32_t* d_action, double* d_ps, uint8_t* d_flags, void* stream) {
    if (!sim || !d_u1 || !d_action || !d_ps || !d_flags) return fail(RG_EINVAL, "NULL argument");
    if (sim->d.policy != RG_POLICY_ORGANIC_USER_COUNT) return fail(RG_ESTATE, "policy is not OrganicUserEventCounter");
    if (!sim->users_reset) return fail(RG_ESTATE, "no reset range");
    hipLaunchKernelGGL(k_debug_ouc_acts, dim3(grid_for(sim->d.n_users)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), sim->d,
                       d_u1, d_action, d_ps, d_flags);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_debug_walk_fate(rg_sim* sim, uint8_t* d_flags, void* stream) {
    if (!sim || !d_flags) return fail(RG_EINVAL, "NULL argument");
    if (!sim->d.use_cache || !sim->walk) return fail(RG_ESTATE, "no walked run (sigma_omega == 0, rg_sim_run to the end)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_debug_fate_round2, dim3(grid_for(sim->d.n_users)), dim3(kBlock), 0, st, sim->d, d_flags);
    if (sim->fate_count)
        hipLaunchKernelGGL(k_debug_fate_last, dim3(grid_for(sim->d.n_users / 16 + 1)), dim3(kBlock), 0, st, sim->d, d_flags, sim->fate_base, sim->fate_count);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_debug_uncertified(rg_sim* sim, uint8_t* d_flags, void* stream) {
    if (!sim || !d_flags) return fail(RG_EINVAL, "NULL argument");
    if (sim->t == 0) return fail(RG_ESTATE, "no step has run");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(d_flags, 0, sim->d.n_users, st));
    hipLaunchKernelGGL(k_debug_uncertified, dim3(grid_for(sim->d.n_users)), dim3(kBlock), 0, st, sim->d, sim->t - 1, d_flags);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_sort_log(rg_sim* sim, int64_t* d_row_offsets, int64_t* d_scratch, rg_event* d_sorted,
                    uint64_t sorted_capacity, void* stream) {
    if (!sim || !d_row_offsets || !d_scratch || !d_sorted) return fail(RG_EINVAL, "NULL argument");
    if (!sim->d.log) return fail(RG_ESTATE, "no log buffer is attached");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const DevSim& d = sim->d;
    const uint32_t n = d.n_users;
    const uint32_t nb = (n + kBlock - 1) / kBlock;
    // d_row_offsets: n + 1 entries (last = total rows); d_scratch: n + nb entries
    int64_t* rows = d_scratch;
    int64_t* block_sums = d_scratch + n;
    hipLaunchKernelGGL(k_rows_per_user, dim3(grid_for(n)), dim3(kBlock), 0, st, d, rows);
    hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(kBlock), 0, st, rows, d_row_offsets, block_sums, n);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, st, block_sums, nb, d_row_offsets + n);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(kBlock), 0, st, d_row_offsets, block_sums, n);
    // rows written so far = log_base[t]; read it on the device side via the scatter bound
    uint64_t n_rows = 0;
    HIP_TRY(hipMemcpyAsync(&n_rows, d.log_base + sim->t, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_rows > d.log_cap) return fail(RG_ELIMIT, "log overflow: %llu rows emitted, capacity %llu",
                                        (unsigned long long)n_rows, (unsigned long long)d.log_cap);
    hipLaunchKernelGGL(k_scatter_rows, dim3(grid_for(n_rows)), dim3(kBlock), 0, st, d, n_rows, d_row_offsets,
                       d_sorted, sorted_capacity);
    hipLaunchKernelGGL(k_scatter_phantom, dim3(grid_for(n)), dim3(kBlock), 0, st, d, d_row_offsets, d_sorted,
                       sorted_capacity);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

int rg_sim_sort_log_aux(rg_sim* sim, const int64_t* d_row_offsets, double* d_sorted_ps, double* d_sorted_p_click,
                        uint64_t sorted_capacity, void* stream) {
    if (!sim || !d_row_offsets) return fail(RG_EINVAL, "NULL argument");
    if (!sim->d.log) return fail(RG_ESTATE, "no log buffer is attached");
    if (!d_sorted_ps && !d_sorted_p_click) return RG_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const DevSim& d = sim->d;
    uint64_t n_rows = 0;
    HIP_TRY(hipMemcpyAsync(&n_rows, d.log_base + sim->t, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_rows > d.log_cap) return fail(RG_ELIMIT, "log overflow: %llu rows emitted, capacity %llu",
                                        (unsigned long long)n_rows, (unsigned long long)d.log_cap);
    hipLaunchKernelGGL(k_scatter_aux, dim3(grid_for(n_rows)), dim3(kBlock), 0, st, d, n_rows, d_row_offsets,
                       d_sorted_ps, d_sorted_p_click, sorted_capacity);
    hipLaunchKernelGGL(k_scatter_aux_phantom, dim3(grid_for(d.n_users)), dim3(kBlock), 0, st, d, d_row_offsets,
                       d_sorted_ps, d_sorted_p_click, sorted_capacity);
    HIP_TRY(hipGetLastError());
    return RG_OK;
}

}  // extern "C"
#endif  // RG_HAS(1): C ABI
