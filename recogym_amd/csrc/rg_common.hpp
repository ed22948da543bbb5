// rg_common.hpp — librecogym_hip.so: the reco-gym-v1 step loop as batched CDNA4 (gfx950) kernels (shared part of its seven units).
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

// Translation units.  The library is built from seven units, one per kernel family — rg_host.hip (host code, the C ABI, small
// kernels), rg_exact.hip (float64 resolve), rg_draw_fp32.hip (fp32 / lean 16-bit sweeps), rg_draw_pipelined.hip (the pipelined sweep
// + per-user cache kernels), rg_draw_wide.hip (wide-K sweep), rg_advance.hip (advance / tail / frozen LogReg), rg_walk.hip (the
// user-major walk) — compiled in parallel and linked by __graft_entry__.build(); recogym_hip.hip includes all seven (a one-unit
// build).  This header holds what they share: types, the workspace layout, device helpers (namespace rgk, identical in every
// unit); a unit hands its kernels to the host code through the *_kernel_for functions declared here.
#pragma once

// RG_WALK_PRECISE_CHUNK = 1 (default): the walk behind k_sweep_xh recomputes a draw's chunk as a float64 dot (one budget delta for
// sweep and recomputed terms); -DRG_WALK_PRECISE_CHUNK=0: in plain fp32 with its own in-chunk budget delta_c (cert_correlated,
// hot_budgets) — measured a tie on C3 (walk -1.0 ms, 4 % more parked users: +0.4 ms of float64 batch; profiles/r5/ab_call27.jsonl)
#ifndef RG_WALK_PRECISE_CHUNK
#define RG_WALK_PRECISE_CHUNK 1
#endif

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

// What k_draw_tp leaves per organic draw of an unsliced step for k_pick (rg_draw_lds.hip), indexed by the draw's position in the
// step's organic list: the draw's uniform, the total and the prefix at the start of the draw's 128-product tile (fp32 roundings of
// the float64 running prefix), the reference and the certificate's delta, the tile (>= n_tiles: no tile found — float64).
struct TpRec { double u; float S, pb, q, dlt; uint32_t ti, pad; };
// ... followed in the same record by the user's omega32 (2 KH floats: what k_pick builds its B operands from) — a record is one
// 128-byte line at K <= 20 (64 bytes at K <= 8).  The draws of a step are counted and listed per (tile, shard): a device-scope
// atomic on ONE address sustains ~6 M/s on this chip (measured: 264 M draws over 79 counters ran at the atomics' rate,
// profiles/r6/ab_call3_tp.jsonl), so every tile has kTpShards counters, taken by work item modulo kTpShards.
constexpr uint32_t kTpShards = 16, kTpBins = 128, kTpLists = kTpBins * kTpShards;
// ... and MORE shards where the tiles are few (a table of P <= 1024 has <= 8 of them: with 16 shards its draws hit <= 128
// counters — BASELINE config 5 with the reference-fitted policies, P = 100, ran 4x slower on 16: profiles/r6/ab_call21_c5.jsonl):
// as many as keep tiles x shards <= kTpLists, at most 128
__host__ __device__ constexpr uint32_t tp_shards_of(uint32_t n_tiles) {
    uint32_t s = kTpShards;
    while (s < 128u && 2u * s * (n_tiles ? n_tiles : 1u) <= kTpLists) s *= 2u;
    return s;
}
__host__ __device__ constexpr uint32_t tp_rec_stride(uint32_t KH) { return (32u + 8u * KH + 63u) & ~63u; }   // (KH = 32: 320)                     // bytes
// draws of one (tile, shard) list at most: every work item (128 users of k_draw_tp, 256 of k_draw_tpw) may put all its draws into one
inline uint32_t tp_shard_cap(uint64_t n, uint32_t shards) { return static_cast<uint32_t>((((n + 255) / 256 + shards - 1) / shards) * 256); }

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
    uint32_t walk_helpers;       // k_walk2: events of a bandit run its owner's lane may hand to the wave's idle lanes in one bandit iteration (0 .. kWalkHelpersMax)
    uint32_t walk_click_join;    // k_walk2: the click batch is taken inside a bandit iteration (1) or in an iteration of its own (0)
    uint32_t walk_line64;        // k_walk2 (host side: which instantiation): round 3's 64-bit history line of 15 products (RECOGYM_WALK_HIST=1)
    uint32_t exact_base;      // first exact_list entry of the batch being resolved
    uint32_t exact_last;      // this is the last batch launched for the step
    float2* sc_scratch;       // [kMaxGrid*4 waves][kMaxSC][32] {sum, reference} of the MFMA draw kernel
    float* chunk_scratch;     // [kMaxGrid*4 waves][n_chunks][32] exp-sum of every 32-product chunk
    char* tp_rec;             // [n_cap][tp_rec_stride] k_draw_tp -> k_pick: {TpRec, omega32} (null: the configuration has no k_draw_tp)
    uint32_t* tp_hist;        // [kTpBins x kTpShards + 4] draws of the step per (128-product tile, shard) (k_draw_tp counts, k_pick's last
                              // block clears), then the blocks of k_pick that are done
    uint32_t* tp_order;       // [n_chunks / 4][kTpShards][tp_cap] list positions of the step's draws, by tile and shard
    uint32_t tp_cap, tp_shards;   // capacity of one (tile, shard) list; shards per tile (tp_shards_of)
    uint32_t tp_cpt;          // 32-product chunks per list tile: 4 (k_draw_tp's 128-product tiles); the wide sweep's super-tiles: 2 x its tiles per stored prefix
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
    // k_sweep_xh (rg_draw_exacthi.hip): the prefix-form sweep of sigma_omega = 0 whose leading accumulator is error-free
    uint32_t XNH, XNL, XRS;   // MFMA k-steps of the exact / the residual group, row stride of xsplit in bytes (0: no such kernel)
    unsigned short* xsplit;   // [P_pad][XRS/2] fixed-point fp16 pieces of Gamma log2 e and mu log2 e (layout: rg_draw_exacthi.hip)
    float* xmulo;             // [P_pad] seed of the residual accumulator: 2^9 (mu' - its two fixed-point pieces), -inf beyond P
    float* xstats;            // [2 KH + 1] max_p |Gamma'_pk - Ghi - Glo| per column, then max |Glo|
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
    // Run-ahead rounds (rg_sim_run to the end without the user-major walk; k_advance_run): a lock-step "step" becomes a ROUND — every
    // listed user's next organic event, or its whole bandit run up to the next organic event / stop / possible click — so users sit
    // at different event indices: ev[user index] = index of the user's next event (the `t` of its draws and rows), and the round's
    // number only indexes the lists and per-step counters.  run_ahead = events a round may take a user through (0 = lock-step).
    uint32_t run_ahead;
    uint32_t* ev;             // [n_users] by user index
    unsigned long long* run_ctl;         // [4] word 0: next free raw-log row (bandit rows of a round are reserved block by block)
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
    const uint8_t* lr_coef8_t; const float* lr_scale8;   // 8-bit copy (q + 128) and its per-row scale: the screening pass reads it instead, or null
    uint32_t* lr_action;      // [n_cap] by user index: action of the user's current history
    // select_randomly (rg_config.lr_select_randomly): the act is SAMPLED per event from softmax(scores) — k_logreg_sample leaves the
    // action and its probability for the step's bandit event (or, for an organic user that stops at this step, for its trailing
    // row) in lr_action / lr_ps, and for a bandit user whose drawn transition is `stop` the trailing row's own draw in lr_action2 / lr_ps2
    uint32_t lr_sample;
    double* lr_ps; uint32_t* lr_action2; double* lr_ps2;
    uint8_t* lr_dirty;        // [n_cap] by user index
    uint32_t* lr_list;        // [n_cap] slots whose act is to be computed this step
    uint32_t* lr_cnt;         // [kMaxSteps + 2]
    uint32_t* lr_part;        // [lr_part_cap][kLrSplit][kLrPartWords]: the screen's result per listed act and class range
    uint32_t lr_part_cap;     // acts of a step the screen / decide pair takes (rows of lr_part: n / 2 + 4096, at most n); the acts a
                              // step lists beyond that go through k_logreg_acts (no scratch)
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
    uint64_t debug_row_base;  // rg_sim_debug_set_row_base: first raw-log row of a reset range (0 outside that test)
    // reco-gym-v0 (env_kind = 1, reco_env_v0.py): the user's state is its current product view, every draw a table look-up
    uint32_t env_kind, e0_cluster;
    const double* e0_cdf_init; const double* e0_cdf_cluster; const double* e0_click_p; const double* e0_click_qn; const double* e0_click_px1;
    uint32_t* pv0;            // [n_cap] by user index: the product currently viewed (reco_env_v0.py:52-54,65-67)
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
    uint32_t run_ahead;       // option: events per user and round of a run to the end (k_advance_run); 0 = lock-step
    uint32_t lr_part_rows;    // rows the workspace holds for the LogReg screen (the option lr_part_cap can only go below it)
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
    void (*xh_kernel)(DevSim, uint32_t, uint32_t);   // k_sweep_xh: the walked run's sweep where it exists (else bf16_kernel)
    size_t xh_smem;
    int xh_waves;            // waves per block of that kernel (4: two blocks per CU, 8: one; RECOGYM_XH_WAVES)
    // k_draw_tp (rg_draw_lds.hip): the unsliced sweep of a step whose draws cannot be cached (sigma_omega > 0) — tile prefixes in
    // LDS, the search on them; nullptr where no instance serves the configuration or its LDS does not fit (then bf16_kernel)
    void (*tp_kernel)(DevSim, uint32_t, uint32_t);
    void (*pick_kernel)(DevSim, uint32_t, uint32_t);
    size_t tp_smem;
    uint32_t tp_nts;         // its LDS row stride (floats) of a user's tile prefixes
    uint32_t sweep_lds;      // option: 1 = use it (default), 0 = k_draw_bf16p everywhere (A/B tests)
    bool handover_auto;      // walk_handover follows the reset range (16 below 2 M users, else 32) until it is set explicitly
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
typedef void (*advance_run_kernel_t)(DevSim, uint32_t, uint32_t);
typedef void (*round_rows_kernel_t)(DevSim, uint32_t, uint32_t);
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
draw_kernel_t tp_kernel_for(const DevSim& d);              // part 9 (nullptr: k_draw_bf16p serves the configuration)
draw_kernel_t tpw_kernel_for(const DevSim& d);             // k_draw_tpw: the same for the wide-K classes (64 users per wave, a prefix per super-tile)
draw_kernel_t pick_kernel_for(const DevSim& d);            // k_pick: its second half (the draws grouped by tile, one tile on the matrix cores)
draw_kernel_t xh_kernel_for(const DevSim& d, int waves);   // part 8 (nullptr: no error-free sweep for this K class); waves per block: 4 or 8
void (*xh_table_kernel())(DevSim);
void (*xh_stats_kernel())(DevSim);
search_kernel_t drift_kernel();                            // part 6
search_kernel_t logreg_select_kernel();
search_kernel_t logreg_acts_kernel();
search_kernel_t logreg_screen_kernel(bool q8);
search_kernel_t logreg_decide_kernel();
search_kernel_t logreg_sample_kernel();
advance_kernel_t advance_kernel();
advance_run_kernel_t advance_run_kernel();
round_rows_kernel_t round_rows_kernel();
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

struct Geom { uint32_t KH, KS, TP, P_pad, n_chunks, sc_chunks, n_sc, N1, N2, N3, RS, TPB, F16, XNH, XNL, XRS; };

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
    // k_sweep_xh classes (exact group: K + 3 slots of 16 NH; residual group: 4 K slots of 16 NL); RECOGYM_XH=0: A/B tests
    if (g.F16 == 1 && (g.KH == 4 || g.KH == 10)) {
        const char* e_x = getenv("RECOGYM_XH");
        if (!(e_x && e_x[0] == '0')) {
            g.XNH = g.KH == 4 ? 1 : 2; g.XNL = g.KH == 4 ? 2 : 5;
            if (c.K + 3 > 16 * g.XNH || 4 * c.K > 16 * g.XNL) g.XNH = g.XNL = 0;
            g.XRS = g.XNH ? 32 * (g.XNH + g.XNL) + 16 : 0;
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

// k_draw_bf16p's product-tile ring in LDS and the blocks per CU it is compiled for.  Default since round 4: the tile in use + ONE in
// flight (48 KB of LDS at K = 20), two blocks per CU (253 registers) — 1 % (C3) to 3 % (C3 with drift) faster than the tile in use
// + two in flight (-DRG_SWEEP_NB=3, 67 KB).  -DRG_SWEEP_OCC=3 (three blocks per CU fit the 48 KB: 168 registers, the steady-state
// loop keeps ~8 scratch accesses per iteration) is 30 - 70 % SLOWER: profiles/r4/ab_call26_*, DESIGN.md §10.4.
#ifndef RG_SWEEP_NB
#define RG_SWEEP_NB 2
#endif
#ifndef RG_SWEEP_OCC
#define RG_SWEEP_OCC 2
#endif
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
    const bool xh = g.XNH != 0 && cache_wanted(c, g);
    unsigned short* xsplit = w.take<unsigned short>(xh ? static_cast<size_t>(g.P_pad) * (g.XRS / 2) : 1);
    float* xmulo = w.take<float>(xh ? g.P_pad : 1);
    float* xstats = w.take<float>(2 * g.KH + 2);
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
    const bool lrs = lr && c.lr_select_randomly;
    double* lr_ps = w.take<double>(lrs ? n : 1);
    uint32_t* lr_action2 = w.take<uint32_t>(lrs ? n : 1);
    double* lr_ps2 = w.take<double>(lrs ? n : 1);
    // the screen's scratch: a row per act of a STEP, not per user — a step lists the users whose history changed and who act now
    // (a quarter of the organic users at the default transition matrix); what a step lists beyond the rows goes through k_logreg_acts
    const size_t lr_part_cap = lr ? (n / 2 + 4096 < n ? n / 2 + 4096 : n) : 0;
    uint32_t* lr_part = w.take<uint32_t>(lr ? lr_part_cap * static_cast<size_t>(8 * (4 + 2 * 8)) : 1);       // kLrSplit x kLrPartWords
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
    uint32_t* ev = w.take<uint32_t>(n);
    uint32_t* pv0 = w.take<uint32_t>(c.env_kind ? n : 1);
    unsigned long long* run_ctl = w.take<unsigned long long>(4);
    // k_draw_tp's / k_draw_tpw's classes (tp_kernel_for), every draw a sweep; k_draw_tp: a user's <= 128 tile prefixes in LDS
    const bool tp = ((g.F16 == 1 && g.KH <= 10 && g.n_chunks / 4 <= kTpBins) || g.F16 == 2) && !cache;
    char* tp_rec = w.take<char>(tp ? static_cast<size_t>(n) * tp_rec_stride(g.KH) : 1);
    uint32_t* tp_hist = w.take<uint32_t>(kTpBins * kTpShards + 4);
    // (lists: tiles x shards; the wide sweep's super-tiles are fixed at create: sized for the 16-shard case, which is the largest product)
    const uint32_t tp_tiles = g.F16 == 2 ? kTpBins : g.n_chunks / 4;
    const uint32_t tp_sh = g.F16 == 2 ? kTpShards : tp_shards_of(tp_tiles);
    uint32_t* tp_order = w.take<uint32_t>(tp ? static_cast<size_t>(tp_tiles) * tp_sh * tp_shard_cap(n, tp_sh) : 1);
    if (d) {
        d->tp_rec = tp ? tp_rec : nullptr; d->tp_hist = tp_hist; d->tp_order = tp_order; d->tp_cap = tp_shard_cap(n, tp_sh); d->tp_shards = tp_sh; d->tp_cpt = 4;
        d->ev = ev; d->run_ctl = run_ctl; d->run_ahead = 0; d->pv0 = pv0;
        d->phantom_ps = phantom_ps; d->utime = utime; d->phantom_time = phantom_time;
        d->drift_list = drift_list; d->drift_sig = drift_sig; d->drift_cnt = drift_cnt;
        d->use_cache = cache ? 1u : 0u; d->cache_rec = cache_rec; d->cache_chunk = cache_chunk; d->cache_resc = cache_resc;
        d->beta32 = cache ? beta32 : nullptr; d->KB4 = static_cast<uint32_t>(KB4);
        d->f64_valid = f64_valid; d->exact_cnt_b = exact_cnt_b; d->cache_row = cache_row; d->cache_row_f = cache_row_f;
        d->park_list = park_list; d->park_t = park_t; d->sweep_only = 0;
        d->walk_ctl = walk_ctl; d->step1_buf = step1_buf;
        d->walk_hot = cache ? walk_hot : nullptr; d->walk_scp = walk_scp;
        d->lr_ps = lr_ps; d->lr_action2 = lr_action2; d->lr_ps2 = lr_ps2; d->lr_sample = lrs ? 1u : 0u;
        d->lr_action = lr_action; d->lr_dirty = lr ? lr_dirty : nullptr; d->lr_list = lr_list; d->lr_cnt = lr_cnt; d->lr_part = lr_part; d->lr_part_cap = static_cast<uint32_t>(lr_part_cap);
        d->uid = uid; d->omega_alt = omega_alt; d->hist_alt = hist_alt;
        d->lpv_alt = lpv_alt; d->uid_alt = uid_alt;
        d->gamma32 = gamma32; d->mu32 = mu32; d->gamma32t = gamma32t; d->has_g32t = gamma32t_wanted(c, g) ? 1u : 0u; d->stats = stats; d->omega = omega; d->list = list;
        d->gamma_rm = gamma_rm; d->XKB = xkb;
        d->exact_rows = static_cast<uint32_t>(exact_rows); d->exact_base = 0;
        d->gammaT = gammaT; d->PT = static_cast<uint32_t>(PT); d->exact_ref = exact_ref; d->exact_sums = exact_sums; d->sc_scratch = sc_scratch; d->chunk_scratch = chunk_scratch;
        d->xsplit = xsplit; d->xmulo = xmulo; d->xstats = xstats;
        d->XNH = xh ? g.XNH : 0; d->XNL = xh ? g.XNL : 0; d->XRS = xh ? g.XRS : 0;
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
    if (c->env_kind > 1) return fail(RG_EINVAL, "unknown env_kind %u", c->env_kind);
    if (c->lr_select_randomly && (c->policy != RG_POLICY_LOGREG_FROZEN || c->num_products > 1024))
        return fail(RG_EINVAL, "lr_select_randomly needs RG_POLICY_LOGREG_FROZEN and at most 1024 products (a class per product)");
    if (c->env_kind == 1 && (c->time_mode || c->policy == RG_POLICY_LOGREG_FROZEN || c->policy == RG_POLICY_LAST_VIEW_TABLE))
        return fail(RG_EINVAL, "env_kind 1 (reco-gym-v0) runs the default clock and the uniform / random / organic-count / external policies");
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

// a 64-bit value of the wave's first active lane, as a scalar (the builtin returns a SIGNED int: OR-ing its low word into a 64-bit
// value without the cast sign-extends it — a raw-log row base beyond 2^31 rows, a mask with bit 31 set)
__device__ __forceinline__ unsigned long long readfirstlane_u64(unsigned long long x) {
#ifdef RG_TEST_SIGNED_RFL
    // (A/B build of tests/test_hip_parity.py::test_raw_log_rows_beyond_2_31_are_kept only: round 5's bug put back, to show that the
    // test sees it — the low word sign-extended into the high one)
    return (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(x >> 32))) << 32) |
           static_cast<unsigned long long>(static_cast<long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(x))));
#endif
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x)));
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x >> 32)));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
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
    if (d.run_ahead) t = d.ev[uidx];          // run-ahead rounds: the user's own event index
    const rg_u32x4 rw = rg_draw(d.seed, user, t, 0, RG_DRAW_EVENT);
    return rg_uniform(rw.w[0], rw.w[1]);
}

// searchsorted(cdf, u, side='right') of numpy's legacy choice: the number of entries <= u (clamped to the last index, as an
// index into [0, n) must be: cdf[n - 1] is exactly 1 > u)
__device__ __forceinline__ uint32_t upper_bound_f64(const double* cdf, uint32_t n, double u) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo < n ? lo : n - 1;
}

// reco-gym-v0: draw_click (reco_env_v0.py:61-63) = RandomState.binomial(1, p) — numpy's legacy inversion for n = 1: with
// qn = exp(log(1 - p)) the draw is [U > qn] unless U - qn > (p qn) / q, which restarts the inversion with a fresh uniform
// (slot j >= 1 of the event's draw); p > 0.5 draws 1 - inversion(1 - p).  `row` = action * P + view.
__device__ __forceinline__ bool env0_click(const DevSim& d, size_t row, uint32_t user, uint32_t t, double u_first) {
    const double p = d.e0_click_p[row], qn = d.e0_click_qn[row], px1 = d.e0_click_px1[row];
    double U = u_first;
    uint32_t X = 0;
    for (uint32_t j = 1;; ++j) {
        if (!(U > qn)) { X = 0; break; }
        if (!(U - qn > px1)) { X = 1; break; }
        if (j >= 64u) { X = 1; break; }                    // (never: each restart has probability ~1e-16)
        const rg_u32x4 w = rg_draw(d.seed, user, t, j, RG_DRAW_EVENT);
        U = rg_uniform(w.w[0], w.w[1]);
    }
    return p <= 0.5 ? X == 1 : X == 0;
}

__device__ __forceinline__ uint32_t* list_ptr(const DevSim& d, uint32_t parity, uint32_t state) {
    return d.list + (static_cast<size_t>(parity) * 2 + state) * d.n_cap;
}



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
        e.u = user; e.t = d.run_ahead ? d.ev[d.uid[slot]] : t; e.code = v; e.ps = __builtin_nanf("");
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


typedef const __attribute__((address_space(4))) double kdouble;   // constant address space: uniform loads become s_load


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
// rho_rel: the stored prefixes' own roundings, relative to S — 2^-20 (kRhoLoose) where they went through up to five fp32 adds
// each, 2^-23 (kRhoTight) where each is ONE rounding of a float64 sum (k_sweep_xh); the hot row carries it per user.
constexpr float kRhoLoose = 9.5463e-7f;    // 2^-20 x 1.001 (.001: second-order terms and the float64 roundings of the test)
constexpr float kRhoTight = 1.1933e-7f;    // 2^-23 x 1.001
// delta_c (< 0: = delta): the budget of the RECOMPUTED in-chunk prefixes a, b where it differs from the sweep terms' — the walk
// behind k_sweep_xh recomputes a chunk in plain fp32 (delta_c ~ 7e-5) against sweep terms good to ~1e-5: a and b are a chunk's
// worth of mass, A and T the whole table's, so the looser in-chunk budget widens the band by ~3 % (tools: DESIGN.md §2 "Round 5").
__device__ __forceinline__ CertLin cert_correlated(double S, double A, double a, double b, double delta, double rho_rel = static_cast<double>(kRhoLoose),
                                                   double delta_c = -1.0) {
    const double dp = delta * (1.0 + 2.0 * delta);         // >= delta / (1 - delta) for delta <= 1/2
    const double dc = delta_c < 0.0 ? dp : delta_c * (1.0 + 2.0 * delta_c);
    const double rho = rho_rel * S;
    const double T = S - A;                                // exact: both are fp32 values
    CertLin c;
    c.valid = T >= 0.0 && delta < 0.25 && delta_c < 0.25;
    c.num_lo = A * (1.0 + dp) + a * (1.0 + dc) + rho;
    c.den_lo = T * (1.0 - dp) + A * (1.0 + dp);
    c.num_hi = A * (1.0 - dp) + b * (1.0 - dc) - rho;
    c.den_hi = T * (1.0 + dp) + A * (1.0 - dp);
    return c;
}
// Float 31 of a user's hot row: rho_rel (2^-20 / 2^-23: the in-chunk budget is delta itself), or — a value >= kHotDcMin — the
// in-chunk budget delta_c of a user k_sweep_xh finalised (its stored prefixes are single roundings: rho_rel = 2^-23)
constexpr float kHotDcMin = 4.0e-6f;
__device__ __forceinline__ void hot_budgets(float r31, double delta, double* rho_rel, double* delta_c) {
    const bool has_dc = r31 >= kHotDcMin;
    *rho_rel = has_dc ? static_cast<double>(kRhoTight) : static_cast<double>(r31);
    *delta_c = has_dc ? static_cast<double>(r31) : delta;
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

// ---- certificate budget of k_sweep_xh (rg_draw_exacthi.hip; derivation: DESIGN.md §2 "round 5") ----
// v_exp_f32 (<= 2 ulp = 2.4e-7), the chunk's summation tree / the recomputed in-chunk prefix (<= 12 fp32 adds of positive
// terms: 7.2e-7): 9.6e-7, + 25 % margin
constexpr double kDeltaFixedXh = 1.2e-6;
// The exact accumulator holds multiples of 2^-16 below 2^24 x 2^-16 = 256: sum of |terms| = sum_k |Ghi whi| + |m1 + m2| + |q|
// <= Ahat log2 e (the joint bound dominates every sum of absolute values) + |q| + what the two fixed-point roundings add (< 1).
__device__ __forceinline__ bool xh_eligible(double Ahat, double qabs) {
    return Ahat * 1.4426950408889634 * 1.001 + qabs + 2.0 < 255.0;
}
// delta of a user swept by k_sweep_xh<.., NL>: Ahat (natural units), absw = sum |omega_k|, egam = sum_k |omega_k| x
// (column k's representation error of Gamma'), lob = bound on the sum of the residual accumulator's |terms| (log2 units,
// unscaled), qabs = the largest |reference| used.  What the SWEEP's terms can be off by (the walk's recomputed in-chunk terms
// have their own budget: xh_delta_chunk).
template <int NL>
__device__ __forceinline__ double xh_delta(const DevSim& d, double Ahat, double absw, double egam, double lob, double qabs) {
    constexpr double e24 = 5.9604644775390625e-08, ln2 = 0.6931471805599453, log2e = 1.4426950408889634;
    const double e_lo = (16.0 * NL + 4.0) * e24 * (lob + 1.6e-5);        // every add of the residual chain rounds at its own size
    const double e_x = e24 * (Ahat * log2e + qabs) * 1.01;               // the join H + 2^-9 L: one rounding of the exp2 argument
    // what the pieces leave out, per coordinate: Glo wlo <= 2^-9 2^-21, the omega tail <= |Ghi| 2^-33 <= 2^-30, the second scaled
    // copies of Glo / wmid (a bit below fp16's normal range at most): 4e-9 covers 2^-29
    const double e_drop = static_cast<double>(d.K) * 4.0e-9;
    if (!xh_eligible(Ahat, qabs))      // the leading sum may round: the two-way split kernel's accumulation budget ON TOP of this kernel's own
        // representation / residual / join terms (the sums are k_sweep_xh's, whatever the budget that certifies them)
        return static_cast<double>(d.K + 5) * e24 * Ahat + kDeltaFixedBf16 + f16_extra_delta(d, static_cast<float>(Ahat), static_cast<float>(absw)) +
               ln2 * (egam + e_drop + e_lo + e_x);
    (void)absw;
    const double sweep = ln2 * (egam + e_drop + e_lo + e_x) + kDeltaFixedXh;
#if RG_WALK_PRECISE_CHUNK
    // (the walk's recomputed terms share delta: a float64 dot of the fp32 tables, exp2 of the fp32-rounded argument)
    const double rec = e24 * (3.0 * Ahat + ln2 * qabs) * 1.01 + kDeltaFixedXh;
    return sweep > rec ? sweep : rec;
#else
    return sweep;
#endif
}
// The in-chunk budget of those users: the walk recomputes the draw's chunk in plain fp32 from the fp32 tables (K fused
// multiply-adds on partial sums <= Ahat, the referenced exp2 argument, exp2, <= 12 adds) — the accumulation budget of the
// fp32 / split kernels, which covers it with room to spare ((K + 5) 2^-24 Ahat + 1e-5 + the split's own terms)
__device__ __forceinline__ double xh_delta_chunk(const DevSim& d, double Ahat, double absw) {
    return static_cast<double>(d.K + 5) * 5.9604644775390625e-08 * Ahat + kDeltaFixedBf16 + f16_extra_delta(d, static_cast<float>(Ahat), static_cast<float>(absw));
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
#ifdef RG_WALK_TIMING
// -DRG_WALK_TIMING: k_walk2's iterations by kind — [4 k + 0] iterations, [4 k + 1] lanes with an event of the kind, [4 k + 2] wave
// cycles (s_memtime) spent in them; k = 0 memo-answered organic, 1 search, 2 bandit, 3 click batch; [16] helpers' events that
// counted, [17] helpers dealt (tools/walk_kinds.py)
static __device__ unsigned long long g_walk_stat[20];
#endif
#ifdef RG_F16W_TIMING
// -DRG_F16W_TIMING: s_memtime per section of the tile loop, summed over wave 0 of every block (tools/wide_probe.py)
static __device__ unsigned long long g_f16w_t[8];
#define RG_TSEC(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define RG_TSEC(i) do {} while (0)
#endif


// user groups per wave of the wide kernel (RECOGYM_F16W_UG: 1 = 8 waves x 32 users, 2 = 4 waves x 64 users)
inline int f16w_ug() {
    const char* e = getenv("RECOGYM_F16W_UG");
    return (e && e[0] == '2') ? 2 : 1;
}

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

// ------------------------------------------------------------------------------------------
// k_advance — one Markov transition for every live user (lane per user).
// ------------------------------------------------------------------------------------------
constexpr int kAdvBlock = 256;


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
constexpr uint32_t kWalkHelpersMax = 7;  // k_walk2: events of a bandit run that idle lanes may take in one iteration (DevSim::walk_helpers)
constexpr int kWClick = 7;              // lane state: bandit event whose click needs ctr (uniform >= kNoClickBelow): taken in batches
__host__ __device__ inline size_t walk2_wave_lds(int hist) { return (hist ? 16 * 64 * 8 : 0) + 64 * 12 + 64; }   // history lines, the search's mailbox (the helpers' tables), the search's lane list
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




















}  // namespace rgk
