// rg_advance.hip — librecogym_hip.so, unit 6 of 7: the lock-step transition (k_advance, k_drift), the tail kernel and the frozen LogReg act kernels.
// (see rg_common.hpp for the shared types and helpers, DESIGN.md for the data layout and the rooflines)

#include "rg_common.hpp"

namespace rgk {

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
            need = d.lr_dirty[uidx] != 0 || d.lr_sample != 0;        // (sampled acts: a fresh draw for every event)
            if (d.lr_sample && is_org) slot |= 0x80000000u;          // ... and the kernel must know which event the act is for
            if (need && is_org) {
                // an organic user needs an act only for its phantom row: when this step's transition stops it — and, in run-ahead
                // rounds, when the transition starts a bandit run (k_advance_run takes the user through it in this round)
                const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
                const rg_u32x4 w = rg_draw(d.seed, user, d.run_ahead ? d.ev[uidx] : t, 0, RG_DRAW_EVENT);
                const double u_trans = rg_uniform(w.w[2], w.w[3]);
                const int ns = (d.cdf_o0 <= u_trans) + (d.cdf_o1 <= u_trans);
                need = (ns == RG_STATE_STOP || (d.run_ahead && ns == RG_STATE_BANDIT)) && !((d.first_user + uidx) < d.organic_only_below);
            }
        }
        const unsigned long long m = __ballot(need);
        uint32_t base = 0;
        if (m && lane_id() == 0) base = atomicAdd(&d.lr_cnt[t], static_cast<uint32_t>(__popcll(m)));
        base = __shfl(static_cast<int>(base), 0);
        if (need) d.lr_list[base + prefix_in_mask(m)] = slot;
    }
}

// RG_POLICY_LOGREG_FROZEN with select_randomly (logreg_ips.py:61-72): a wave per listed user — the class scores in float64 (scipy's
// CSR x dense order: viewed products ascending, multiply then add, intercept last), softmax (exp(s - max) / sum), and
// rng.choice(P, p = proba) = the first class whose normalised cumulative probability exceeds the event's second policy uniform; ps =
// proba[action].  The sums here run in wave order (the reference: numpy's pairwise sum and sequential cumsum): 1e-16-level
// deviations, as everywhere the float64 path follows numpy.  A bandit user whose drawn transition is `stop` gets its trailing
// row's draw too (same probabilities, the uniform of event t + 1); an organic user listed because it stops: that draw only.
__global__ void __launch_bounds__(kBlock) k_logreg_sample(DevSim d, uint32_t t) {
    __shared__ double s_p[kBlock / 64][1024];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    double* pr = s_p[wv];
    const uint32_t n = d.lr_cnt[t], C = d.lr_n;
    const uint32_t waves_total = gridDim.x * (kBlock / 64);
    unsigned long long c_acts = 0, c_rows = 0;
    for (uint32_t w = blockIdx.x * (kBlock / 64) + wv; w < n; w += waves_total) {
        const uint32_t entry = d.lr_list[w];
        const bool org = (entry >> 31) != 0;
        const uint32_t slot = entry & 0x7FFFFFFFu;
        const uint32_t uidx = d.uid[slot];
        const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
        const hent_t* hr = hist_row(d, slot) + 1;
        const uint32_t nd = h_cnt(hr[-1]);
        c_acts += 1; c_rows += nd;
        double mx = -INFINITY;
        for (uint32_t c0 = 0; c0 < C; c0 += 64) {
            const uint32_t c = c0 + lane;
            double sc = 0.0;
            if (c < C) {
                for (uint32_t i = 0; i < nd; ++i) {
                    const hent_t x = hr[i];
                    sc = __dadd_rn(sc, __dmul_rn(static_cast<double>(h_cnt(x)), d.lr_coef_t[static_cast<size_t>(h_prod(x)) * C + c]));
                }
                sc = __dadd_rn(sc, d.lr_intercept[c]);
                pr[c] = sc;
                mx = fmax(mx, sc);
            }
        }
        mx = wave_max(mx);
        double sum = 0.0;
        for (uint32_t c0 = 0; c0 < C; c0 += 64) {
            const uint32_t c = c0 + lane;
            if (c < C) { const double e = exp(pr[c] - mx); pr[c] = e; sum += e; }
        }
        sum = wave_sum(sum);
        double tot = 0.0;
        for (uint32_t c0 = 0; c0 < C; c0 += 64) {
            const uint32_t c = c0 + lane;
            if (c < C) { const double p = pr[c] / sum; pr[c] = p; tot += p; }
        }
        tot = wave_sum(tot);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        auto pick = [&](double u, uint32_t* a_out, double* ps_out) {
            double run = 0.0;
            uint32_t a = C - 1;
            for (uint32_t c0 = 0; c0 < C; c0 += 64) {
                const uint32_t c = c0 + lane;
                const double p = c < C ? pr[c] : 0.0;
                const double incl = wave_scan(p, lane) + run;
                const unsigned long long hit = __ballot(c < C && incl / tot > u);
                if (hit) { a = c0 + static_cast<uint32_t>(__builtin_ctzll(hit)); break; }
                run = __shfl(incl, 63);
            }
            *a_out = a; *ps_out = pr[a];
        };
        const uint32_t t_act = org ? t + 1 : t;
        const rg_u32x4 wp = rg_draw(d.policy_seed, user, t_act, 0, RG_DRAW_POLICY);
        uint32_t a; double ps;
        pick(rg_uniform(wp.w[2], wp.w[3]), &a, &ps);
        if (lane == 0) { d.lr_action[uidx] = a; d.lr_ps[uidx] = ps; d.lr_dirty[uidx] = 0; }
        if (!org) {
            const rg_u32x4 we = rg_draw(d.seed, user, t, 0, RG_DRAW_EVENT);
            const double u_trans = rg_uniform(we.w[2], we.w[3]);
            if ((d.cdf_b0 <= u_trans) + (d.cdf_b1 <= u_trans) == RG_STATE_STOP) {
                const rg_u32x4 w2 = rg_draw(d.policy_seed, user, t + 1, 0, RG_DRAW_POLICY);
                uint32_t a2; double ps2;
                pick(rg_uniform(w2.w[2], w2.w[3]), &a2, &ps2);
                if (lane == 0) { d.lr_action2[uidx] = a2; d.lr_ps2[uidx] = ps2; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && c_acts) {
        atomicAdd(&d.counters[RG_CNT_LR_ACTS], c_acts);
        atomicAdd(&d.counters[RG_CNT_LR_ROWS], c_rows);
    }
}

__global__ void __launch_bounds__(kBlock) k_logreg_acts(DevSim d, uint32_t t) {
    const int lane = lane_id();
    const uint32_t n = d.lr_cnt[t];
    const uint32_t waves_total = gridDim.x * (kBlock / 64);
    unsigned long long c_acts = 0, c_rows = 0, c_exact = 0;
    // beside the fp16 screen / decide pair this kernel takes what the step lists beyond the screen's scratch rows
    const uint32_t w0 = d.lr_coef16_t ? d.lr_part_cap : 0u;
    for (uint32_t w = w0 + blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w < n; w += waves_total) {
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

// (blocks per CU = waves per SIMD of k_logreg_screen: acts of C5 169 ms at 4, 176 at 5, 233 at 6 where it spills: ab_call29_c5.jsonl)
#ifndef RG_LR_OCC
#define RG_LR_OCC 4
#endif
#ifndef RG_LR_RANGE_MAJOR
#define RG_LR_RANGE_MAJOR 0
#endif
// maximum over the 64 lanes (all active), in every lane: DPP inside the rows of 16, v_readlane across them — no LDS round trip
// (six dependent ds_bpermute per maximum were ~0.3 us of latency per batch of the screen)
__device__ __forceinline__ float wave_max_dpp(float x) {
    auto dpp = [](float v, auto ctrl) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    x = fmaxf(x, dpp(x, std::integral_constant<int, 0xB1>{}));      // quad_perm:[1,0,3,2]
    x = fmaxf(x, dpp(x, std::integral_constant<int, 0x4E>{}));      // quad_perm:[2,3,0,1]
    x = fmaxf(x, dpp(x, std::integral_constant<int, 0x141>{}));     // row_half_mirror
    x = fmaxf(x, dpp(x, std::integral_constant<int, 0x140>{}));     // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// acc + a * (fp16 half of w): v_fma_mix_f32 takes the fp16 operand as it lies in the row (one instruction and no converted copy;
// left to itself the compiler converts all 64 halves of a batch first — 64 more live registers — for packed fp32 fmas)
__device__ __forceinline__ float fma_mix_lo(float a, uint32_t w, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(acc));
    return r;
}
__device__ __forceinline__ float fma_mix_hi(float a, uint32_t w, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(acc));
    return r;
}
template <bool Q8>
__global__ void __launch_bounds__(kBlock, RG_LR_OCC) k_logreg_screen(DevSim d, uint32_t t) {
    constexpr uint32_t kCandCap = 64;          // candidates of a range while it streams (a lane each in the second level)
    __shared__ uint32_t s_cand[kBlock / 64][kCandCap];
    __shared__ float s_cval[kBlock / 64][kCandCap];
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const uint32_t n = min(d.lr_cnt[t], d.lr_part_cap);         // (the rest: k_logreg_acts)
    const uint32_t C = d.lr_n;
    const uint32_t RC = ((C + kLrSplit - 1) / kLrSplit + 7u) & ~7u;       // classes per range (a multiple of 8)
    const uint32_t waves_total = gridDim.x * (kBlock / 64);
    for (uint32_t item = blockIdx.x * (kBlock / 64) + wave; item < n * kLrSplit; item += waves_total) {
#if RG_LR_RANGE_MAJOR
        const uint32_t w = item % n, r = item / n;      // range-major: the waves in flight read ONE eighth of the table's columns
#else
        const uint32_t w = item / kLrSplit, r = item % kLrSplit;
#endif
        // (everything about the act is wave-uniform: kept in scalar registers, the history's loads scalar or broadcast)
        const uint32_t slot = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(d.lr_list[w])));
        const hent_t* hr = hist_row(d, slot) + 1;
        // the header and the first eight entries leave TOGETHER (one 128-byte line; entries beyond the count are whatever the row
        // held before: replaced below, never used as an address) — the screen is a chain of dependent round trips, list -> history
        // -> rows -> rows ..., and each one fewer is worth ~8 % of it
        hent_t h8[8];
        const hent_t hdr = hr[-1];
#pragma unroll
        for (int e = 0; e < 8; ++e) h8[e] = hr[e];
        const uint32_t nd = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(h_cnt(hdr))));   // (scalar loop control)
        {
            hent_t prev = 0;       // (product 0, count 0)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(h8[e]))));
                uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(h8[e] >> 32))));
                const hent_t x = (static_cast<hent_t>(hi) << 32) | lo;
                h8[e] = static_cast<uint32_t>(e) < nd ? x : prev;
                prev = h8[e];
            }
        }
        // ---- the error bound of this history ----
        float A = 0.0f, V = 0.0f;
        for (uint32_t i = lane; i < nd; i += 64) {
            const hent_t x = hr[i];
            const float cnt = static_cast<float>(h_cnt(x));
            A = fmaf(cnt, d.lr_wmax[h_prod(x)], A);
            V += cnt;
        }
        for (int o = 32; o > 0; o >>= 1) { A += __shfl_xor(A, o); V += __shfl_xor(V, o); }
        // (the rows' own error: 2^-11 of wmax per weight from the fp16 copy, wmax / 254 from the 8-bit one)
        constexpr bool q8 = Q8;
        // (8-bit: the sums are taken on q + 128 and the offset removed at the end: partial sums up to ~3 A instead of A)
        const float B16 = (A * 4.8828125e-4f + V * 2.98023224e-8f +
                           static_cast<float>(nd + 3) * 5.9604644775390625e-08f * (d.lr_bmax + A)) * 1.02f;
        const float B = !q8 ? B16 : (A * 3.9764e-3f + V * 2.98023224e-8f +
                                     static_cast<float>(nd + 3) * 5.9604644775390625e-08f * (d.lr_bmax + 3.1f * A)) * 1.02f;
        float thr = 2.0f * B * 1.01f + 1e-30f;
        const uint32_t c_lo = r * RC, c_hi = min(c_lo + RC, C);
        float rb = -INFINITY;
        uint32_t n_cand = 0;
        bool overflow = false;
        // candidates of one batch of 512 classes (lane = 8 consecutive classes) against the running maximum
        auto collect = [&](uint32_t c, bool in, const float (&acc)[8]) {
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, acc[j]);
            if (!in) m = -INFINITY;
            rb = fmaxf(rb, wave_max_dpp(m));
            const float cut = rb - thr;
            if (__ballot(m >= cut) == 0ull) return;          // (no lane holds a class within the band: most batches but one)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool pass = in && acc[j] >= cut;
                const unsigned long long pm = __ballot(pass);
                if (pm && !overflow) {
                    const uint32_t np = static_cast<uint32_t>(__popcll(pm));
                    if (n_cand + np > kCandCap) overflow = true;
                    else {
                        if (pass) { const uint32_t k = n_cand + prefix_in_mask(pm); s_cand[wave][k] = c + j; s_cval[wave][k] = acc[j]; }
                        n_cand += np;
                    }
                }
            }
        };
        if (!q8) {
            // fp16 rows.  The first eight history rows of a batch and its intercepts are one round trip; rows beyond the eighth (long
            // histories) four at a time.  (The next batch's rows in flight while this one is summed — 40 more registers — LOST to
            // the waves per SIMD it costs: acts 198 ms at 3 waves with the prefetch, 173 at 4 without: ab_call28_c5.jsonl)
            struct Batch { uint4 hv[8]; float4 b0, b1; };
            const uint32_t p8[8] = {h_prod(h8[0]), h_prod(h8[1]), h_prod(h8[2]), h_prod(h8[3]), h_prod(h8[4]), h_prod(h8[5]), h_prod(h8[6]), h_prod(h8[7])};
            float cn8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cn8[e] = static_cast<uint32_t>(e) < nd ? static_cast<float>(h_cnt(h8[e])) : 0.0f;
            const unsigned short* rp[8];           // (the rows' base addresses once per act, not per batch)
#pragma unroll
            for (int e = 0; e < 8; ++e) rp[e] = d.lr_coef16_t + static_cast<size_t>(p8[e]) * C;
            auto fetch = [&](uint32_t c0, Batch& bt) {
                const uint32_t c = c0 + 8u * static_cast<uint32_t>(lane);
                const uint32_t cl = c < c_hi ? c : c_lo;
                bt.b0 = *reinterpret_cast<const float4*>(d.lr_intercept32 + cl);
                bt.b1 = *reinterpret_cast<const float4*>(d.lr_intercept32 + cl + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) bt.hv[e] = *reinterpret_cast<const uint4*>(rp[e] + cl);
            };
            Batch cur;
            fetch(c_lo, cur);
            for (uint32_t c0 = c_lo; c0 < c_hi && !overflow; c0 += 512) {
                const bool more = c0 + 512 < c_hi;
                const uint32_t c = c0 + 8u * static_cast<uint32_t>(lane);
                const bool in = c < c_hi;
                const uint32_t cl = in ? c : c_lo;
                float acc[8] = {cur.b0.x, cur.b0.y, cur.b0.z, cur.b0.w, cur.b1.x, cur.b1.y, cur.b1.z, cur.b1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {                        // (rows beyond the history: count 0, no branch)
                    const uint32_t wq[4] = {cur.hv[e].x, cur.hv[e].y, cur.hv[e].z, cur.hv[e].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[2 * j] = fma_mix_lo(cn8[e], wq[j], acc[2 * j]);
                        acc[2 * j + 1] = fma_mix_hi(cn8[e], wq[j], acc[2 * j + 1]);
                    }
                }
                for (uint32_t i0 = 8; i0 < nd; i0 += 4) {            // long histories: four more rows in flight
                    uint4 hv[4];
                    float cn[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const hent_t x = hr[min(i0 + e, nd - 1)];
                        cn[e] = i0 + e < nd ? static_cast<float>(h_cnt(x)) : 0.0f;
                        hv[e] = *reinterpret_cast<const uint4*>(d.lr_coef16_t + static_cast<size_t>(h_prod(x)) * C + cl);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t wq[4] = {hv[e].x, hv[e].y, hv[e].z, hv[e].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[2 * j] = fma_mix_lo(cn[e], wq[j], acc[2 * j]);
                            acc[2 * j + 1] = fma_mix_hi(cn[e], wq[j], acc[2 * j + 1]);
                        }
                    }
                }
                collect(c, in, acc);
                if (more) fetch(c0 + 512, cur);
            }
        } else {
            // 8-bit rows (q + 128: v_cvt_f32_ubyte*), count x scale folded into one factor per row, the offset 128 sum(count x scale)
            // taken off once at the end.  A lane takes kCpl8 = 20 consecutive classes — 20 bytes per row — so that a range of
            // <= 1 280 classes (C5: 1 256) is ONE round trip for its first eight history rows: the screen is a chain of dependent
            // round trips (list -> history -> rows), not bytes — the 8-bit rows at 8 classes per lane took as long as the fp16 ones
            // (profiles/r6/ab_call29_c5.jsonl).  Rows are read up to 16 bytes past a range's last class (rg_sim_set_logreg_int8: the
            // array carries 16 bytes of padding).
            constexpr int kCpl8 = 20;
            struct __attribute__((packed, aligned(4))) Row8 { uint32_t w[kCpl8 / 4]; };
            uint32_t p8[8];
            float cs8[8];
            float corr8 = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p8[e] = h_prod(h8[e]);
                cs8[e] = static_cast<uint32_t>(e) < nd ? static_cast<float>(h_cnt(h8[e])) * d.lr_scale8[p8[e]] : 0.0f;   // (p8 is a product always)
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) corr8 = fmaf(cs8[e], 128.0f, corr8);
            for (uint32_t c0 = c_lo; c0 < c_hi && !overflow; c0 += 64 * kCpl8) {
                const uint32_t c = c0 + static_cast<uint32_t>(kCpl8) * static_cast<uint32_t>(lane);
                const bool in = c < c_hi;                            // (some of the lane's classes; each is checked below)
                const uint32_t cl = in ? c : c_lo;
                Row8 bv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) bv[e] = *reinterpret_cast<const Row8*>(d.lr_coef8_t + static_cast<size_t>(p8[e]) * C + cl);
                float acc[kCpl8];
#pragma unroll
                for (int q = 0; q < kCpl8 / 4; ++q) {
                    const float4 b = *reinterpret_cast<const float4*>(d.lr_intercept32 + min(cl + 4u * q, C - 4u));
                    acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                }
                float corr = corr8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int j = 0; j < kCpl8; ++j)
                        acc[j] = fmaf(cs8[e], static_cast<float>((bv[e].w[j / 4] >> (8 * (j % 4))) & 0xFFu), acc[j]);
                }
                for (uint32_t i0 = 8; i0 < nd; i0 += 2) {            // long histories: two more rows in flight
                    Row8 bw[2];
                    float cs[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const hent_t x = hr[min(i0 + e, nd - 1)];
                        const uint32_t pp = h_prod(x);
                        cs[e] = static_cast<float>(h_cnt(x)) * d.lr_scale8[pp] * (i0 + e < nd ? 1.0f : 0.0f);     // (the load unconditional)
                        bw[e] = *reinterpret_cast<const Row8*>(d.lr_coef8_t + static_cast<size_t>(pp) * C + cl);
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        corr = fmaf(cs[e], 128.0f, corr);
#pragma unroll
                        for (int j = 0; j < kCpl8; ++j)
                            acc[j] = fmaf(cs[e], static_cast<float>((bw[e].w[j / 4] >> (8 * (j % 4))) & 0xFFu), acc[j]);
                    }
                }
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < kCpl8; ++j) {
                    acc[j] -= corr;
                    if (c + j < c_hi) m = fmaxf(m, acc[j]);
                }
                if (!in) m = -INFINITY;
                rb = fmaxf(rb, wave_max_dpp(m));
                const float cut = rb - thr;
                // (few classes pass: one ballot over "any of mine", then the passing lanes' classes one by one)
                bool any = false;
#pragma unroll
                for (int j = 0; j < kCpl8; ++j) any = any || (c + j < c_hi && acc[j] >= cut);
                if (!in) any = false;
                if (__ballot(any)) {
#pragma unroll
                    for (int j = 0; j < kCpl8; ++j) {
                        const bool pass = in && c + j < c_hi && acc[j] >= cut;
                        const unsigned long long pm = __ballot(pass);
                        if (pm && !overflow) {
                            const uint32_t np = static_cast<uint32_t>(__popcll(pm));
                            if (n_cand + np > kCandCap) overflow = true;
                            else {
                                if (pass) { const uint32_t k = n_cand + prefix_in_mask(pm); s_cand[wave][k] = c + j; s_cval[wave][k] = acc[j]; }
                                n_cand += np;
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (q8 && !overflow) {
            // ---- second level of the 8-bit screen: the range's true maximum is among the candidates above (within 2 B8 of the
            // range's 8-bit maximum); their scores once more from the fp16 rows — a lane per candidate, the fp16 pass's own
            // arithmetic (intercept32, then fma in history order), so that pass's bound B16 holds for them: from here on the
            // range's maximum is the candidates' fp16 maximum and the band 2 B16, and k_logreg_decide works as it does on the
            // fp16 screen (the true argmax is a candidate, its fp16 score within 2 B16 of any other candidate's) ----
            const bool mine2 = static_cast<uint32_t>(lane) < n_cand;
            const uint32_t cc = mine2 ? s_cand[wave][lane] : c_lo;
            float acc = d.lr_intercept32[cc];
            for (uint32_t i0 = 0; i0 < nd; i0 += 8) {                // eight values in flight
                unsigned short wv[8];
                float cn[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const hent_t x = hr[min(i0 + e, nd - 1)];
                    cn[e] = i0 + e < nd ? static_cast<float>(h_cnt(x)) : 0.0f;
                    wv[e] = d.lr_coef16_t[static_cast<size_t>(h_prod(x)) * C + cc];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = fmaf(cn[e], static_cast<float>(__builtin_bit_cast(_Float16, wv[e])), acc);
            }
            float m2 = mine2 ? acc : -INFINITY;
            if (mine2) s_cval[wave][lane] = acc;
            for (int o = 32; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o));
            rb = m2;
            thr = 2.0f * B16 * 1.01f + 1e-30f;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
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
    const uint32_t n = min(d.lr_cnt[t], d.lr_part_cap);
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
                    for (uint32_t i0 = 0; i0 < nd; i0 += 8) {        // eight coefficients in flight, summed in history order
                        double wv[8], cn[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const hent_t x = hr[min(i0 + e, nd - 1)];
                            cn[e] = static_cast<double>(h_cnt(x));
                            wv[e] = d.lr_coef_t[static_cast<size_t>(h_prod(x)) * C + cc];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (i0 + e < nd) sc = __dadd_rn(sc, __dmul_rn(cn[e], wv[e]));
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
    // the counters once per BLOCK: per wave — one act each on a grid sized for the step's worst case — these three atomics on three
    // addresses were the kernel's whole time (profiles/r6/c5_fp16_kernel_stats_call31.csv)
    __shared__ unsigned long long s_cnt[3];
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0ull;
    __syncthreads();
    if (lane == 0 && c_acts) {
        atomicAdd(&s_cnt[0], c_acts);
        atomicAdd(&s_cnt[1], c_rows);
        if (c_exact) atomicAdd(&s_cnt[2], c_exact);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt[0]) {
        atomicAdd(&d.counters[RG_CNT_LR_ACTS], s_cnt[0]);
        atomicAdd(&d.counters[RG_CNT_LR_ROWS], s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&d.counters[RG_CNT_LR_EXACT], s_cnt[2]);
    }
}

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
                const bool need_ctr = !d.env_kind && (d.aux_pclick != nullptr || !(u_click < kNoClickBelow));
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
                if (d.policy == RG_POLICY_EXTERNAL) {
                    // (an action outside [0, P) — e.g. the -1 of a caller that had none — must not index beta / mu_b: product 0)
                    const int32_t ai = actions[uidx];
                    const bool bad = ai < 0 || static_cast<uint32_t>(ai) >= d.P;
                    if (bad) atomicAdd(&d.counters[RG_CNT_BAD_ACTION], 1ull);      // (a caller bug: counted, see recogym_hip.h)
                    a = bad ? 0u : static_cast<uint32_t>(ai);
                    ps = __builtin_nan("");
                }
                else if (d.policy == RG_POLICY_LOGREG_FROZEN) { a = lr_a; ps = d.lr_sample ? d.lr_ps[uidx] : 1.0; }
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
                if (d.env_kind) {      // reco-gym-v0: click_probs[action][view], binomial(1, p) (reco_env_v0.py:61-63)
                    const size_t cell = static_cast<size_t>(a) * d.P + d.pv0[uidx];
                    ctr = d.e0_click_p[cell];
                    click = env0_click(d, cell, user, t, u_click);
                } else
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
                    uint32_t a;
                    if (d.policy != RG_POLICY_LOGREG_FROZEN) a = policy_act(d, slot, user, t + 1, &ps);
                    else if (!d.lr_sample) a = lr_a;
                    else if (is_org) { a = lr_a; ps = d.lr_ps[uidx]; }                       // k_logreg_sample drew it for event t + 1
                    else { a = d.lr_action2[uidx]; ps = d.lr_ps2[uidx]; }
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

// ------------------------------------------------------------------------------------------
// k_advance_run — k_advance for run-ahead rounds (rg_sim_run to the end; DevSim::run_ahead): a user goes from the event the round
// found it at — an organic event the sweep has just drawn, or a bandit event — through every following event that needs nothing
// from another kernel: omega only moves when a transition draws "organic" (reco_env_v1.py:95-98, change_omega_for_bandits off),
// the view history and the last viewed product only at organic events, so a whole bandit run (~17 events) is decided by the
// user's state as it is now.  The round ends for the user at the first event whose drawn next state is not "bandit" (organic:
// the next round's sweep draws the product; stop), at a bandit event whose uniform can click (kNoClickBelow: 3 % of them — a
// click sends the user organic), or after `hops` (<= kRunAheadMax) events.
//   pass 1  a lane per user, the event draws only (one Philox block per event): how many events L the round takes the user
//           through, and the drawn state after the last -> the user's bandit rows; a block prefix sum and ONE atomic reserve the
//           block's raw rows: every row is used, a user's rows are consecutive;
//   pass 2  a lane per EVENT: the events of a run are independent of each other (draws addressed by (user, event index), the
//           state they read does not move inside the run), so the wave's rows are dealt to its lanes 64 at a time — owner lane and
//           position from a byte map in LDS, the owner's registers by ds_bpermute — instead of every lane walking its own run
//           while the wave waits for its longest (measured: rounds of <= 32 events were SLOWER than lock-step that way);
//           the last event's click goes back to the owner through LDS;
//   then    the owner closes the books as k_advance does: click override, stop + phantom row, next lists, drift list.
// Per-user clocks (NormalTimeGenerator) chain the events of a run (every row carries the clock so far): that mode keeps the lane
// per user walk.  The lists, the drift list and the per-round counters are k_advance's; ev[user] moves on by L.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kRunAheadMax = 64;
// timing experiments of k_advance_run (RECOGYM_ABLATE bits 24: no pass 2, 25: no act, 26: no row stores) exist in -DRG_ADV_TIMING
// builds only (profiles/r4/ab_call22_advance_run_timing.jsonl)
#ifdef RG_ADV_TIMING
#define RG_ADV_ABL(bit) (d.ablate & (1u << (bit)))
#else
#define RG_ADV_ABL(bit) (false)
#endif

// one bandit event of a user whose state does not move: act, click, row (k_advance's arithmetic) -> click
// (may_click = false: pass 1 saw the event's uniform below kNoClickBelow — it cannot click whatever the click probability is, and
// the event draw, a Philox block, is not taken again)
__device__ __forceinline__ bool run_bandit_event(const DevSim& d, uint32_t slot, uint32_t user, uint32_t te, uint32_t lr_a,
                                                 uint64_t row, double clock, bool may_click) {
    double u_click = 0.0;
    if (may_click) {
        const rg_u32x4 w = rg_draw(d.seed, user, te, 0, RG_DRAW_EVENT);
        u_click = rg_uniform(w.w[0], w.w[1]);
    }
    const bool need_ctr = d.aux_pclick != nullptr || may_click;
    double ps = 1.0;
    uint32_t a = 0;
    if (RG_ADV_ABL(25)) {}
    else if (d.policy == RG_POLICY_LOGREG_FROZEN) a = lr_a;
    else a = policy_act(d, slot, user, te, &ps);
    double ctr = 0.0;
    bool click = false;
    if (need_ctr) {
        // beta[a] . omega, k ascending (the oracle's association), loads eight k at a time
        const double* b = d.beta + static_cast<size_t>(a) * d.K;
        const double* om = d.omega + static_cast<size_t>(slot) * d.OMS;
        double x = 0.0;
        for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {
            double wv[8], bv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t k = min(k0 + q, d.K - 1);
                wv[q] = om[k];
                bv[q] = b[k];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (k0 + q < d.K) x += bv[q] * wv[q];
        }
        ctr = ff64(x + d.mu_b[a]);
        const double p0 = 1.0 - ctr;
        click = (p0 / (p0 + ctr)) <= u_click;
    }
    if (d.log && row < d.log_cap && !RG_ADV_ABL(26)) {
        rg_event e;
        e.u = user; e.t = te;
        e.code = RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a;
        e.ps = static_cast<float>(ps);
        d.log[row] = e;
        if (d.aux_ps) d.aux_ps[row] = ps;
        if (d.aux_pclick) d.aux_pclick[row] = ctr;
        if (d.aux_time) d.aux_time[row] = clock;
    }
    return click;
}

#ifndef RG_ADV_RUN_WAVES
#define RG_ADV_RUN_WAVES 2
#endif
__global__ void __launch_bounds__(kAdvBlock) __attribute__((amdgpu_waves_per_eu(RG_ADV_RUN_WAVES, RG_ADV_RUN_WAVES)))
k_advance_run(DevSim d, uint32_t t, uint32_t hops) {
    constexpr int NW = kAdvBlock / 64;
    __shared__ uint32_t s_cnt_o[NW], s_cnt_b[NW], s_cnt_d[NW], s_rows[NW], s_base_o, s_base_b, s_base_d;
    __shared__ unsigned long long s_row0;
    __shared__ uint8_t s_owner[NW][64 * kRunAheadMax], s_click[NW][64];
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_b = d.step_cnt[2 * t + RG_STATE_BANDIT];
    const uint32_t n = n_o + n_b;
    const uint32_t* cur_o = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const uint32_t* cur_b = list_ptr(d, t & 1, RG_STATE_BANDIT);
    uint32_t* next_o = list_ptr(d, (t + 1) & 1, RG_STATE_ORGANIC);
    uint32_t* next_b = list_ptr(d, (t + 1) & 1, RG_STATE_BANDIT);
    uint32_t* next_cnt = d.step_cnt + 2 * (t + 1);
    const int wave = threadIdx.x >> 6, lane = lane_id();
    hops = min(hops, kRunAheadMax);

    uint32_t clicks = 0, phantoms = 0, extra = 0, max_t = 0;
    const uint32_t n_iter = (n + kAdvBlock - 1) / kAdvBlock;
    for (uint32_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
        const uint32_t i = it * kAdvBlock + threadIdx.x;
        const bool live = i < n;
        const bool is_org = i < n_o;
        uint32_t slot = 0, uidx = 0, user = 0, te0 = 0, L = 0, lr_a = 0;
        bool organic_only = false;
        int ns = RG_STATE_STOP;       // drawn state after the round's last event; inactive lanes look dead
        bool may_click = false;       // the round's last event is a bandit event whose uniform can click
        // ---- pass 1: how far this round takes the user
        if (live) {
            slot = is_org ? cur_o[i] : cur_b[i - n_o];
            uidx = d.uid[slot];
            user = static_cast<uint32_t>(d.first_user + uidx);
            te0 = d.ev[uidx];
            organic_only = (d.first_user + uidx) < d.organic_only_below;
            if (d.policy == RG_POLICY_LOGREG_FROZEN) lr_a = d.lr_action[uidx];     // one act serves the whole run and the phantom row
            bool org = is_org;
            for (;;) {
                const rg_u32x4 w = rg_draw(d.seed, user, te0 + L, 0, RG_DRAW_EVENT);
                const double u_trans = rg_uniform(w.w[2], w.w[3]);
                const double c0 = org ? d.cdf_o0 : d.cdf_b0, c1 = org ? d.cdf_o1 : d.cdf_b1;
                ns = (c0 <= u_trans) + (c1 <= u_trans);
                L += 1;
                if (!org && !(rg_uniform(w.w[0], w.w[1]) < kNoClickBelow)) { may_click = true; break; }   // the round's last event
                if (ns != RG_STATE_BANDIT || organic_only || L >= hops) break;
                org = false;
            }
        }
        const uint32_t skip = (live && is_org) ? 1u : 0u;               // the organic event's row is the sweep's
        const uint32_t n_rows = L - skip;
        uint32_t incl = n_rows;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        const uint32_t excl = incl - n_rows;
        const uint32_t rows_w = __shfl(incl, 63);
        if (lane == 63) s_rows[wave] = incl;
        if (!d.time_mode) {
            for (uint32_t q = 0; q < n_rows; ++q) s_owner[wave][excl + q] = static_cast<uint8_t>(lane);
            s_click[wave][lane] = 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long tot = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) tot += s_rows[w2];
            s_row0 = tot ? atomicAdd(&d.run_ctl[0], tot) : 0ull;
        }
        __syncthreads();
        uint64_t row_w = s_row0;                                        // first raw row of this wave's users
        for (int w2 = 0; w2 < wave; ++w2) row_w += s_rows[w2];

        // ---- pass 2: the events
        bool click = false;           // of the round's last event
        double drift_sig = d.sigma_omega;
        double clock = 0.0;
        if (!d.time_mode) {
            for (uint32_t k0 = 0; k0 < rows_w; k0 += 64) {
                const uint32_t k = k0 + lane;
                const bool mine = k < rows_w && !RG_ADV_ABL(24);
                const int owner = mine ? s_owner[wave][k] : 0;
                const uint32_t o_slot = __shfl(slot, owner), o_user = __shfl(user, owner), o_lr = __shfl(lr_a, owner);
                const uint32_t o_te = __shfl(te0 + skip, owner), o_excl = __shfl(excl, owner), o_rows = __shfl(n_rows, owner);
                const bool o_mc = __shfl(static_cast<int>(may_click), owner) != 0;
                if (mine) {
                    const uint32_t h = k - o_excl;
                    const bool c = run_bandit_event(d, o_slot, o_user, o_te + h, o_lr, row_w + k, 0.0, o_mc && h + 1 == o_rows);
                    clicks += c;
                    if (c && h + 1 == o_rows) s_click[wave][owner] = 1;
                }
            }
            __syncthreads();
            click = live && n_rows && s_click[wave][lane] != 0;
        } else if (live) {
            // per-user clocks: the lane walks its own run (the clock of every row is the sum of the time steps before it)
            clock = d.utime[uidx];
            uint64_t row = row_w + excl;
            for (uint32_t h = 0; h < L; ++h) {
                const uint32_t te = te0 + h;
                if (h >= skip) {
                    click = run_bandit_event(d, slot, user, te, lr_a, row, clock, may_click && h + 1 == L);
                    clicks += click;
                    row += 1;
                }
                double z0, z1;
                normal_pair(d.seed, user, te, 0, RG_DRAW_TIME, &z0, &z1);
                const double dt = fabs(d.time_mu + d.time_sigma * z0);
                clock = clock + dt;
                drift_sig = d.sigma_omega * (dt == 0.0 ? 1.0 : dt);
            }
            d.utime[uidx] = clock;
        }

        // ---- the owner closes the round (update_state, reco_env_v1.py:85-100)
        bool drift_me = false;
        if (live) {
            // omega drifts when the DRAWN next state is organic (the click override below does not redraw it)
            drift_me = d.sigma_omega != 0.0 && (d.change_omega_for_bandits || ns == RG_STATE_ORGANIC);
            const uint32_t t_last = te0 + L - 1u;
            d.ev[uidx] = te0 + L;
            extra += L - 1u;
            if (click) ns = RG_STATE_ORGANIC;          // abstract.py:180-181
            if (organic_only && ns != RG_STATE_ORGANIC) {
                ns = RG_STATE_STOP;                    // warm-up users end with their first session
                d.n_events[uidx] = t_last + 1;
                max_t = max(max_t, t_last + 1);
            } else if (ns == RG_STATE_STOP) {
                d.n_events[uidx] = t_last + 1;
                max_t = max(max_t, t_last + 1);
                // final step_offline(done=True): one more act, reward 0 (abstract.py:223-233,311-316)
                double ps = 1.0;
                const uint32_t a = d.policy == RG_POLICY_LOGREG_FROZEN ? lr_a : policy_act(d, slot, user, t_last + 1, &ps);
                rg_event e;
                e.u = user; e.t = t_last + 1; e.code = RG_EV_BANDIT | RG_EV_PHANTOM | a;
                e.ps = static_cast<float>(ps);
                d.phantom[uidx] = e;
                d.phantom_ps[uidx] = ps;
                if (d.time_mode) d.phantom_time[uidx] = clock;
                d.has_phantom[uidx] = 1;
                phantoms += 1;
            }
        }
        // ordered compaction of the survivors into the next round's lists (as k_advance)
        const unsigned long long mo = __ballot(ns == RG_STATE_ORGANIC), mb = __ballot(ns == RG_STATE_BANDIT), md = __ballot(drift_me);
        if (lane == 0) { s_cnt_o[wave] = __popcll(mo); s_cnt_b[wave] = __popcll(mb); s_cnt_d[wave] = __popcll(md); }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t to = 0, tb = 0, td = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) { to += s_cnt_o[w2]; tb += s_cnt_b[w2]; td += s_cnt_d[w2]; }
            s_base_d = td ? atomicAdd(&d.drift_cnt[t], td) : 0u;
            unsigned long long base = 0;
            if (to | tb)
                base = atomicAdd(reinterpret_cast<unsigned long long*>(next_cnt),
                                 static_cast<unsigned long long>(to) | (static_cast<unsigned long long>(tb) << 32));
            s_base_o = static_cast<uint32_t>(base);
            s_base_b = static_cast<uint32_t>(base >> 32);
        }
        __syncthreads();
        uint32_t wo = s_base_o, wb = s_base_b, wd = s_base_d;
        for (int w2 = 0; w2 < wave; ++w2) { wo += s_cnt_o[w2]; wb += s_cnt_b[w2]; wd += s_cnt_d[w2]; }
        if (ns == RG_STATE_ORGANIC) next_o[wo + prefix_in_mask(mo)] = slot;
        if (ns == RG_STATE_BANDIT) next_b[wb + prefix_in_mask(mb)] = slot;
        if (drift_me) {
            const uint32_t e = wd + prefix_in_mask(md);
            d.drift_list[e] = slot;
            if (d.time_mode) d.drift_sig[e] = drift_sig;
        }
        __syncthreads();
    }
    // counters: one atomic per wave per kernel (the events beyond a listed user's first are bandit events no list counts)
    for (int o = 32; o > 0; o >>= 1) {
        clicks += __shfl_xor(clicks, o); phantoms += __shfl_xor(phantoms, o); extra += __shfl_xor(extra, o);
        max_t = max(max_t, static_cast<uint32_t>(__shfl_xor(static_cast<int>(max_t), o)));
    }
    if (lane == 0) {
        if (clicks) atomicAdd(&d.counters[RG_CNT_CLICKS], static_cast<unsigned long long>(clicks));
        if (phantoms) atomicAdd(&d.counters[RG_CNT_PHANTOM], static_cast<unsigned long long>(phantoms));
        if (extra) atomicAdd(&d.counters[kCntTailBandit], static_cast<unsigned long long>(extra));
        if (max_t) atomicMax(&d.counters[kCntTailMaxT], static_cast<unsigned long long>(max_t));
    }
}

// the raw-log books of run-ahead rounds: round r's rows are [log_base[r], log_base[r + 1]) = the sweep's organic rows (position in
// the organic list) and then the bandit rows k_advance_run reserved; one thread, before round 0 and after every k_advance_run
__global__ void k_round_rows(DevSim d, uint32_t t, uint32_t begin) {
    if (begin) { d.run_ctl[0] = d.log_base[0] + d.step_cnt[RG_STATE_ORGANIC]; return; }
    const unsigned long long next = d.run_ctl[0];
    d.log_base[t + 1] = next;
    d.run_ctl[0] = next + d.step_cnt[2 * (t + 1) + RG_STATE_ORGANIC];
}

// k_drift — omega <- omega + sigma_omega (time delta) Z(K) (reco_env_v1.py:95-98) of the users k_advance listed at step t: a lane per
// (user, Box-Muller pair), the K normals addressed by (user, t, pair) as everywhere else.
__global__ void __launch_bounds__(kBlock) k_drift(DevSim d, uint32_t t) {
    const uint32_t n = d.drift_cnt[t];
    const uint32_t KP = (d.K + 1) / 2;
    const uint64_t items = static_cast<uint64_t>(n) * KP;
    for (uint64_t it = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; it < items; it += static_cast<uint64_t>(gridDim.x) * kBlock) {
        const uint32_t e = static_cast<uint32_t>(it / KP), j = static_cast<uint32_t>(it % KP);
        const uint32_t slot = d.drift_list[e];
        const uint32_t uidx = d.uid[slot];
        const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
        const double sig = d.time_mode ? d.drift_sig[e] : d.sigma_omega;
        double z0, z1;
        // run-ahead rounds: the drifting event is the last one k_advance_run took the user through
        normal_pair(d.seed, user, d.run_ahead ? d.ev[uidx] - 1u : t, j, RG_DRAW_DRIFT, &z0, &z1);
        double* o0 = d.omega + static_cast<size_t>(slot) * d.OMS + 2 * j;
        *o0 = *o0 + sig * z0;
        if (2 * j + 1 < d.K) { double* o1 = o0 + 1; *o1 = *o1 + sig * z1; }
    }
}
search_kernel_t drift_kernel() { return k_drift; }

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
        const uint32_t t_first = d.run_ahead ? d.ev[uidx] : t0;      // run-ahead rounds: the user's own event index
        for (uint32_t t = t_first;; ++t) {
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
                if (t > t_first) { if (state == RG_STATE_ORGANIC) c_org += 1; else c_ban += 1; }
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
search_kernel_t logreg_screen_kernel(bool q8) { return q8 ? k_logreg_screen<true> : k_logreg_screen<false>; }
search_kernel_t logreg_decide_kernel() { return k_logreg_decide; }
search_kernel_t logreg_sample_kernel() { return k_logreg_sample; }
advance_kernel_t advance_kernel() { return k_advance; }
advance_run_kernel_t advance_run_kernel() { return k_advance_run; }
round_rows_kernel_t round_rows_kernel() { return k_round_rows; }
search_kernel_t tail_kernel() { return k_tail; }

}  // namespace rgk
