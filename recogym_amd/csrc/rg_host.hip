// rg_host.hip — librecogym_hip.so, unit 1 of 7: host code (launch logic, the C ABI) and the small kernels (reset, table copies, sort / scatter, exports, test hooks).
// (see rg_common.hpp for the shared types and helpers, DESIGN.md for the data layout and the rooflines)

#include "rg_common.hpp"

#ifndef RG_TP_SMEM_PAD
#define RG_TP_SMEM_PAD 0
#endif
namespace rgk {

// ------------------------------------------------------------------------------------------
// k_reset_users
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_reset_users(DevSim d) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.step_cnt[0] = d.n_users;   // everyone starts organic (abstract.py:93)
        d.step_cnt[1] = 0;
        d.log_base[0] = d.debug_row_base;
    }
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        const uint32_t user = static_cast<uint32_t>(d.first_user + i);
        if (d.env_kind) {      // reco-gym-v0: the first product view (reco_env_v0.py:52-54), no omega
            const rg_u32x4 w = rg_draw(d.seed, user, 0u, 0u, RG_DRAW_RESET);
            d.pv0[i] = upper_bound_f64(d.e0_cdf_init, d.P, rg_uniform(w.w[0], w.w[1]));
        } else
        for (uint32_t j = 0; 2 * j < d.K; ++j) {
            double z0, z1;
            normal_pair(d.seed, user, 0u, j, RG_DRAW_RESET, &z0, &z1);
            d.omega[static_cast<size_t>(i) * d.OMS + 2 * j] = 0.0 + d.sigma0 * z0;
            if (2 * j + 1 < d.K) d.omega[static_cast<size_t>(i) * d.OMS + 2 * j + 1] = 0.0 + d.sigma0 * z1;
        }
        list_ptr(d, 0, RG_STATE_ORGANIC)[i] = i;
        d.uid[i] = i;
        d.n_events[i] = 0;
        d.ev[i] = 0;
        d.has_phantom[i] = 0;
        if (d.time_mode) d.utime[i] = 0.0;
        if (d.lr_dirty) d.lr_dirty[i] = 1;
        if (d.hist_cap) d.hist[static_cast<size_t>(i) * d.hist_cap] = 0ull;
        if (d.use_cache) { d.f64_valid[i] = 0; d.cache_resc[i] = 0; }
    }
}

// reco-gym-v0: update_product_view (reco_env_v0.py:65-67) of the step's organic users — the next view is drawn inside the cluster of
// the current one: choice(P, p = product_transition[view]) = cluster start + searchsorted(cdf_cluster, u, 'right').
__global__ void __launch_bounds__(kBlock) k_draw_env0(DevSim d, uint32_t t) {
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    for (uint32_t pos = blockIdx.x * kBlock + threadIdx.x; pos < n_o; pos += gridDim.x * kBlock) {
        const uint32_t slot = cur[pos], uidx = d.uid[slot];
        const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
        const double u = organic_uniform(d, uidx, user, t);
        const uint32_t view = d.pv0[uidx];
        uint32_t v = (view / d.e0_cluster) * d.e0_cluster + upper_bound_f64(d.e0_cdf_cluster, d.e0_cluster, u);
        if (v >= d.P) v = d.P - 1;
        d.pv0[uidx] = v;
        write_organic_row(d, t, pos, slot, user, v);
        if (d.hist_cap) history_add(d, slot, v);
    }
}

// fp32 copies of Gamma / mu_organic for the MFMA path: gamma32 [P_pad][KS] (columns >= K and rows
// >= P are zero), mu32 [P_pad] (-inf beyond P, so padded products get probability exactly 0).
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

// gsplit[p] = [G1(K) | G2(K) | G3(K) | 0 ... 0 | 1 1 1] (bf16), the A operand rows of the split-bf16
// kernel, G = fl32(Gamma log2 e): the MFMA then yields logits in log2 units, and the three ones
// multiply the three bf16 pieces of -reference that sit in the user's B row.
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

// float64 transpose of Gamma for the float64 draw kernel: lane-per-product reads coalesce
__global__ void __launch_bounds__(kBlock) k_make_gammaT(DevSim d) {
    const size_t n = static_cast<size_t>(d.K) * d.PT;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t k = i / d.PT, p = i % d.PT;
        d.gammaT[i] = p < d.P ? d.gamma[p * d.K + k] : 0.0;
    }
}

__global__ void __launch_bounds__(kBlock) k_make_beta32(DevSim d) {
    const size_t n = static_cast<size_t>(d.P) * d.KB4;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t p = i / d.KB4, k = i % d.KB4;
        d.beta32[i] = k < d.K ? static_cast<float>(d.beta[p * d.K + k]) : 0.0f;
    }
}

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

// Table statistics for the logit error bound of the MFMA path (one block per statistic):
//   block k < 2KH : max_p |Gamma[p][k]|      block 2KH : max_p ||Gamma[p]||_2
//   block 2KH+1   : max_p |mu_o[p]|
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

// closes the books of the tail: step t0 + 1 exists, is empty, and starts after the tail's rows
__global__ void k_tail_finish(DevSim d, uint32_t t0) {
    d.log_base[t0 + 1] = d.log_base[t0] + d.counters[kCntTailRows];
    d.step_cnt[2 * (t0 + 1)] = 0;
    d.step_cnt[2 * (t0 + 1) + 1] = 0;
}

// closes the books of a walked run: no lock-step step holds events; step 1 exists, is empty and starts after the raw rows
__global__ void k_walk_finish(DevSim d) {
    d.step_cnt[0] = 0; d.step_cnt[1] = 0; d.step_cnt[2] = 0; d.step_cnt[3] = 0;
    d.log_base[0] = d.debug_row_base;
    d.log_base[1] = d.counters[kCntTailRows];
}

// totals that are sums over the per-step counts
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

__global__ void __launch_bounds__(kBlock) k_export_omega(DevSim d, double* out) {
    const size_t n = static_cast<size_t>(d.n_users) * d.K;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t u = i / d.K, k = i % d.K;
        out[i] = d.omega[u * d.OMS + k];
    }
}

// test hooks
__global__ void __launch_bounds__(kBlock) k_debug_set_omega(DevSim d, const double* in) {
    const size_t n = static_cast<size_t>(d.n_users) * d.K;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t u = i / d.K, k = i % d.K;
        d.omega[u * d.OMS + k] = in[i];
    }
}

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

// live users only (after a repack the slots of users that left are gone); `out` is zero-filled first
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

// ------------------------------------------------------------------------------------------
// repack: the live lists lose their order step by step (the block that reserves first writes
// first) and thin out as users leave, so the per-user gathers of omega / the view history turn
// into scattered single-line fetches (measured: k_advance 0.22 -> 0.57 ns/event between steps
// 0-20 and 220-240 of the 10 M-user run).  Every few steps the state of the users still alive is
// therefore copied into the second buffer in list order — new slot = position in [organic |
// bandit] — and the lists become the identity.  Pure relabelling: user ids travel in uid[].
// ------------------------------------------------------------------------------------------
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

__global__ void __launch_bounds__(kBlock) k_repack_lists(DevSim d, uint32_t t) {
    const uint32_t n_o = d.step_cnt[2 * t], n_b = d.step_cnt[2 * t + 1];
    uint32_t* cur_o = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    uint32_t* cur_b = list_ptr(d, t & 1, RG_STATE_BANDIT);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_o + n_b; i += gridDim.x * kBlock) {
        if (i < n_o) cur_o[i] = i;
        else cur_b[i - n_o] = i;
    }
}

// ------------------------------------------------------------------------------------------
// log reordering: rows of user u occupy [off[u], off[u] + n_events[u] + has_phantom[u])
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_rows_per_user(DevSim d, int64_t* rows) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock)
        rows[i] = static_cast<int64_t>(d.n_events[i]) + d.has_phantom[i];
}

// exclusive scan, three phases (block sums -> scan of sums by one block -> add)
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

__global__ void __launch_bounds__(kBlock) k_scan_add(int64_t* out, const int64_t* block_sums, uint32_t n) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] += block_sums[blockIdx.x];
}

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

// k_scatter_rows through LDS.  The walk writes its raw log a wave iteration at a time — one row of each of its 64 users — so 64
// consecutive raw rows go to 64 different places of the ordered log, 16 bytes each: the plain scatter above ran at 1.6 TB/s of
// rows read + written (19.2 ms of rg_sim_sort_log's 20.2 on the C3 log: profiles/r6/sort_kernel_stats_call37.csv).  Here a block
// takes a TILE of kSortTile consecutive raw rows, groups them by user in LDS (a small hash table of the tile's users: count, first
// and last event index; a user's rows inside a tile are a run of consecutive event indices) and writes every user's run as one
// contiguous piece.  A tile whose users do not fit the table (the logs of lock-step / round runs: one row per user), or where a
// user's indices have a gap, is written the plain way.
constexpr uint32_t kSortTile = 2048, kSortHash = 512;
__global__ void __launch_bounds__(kBlock, 3) k_scatter_rows_tiled(DevSim d, uint64_t n_rows, const int64_t* off,
                                                                 rg_event* out, uint64_t out_cap) {
    constexpr uint32_t RPT = kSortTile / kBlock;               // rows per thread
    __shared__ rg_event s_rows[kSortTile];
    __shared__ unsigned short s_slot[kSortTile];
    __shared__ uint32_t h_key[kSortHash], h_tmin[kSortHash], h_tmax[kSortHash], h_cnt[kSortHash], h_base[kSortHash];
    __shared__ unsigned long long h_dst[kSortHash];
    __shared__ uint32_t s_scan[kBlock], s_bad;
    const uint64_t n_tiles = (n_rows + kSortTile - 1) / kSortTile;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t r0 = tile * kSortTile;
        for (uint32_t i = threadIdx.x; i < kSortHash; i += kBlock) { h_key[i] = 0xFFFFFFFFu; h_tmin[i] = 0xFFFFFFFFu; h_tmax[i] = 0u; h_cnt[i] = 0u; }
        if (threadIdx.x == 0) s_bad = 0u;
        __syncthreads();
        rg_event e[RPT];
        uint32_t slot[RPT];
#pragma unroll
        for (uint32_t q = 0; q < RPT; ++q) {                     // (coalesced: thread x takes rows x, x + 256, ...)
            const uint64_t r = r0 + q * kBlock + threadIdx.x;
            if (r < n_rows) e[q] = d.log[r];
            else { e[q].u = 0; e[q].t = 0; e[q].code = kHoleCode; e[q].ps = 0.0f; }
        }
#pragma unroll
        for (uint32_t q = 0; q < RPT; ++q) {
            slot[q] = 0xFFFFFFFFu;
            if (e[q].code == kHoleCode) continue;
            if (e[q].u == 0xFFFFFFFFu) { s_bad = 1u; continue; }   // (the highest user id is the table's "empty" key: the plain path)
            uint32_t h = (e[q].u * 2654435761u) >> 23;            // 9 bits
            for (uint32_t probe = 0; probe < 16; ++probe) {
                const uint32_t k = atomicCAS(&h_key[h], 0xFFFFFFFFu, e[q].u);
                if (k == 0xFFFFFFFFu || k == e[q].u) { slot[q] = h; break; }
                h = (h + 1) & (kSortHash - 1);
            }
            if (slot[q] == 0xFFFFFFFFu) { s_bad = 1u; continue; }
            atomicMin(&h_tmin[slot[q]], e[q].t);
            atomicMax(&h_tmax[slot[q]], e[q].t);
            atomicAdd(&h_cnt[slot[q]], 1u);
        }
        __syncthreads();
        // runs without a gap?  their places in the tile: an exclusive scan of the counts in table order
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x * (kSortHash / kBlock); i < (threadIdx.x + 1) * (kSortHash / kBlock); ++i) {
            if (h_cnt[i] && h_tmax[i] - h_tmin[i] + 1u != h_cnt[i]) s_bad = 1u;
            mine += h_cnt[i];
        }
        s_scan[threadIdx.x] = mine;
        __syncthreads();
        for (uint32_t o = 1; o < kBlock; o <<= 1) {
            const uint32_t y = threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0u;
            __syncthreads();
            s_scan[threadIdx.x] += y;
            __syncthreads();
        }
        const bool bad = s_bad != 0u;
        if (bad) {                                               // the plain scatter for this tile
#pragma unroll
            for (uint32_t q = 0; q < RPT; ++q) {
                if (e[q].code == kHoleCode) continue;
                const uint64_t dst = static_cast<uint64_t>(off[e[q].u - d.first_user]) + e[q].t;
                if (dst < out_cap) out[dst] = e[q];
            }
            __syncthreads();
            continue;
        }
        {
            uint32_t run = s_scan[threadIdx.x] - mine;
            for (uint32_t i = threadIdx.x * (kSortHash / kBlock); i < (threadIdx.x + 1) * (kSortHash / kBlock); ++i) {
                h_base[i] = run;
                if (h_cnt[i]) h_dst[i] = static_cast<unsigned long long>(off[h_key[i] - d.first_user]) + h_tmin[i];
                run += h_cnt[i];
            }
        }
        const uint32_t total = s_scan[kBlock - 1];
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < RPT; ++q) {
            if (e[q].code == kHoleCode) continue;
            const uint32_t pos = h_base[slot[q]] + (e[q].t - h_tmin[slot[q]]);
            s_rows[pos] = e[q];
            s_slot[pos] = static_cast<unsigned short>(slot[q]);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < total; i += kBlock) {
            const uint32_t sl = s_slot[i];
            const uint64_t dst = h_dst[sl] + (i - h_base[sl]);
            if (dst < out_cap) out[dst] = s_rows[i];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kBlock) k_scatter_phantom(DevSim d, const int64_t* off, rg_event* out,
                                                          uint64_t out_cap) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        if (!d.has_phantom[i]) continue;
        const uint64_t dst = static_cast<uint64_t>(off[i]) + d.n_events[i];
        if (dst < out_cap) out[dst] = d.phantom[i];
    }
}

// the float64 side arrays in the same order: NaN where the reference's column is NaN (organic rows; p_click of
// the phantom row, which is never drawn)
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

__global__ void __launch_bounds__(kBlock) k_scatter_time(DevSim d, uint64_t n_rows, const int64_t* off, double* out, uint64_t out_cap) {
    for (uint64_t r = blockIdx.x * static_cast<uint64_t>(kBlock) + threadIdx.x; r < n_rows;
         r += static_cast<uint64_t>(gridDim.x) * kBlock) {
        const rg_event e = d.log[r];
        if (e.code == kHoleCode) continue;
        const uint64_t dst = static_cast<uint64_t>(off[e.u - d.first_user]) + e.t;
        if (dst < out_cap) out[dst] = d.aux_time ? d.aux_time[r] : static_cast<double>(e.t);
    }
}

__global__ void __launch_bounds__(kBlock) k_scatter_time_phantom(DevSim d, const int64_t* off, double* out, uint64_t out_cap) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock) {
        if (!d.has_phantom[i]) continue;
        const uint64_t dst = static_cast<uint64_t>(off[i]) + d.n_events[i];
        if (dst < out_cap) out[dst] = d.time_mode ? d.phantom_time[i] : static_cast<double>(d.n_events[i]);
    }
}

__global__ void __launch_bounds__(kBlock) k_export_time(DevSim d, uint32_t t, double* out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < d.n_users; i += gridDim.x * kBlock)
        out[i] = d.time_mode ? d.utime[i] : static_cast<double>(d.n_events[i] ? d.n_events[i] : t);
}

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
    if (d.env_kind) {          // reco-gym-v0: a table look-up per organic user
        hipLaunchKernelGGL(k_draw_env0, dim3(grid_for(upper)), dim3(kBlock), 0, st, d, t);
        if (int rc = prof_mark(sim, st)) return rc;
        if (int rc = prof_mark(sim, st)) return rc;
    } else if (d.use_mfma == 2 && d.use_cache && t > 0) {
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
        const bool tp = S == 1 && sim->tp_kernel && sim->sweep_lds && !d.use_cache;
        if (tp) hipLaunchKernelGGL(sim->tp_kernel, dim3(grid), dim3(kBlock), sim->tp_smem, st, d, t, sim->tp_nts);
        else
        hipLaunchKernelGGL(sim->bf16_kernel, dim3(grid), dim3(sim->draw_threads), sim->bf16_smem, st, d, t, S);
        if (int rc = prof_mark(sim, st)) return rc;
        if (tp) {
            // k_pick: a wave per 32 draws of one tile (the sweep listed the draws by tile), grid-stride over the groups
            const size_t psmem = (2 * kTpBins * kTpShards + 8) * sizeof(uint32_t) + 4 * 32 * 2 * static_cast<size_t>(d.KH) * sizeof(float);
            if (psmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sim->pick_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(psmem));
            uint32_t pgrid = upper / 128u + kTpLists / 4u + 1u;      // groups / 4 waves (a part-filled group per (tile, shard))
            const uint32_t pcap = static_cast<uint32_t>(device_cus(sim)) * 4u;
            if (pgrid > pcap) pgrid = pcap;
            hipLaunchKernelGGL(sim->pick_kernel, dim3(pgrid), dim3(kBlock), psmem, st, d, t, 0u);
        }
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
        if (d.lr_sample)           // select_randomly: a softmax and a draw per act (a wave each)
            hipLaunchKernelGGL(logreg_sample_kernel(), dim3(grid_for(static_cast<uint64_t>(upper) + 64, kBlock / 64)), dim3(kBlock), 0, st, d, t);
        else if (d.lr_coef16_t) {       // screen (a wave per act and class range), then decide (a wave per act)
            // (the number of acts is only known on the device — at most a quarter of the live users, usually a fiftieth: the kernels
            // walk their lists grid-stride, and the grids are capped at a few blocks per CU.  Sized for the worst case they were
            // mostly blocks that start and leave — and k_logreg_decide's three counter atomics per WAVE, one act each, were all of
            // its time: 41 of the 169 ms of C5's acts, profiles/r6/c5_fp16_kernel_stats_call31.csv)
            const int cus = device_cus(sim);
            auto capped = [&](uint64_t waves, int blocks_per_cu) {
                const int g = grid_for(waves, kBlock / 64);
                return g < cus * blocks_per_cu ? g : cus * blocks_per_cu;
            };
            hipLaunchKernelGGL(logreg_screen_kernel(d.lr_coef8_t != nullptr), dim3(capped((static_cast<uint64_t>(upper) / 4 + 64) * kLrSplit, 16)),
                               dim3(kBlock), 0, st, d, t);
            hipLaunchKernelGGL(logreg_decide_kernel(), dim3(capped(static_cast<uint64_t>(upper) / 4 + 64, 8)), dim3(kBlock), 0, st, d, t);
            if (upper > d.lr_part_cap)     // the step may list more acts than the screen's scratch has rows: the rest in fp32 / float64
                hipLaunchKernelGGL(logreg_acts_kernel(), dim3(capped(static_cast<uint64_t>(upper - d.lr_part_cap) / 4 + 64, 8)), dim3(kBlock), 0, st, d, t);
        } else {
            const int g = grid_for(static_cast<uint64_t>(upper) / 4 + 64, kBlock / 64), cap = device_cus(sim) * 8;
            hipLaunchKernelGGL(logreg_acts_kernel(), dim3(g < cap ? g : cap), dim3(kBlock), 0, st, d, t);
        }
    }
    if (int rc = prof_mark(sim, st)) return rc;
    // 2. click draws, transitions, drift, next lists, bandit + phantom rows
    if (d.run_ahead)       // a round: every listed user through its bandit run (k_advance_run), then the round's raw-log books
        hipLaunchKernelGGL(advance_run_kernel(), dim3(grid_for(upper, kAdvBlock)), dim3(kAdvBlock), 0, st, d, t, d.run_ahead);
    else
        hipLaunchKernelGGL(advance_kernel(), dim3(grid_for(upper, kAdvBlock)), dim3(kAdvBlock), 0, st, d, t, d_actions);
    if (d.sigma_omega != 0.0)
        hipLaunchKernelGGL(drift_kernel(), dim3(grid_for(static_cast<uint64_t>(upper) * ((d.K + 1) / 2))), dim3(kBlock), 0, st, d, t);
    if (d.run_ahead) hipLaunchKernelGGL(round_rows_kernel(), dim3(1), dim3(1), 0, st, d, t, 0u);
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
    if (d.debug_row_base)      // test hook: the walk reserves its raw rows from this counter
        HIP_TRY(hipMemcpyAsync(d.counters + kCntTailRows, &sim->d.debug_row_base, sizeof(unsigned long long), hipMemcpyHostToDevice, st));
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
    if (d.debug_row_base)      // test hook: the walk reserves its raw rows from this counter
        HIP_TRY(hipMemcpyAsync(d.counters + kCntTailRows, &sim->d.debug_row_base, sizeof(unsigned long long), hipMemcpyHostToDevice, st));
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
            if (sim->xh_kernel) hipLaunchKernelGGL(sim->xh_kernel, dim3(grid_for((dg.grp_n + 32u * sim->xh_waves - 1) / (32u * sim->xh_waves), 1)), dim3(64 * sim->xh_waves), sim->xh_smem, sS, ds, 0u, 1u);
            else
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


}  // namespace rgk

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
    d.env_kind = cfg->env_kind;
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
        s->bf16_smem = bf16_smem_bytes(geom_of(*cfg), 2 * d.KH, s->bf16_kernel == bf16p_kernel_for(d) ? static_cast<uint32_t>(RG_SWEEP_NB) : 2u);
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
    // the walked run's sweep with an error-free leading accumulator (k_sweep_xh) where the pipelined fp16 sweep would run it
    s->xh_kernel = nullptr; s->xh_smem = 0;
    if (d.XNH && d.use_cache && d.use_mfma == 2 && s->bf16_kernel && s->bf16_kernel == bf16p_kernel_for(d) && d.f16 && !d.wide &&
        static_cast<size_t>(d.P_pad) * d.XRS < (1ull << 31))
    {
        s->xh_waves = 4;
        if (const char* e = getenv("RECOGYM_XH_WAVES")) s->xh_waves = atoi(e) == 8 ? 8 : 4;
        s->xh_kernel = xh_kernel_for(d, s->xh_waves);
    }
    if (s->xh_kernel) {
        s->xh_smem = 2 * (128 * static_cast<size_t>(d.XRS) + 512) + 256 + static_cast<size_t>(s->xh_waves) * 32 * kMaxSC * sizeof(float);   // tiles, seeds, the super-chunk prefix stage
        if (const char* e = getenv("RECOGYM_XH_SMEM_PAD")) s->xh_smem += static_cast<size_t>(atoi(e));   // occupancy experiments: one block per CU
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(s->xh_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(s->xh_smem));
    } else { d.XNH = d.XNL = d.XRS = 0; }
    // the sweep whose search stays in LDS (k_draw_tp): where every draw sweeps (no per-user cache) and the two-way fp16 split of
    // K <= 20 serves the table; a user's tile prefixes must fit beside the tiles (P <= ~12 000 at two blocks per CU)
    s->tp_kernel = nullptr; s->pick_kernel = nullptr; s->tp_smem = 0; s->tp_nts = 0; s->sweep_lds = 1;
    if (const char* e = getenv("RECOGYM_SWEEP_LDS")) s->sweep_lds = e[0] != '0';
    if (d.use_mfma == 2 && !d.use_cache && s->bf16_kernel && s->bf16_kernel == bf16p_kernel_for(d) && d.f16 && !d.wide) {
        draw_kernel_t kt = tp_kernel_for(d), kp = pick_kernel_for(d);
        if (kt && kp && d.tp_rec) {
            const uint32_t nts = ((d.n_chunks / 4) + 3u) & ~3u;
            // (-DRG_TP_SMEM_PAD=bytes: a timing build that pushes the kernel to ONE block per CU — what the second block is worth)
            const size_t smem = 2 * (128 * static_cast<size_t>(d.RS) + 512) + 256 + 4 * 32 * static_cast<size_t>(nts) * sizeof(float) + RG_TP_SMEM_PAD;
            // (<= 128 tile lists; and >= 4 tiles: below that — P <= 384 — the sweep is a few hundred MFMAs per draw and the per-draw
            // list atomic on <= 3 x 128 counters would pace it: those tables keep k_draw_bf16p)
            if (smem <= 160 * 1024 && d.n_chunks / 4 <= 128u && d.n_chunks / 4 >= 4u) {
                s->tp_kernel = kt; s->pick_kernel = kp; s->tp_smem = smem; s->tp_nts = nts;
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
            }
        }
    }
    if (d.use_mfma == 2 && !d.use_cache && d.wide && s->bf16_kernel && s->bf16_kernel == f16w_kernel_for(d) && d.tp_rec) {
        // wide K: k_draw_tpw keeps a prefix per SUPER-TILE of G 64-product tiles — what fits beside its three tile buffers
        draw_kernel_t kt = tpw_kernel_for(d), kp = pick_kernel_for(d);
        const size_t tile_b = 64 * static_cast<size_t>(d.RS);
        const size_t avail = 160 * 1024 - 3 * tile_b - 1024;
        uint32_t nts_max = static_cast<uint32_t>(avail / (256 * sizeof(float))) & ~3u;
        if (nts_max > kTpBins) nts_max = kTpBins;
        if (kt && kp && nts_max >= 8) {
            const uint32_t n_t = d.n_chunks / 2;
            const uint32_t G = (n_t + nts_max - 1) / nts_max;
            const uint32_t n_s = (n_t + G - 1) / G;
            uint32_t nts = (n_s + 3u) & ~3u;
            // (the omega32 stage of the block's 256 users lies over tile buffers 1, 2 and the prefix rows)
            while (2 * tile_b + 256 * static_cast<size_t>(nts) * sizeof(float) < 256 * 2 * static_cast<size_t>(d.KH) * sizeof(float)) nts += 4;
            const size_t smem = 3 * tile_b + 1024 + 256 * static_cast<size_t>(nts) * sizeof(float);
            if (smem <= 160 * 1024) {
                s->tp_kernel = kt; s->pick_kernel = kp; s->tp_smem = smem; s->tp_nts = nts;
                d.tp_cpt = 2 * G;
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
            }
        }
    }
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
        // ... and not at all beyond P x K = 10^6 there: a block streams the whole float64 table per draw (51 MB at 10^5 x 64: 0.4 ms,
        // as long as a sliced round of ALL the users left).  A C4 shard with the tail from 128 users 1 615 ms, from 32 1 582, from 2
        // 1 538, never 1 503 (profiles/r6/ab_call45_c4_tail.jsonl, ab_call46_c4_tail.jsonl); at 10^4 x 20 it makes no difference
        // (c3drift 801 / 795 / 801 ms from 4 096 / 1 024 / never: ab_call47_tail.jsonl)
        if (d.use_mfma == 2 && static_cast<double>(d.P) * static_cast<double>(d.K) > 1.0e6) s->tail_below = 0;
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
    s->handover_auto = true;
    if (const char* e = getenv("RECOGYM_WALK_HANDOVER")) { d.walk_handover = static_cast<uint32_t>(atoi(e)); s->handover_auto = false; }
    d.walk_click_batch = 8;
    if (const char* e = getenv("RECOGYM_WALK_CLICK_BATCH")) d.walk_click_batch = static_cast<uint32_t>(atoi(e));
    d.walk_search_batch = 16;
    d.walk_helpers = kWalkHelpersMax;
    d.walk_click_join = 1;
    if (const char* e = getenv("RECOGYM_WALK_CLICK_JOIN")) d.walk_click_join = e[0] != '0';
    if (const char* e = getenv("RECOGYM_WALK_HELPERS")) d.walk_helpers = static_cast<uint32_t>(atoi(e)) > kWalkHelpersMax ? kWalkHelpersMax : static_cast<uint32_t>(atoi(e));
    if (const char* e = getenv("RECOGYM_WALK_SEARCH_BATCH")) d.walk_search_batch = static_cast<uint32_t>(atoi(e)) ? static_cast<uint32_t>(atoi(e)) : 1u;
    if (const char* e = getenv("RECOGYM_WALK_REFILL")) d.walk_refill = static_cast<uint32_t>(atoi(e));
    if (const char* e = getenv("RECOGYM_WALK_BIAS")) d.walk_bias = static_cast<uint32_t>(atoi(e));
    // k_walk2 where it is instantiated for the configuration (RECOGYM_WALK=1: k_walk), four blocks per CU at K <= 20
    s->walk2 = s->walk && walk2_kernel_for(d, 4) != nullptr;
    if (const char* e = getenv("RECOGYM_WALK")) if (e[0] == '1') s->walk2 = false;
    // (k_walk2: 4 since the act is a count on the compact history line — the bandit iteration got shorter, so the organic kind
    // waits for more lanes: profiles/r4/ab_call6_walk_bias.jsonl; k_walk keeps round 2's 8)
    // (round 5, with helpers: a bandit iteration is full whatever the number of bandit lanes, so the organic kinds wait for
    // more of theirs — bias 2, search batch 24: profiles/r5/ab_call18_walk_tuning.jsonl)
    if (s->walk2 && !getenv("RECOGYM_WALK_BIAS")) d.walk_bias = d.walk_helpers ? 2 : 4;
    if (s->walk2 && !getenv("RECOGYM_WALK_SEARCH_BATCH") && d.walk_helpers) d.walk_search_batch = 24;
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
    if (const char* e = getenv("RECOGYM_TAIL"))        // (never for the frozen LogReg policy: k_tail acts on one thread — 46 s per C5 step)
        if (d.policy != RG_POLICY_LOGREG_FROZEN) s->tail_below = static_cast<uint32_t>(atoi(e));
    if (const char* e = getenv("RECOGYM_REPACK")) s->repack_every = static_cast<uint32_t>(atoi(e));
    s->run_ahead = 32;         // events a round of a run to the end may take a user through (0: lock-step, an event per launch)
    if (const char* e = getenv("RECOGYM_RUN_AHEAD")) s->run_ahead = static_cast<uint32_t>(atoi(e));
    s->lr_part_rows = d.lr_part_cap;
    if (const char* e = getenv("RECOGYM_LR_PART_CAP")) { const uint32_t v = static_cast<uint32_t>(atoi(e)); if (v < d.lr_part_cap) d.lr_part_cap = v; }
    if (const char* e = getenv("RECOGYM_ABLATE")) d.ablate = static_cast<uint32_t>(atoi(e));
    if (d.lr_sample) s->run_ahead = 0;     // a sampled act per event: lock-step (k_advance_run reuses one act for a whole bandit run)
    if (d.env_kind) {
        // reco-gym-v0: no omega, no product sweep — the lock-step loop with its own draw kernel (k_draw_env0), nothing else
        d.use_mfma = 0; d.use_cache = 0; d.XNH = d.XNL = d.XRS = 0;
        s->bf16_kernel = nullptr; s->xh_kernel = nullptr;
        s->walk = s->walk2 = false;
        s->tail_below = 0;
        s->run_ahead = 0;
    }
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
    if (!strcmp(n, "walk_helpers")) return &s->d.walk_helpers;
    if (!strcmp(n, "walk_click_join")) return &s->d.walk_click_join;
    if (!strcmp(n, "pipe_min_users")) return &s->pipe_min_users;
    if (!strcmp(n, "tail_below")) return &s->tail_below;
    if (!strcmp(n, "repack_every")) return &s->repack_every;
    if (!strcmp(n, "run_ahead")) return &s->run_ahead;
    if (!strcmp(n, "sweep_lds")) return &s->sweep_lds;
    if (!strcmp(n, "lr_part_cap")) return &s->d.lr_part_cap;
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
        if (value < -2147483647 || value > 2147483647) return fail(RG_EINVAL, "%s out of the range of an int", name);
        if (!strcmp(name, "pipe_groups") && value < 0) return fail(RG_EINVAL, "pipe_groups must be >= 0");
        if (!strcmp(name, "pipe_mode") && (value < 0 || value > 2)) return fail(RG_EINVAL, "pipe_mode must be 0, 1 or 2");
        if (!strcmp(name, "slices") && value < -1) return fail(RG_EINVAL, "slices must be >= -1 (-1 = by population)");
        if ((!strcmp(name, "exact_tile") || !strcmp(name, "resident_grid") || !strcmp(name, "sweep_prefix_off") || !strcmp(name, "debug")) &&
            (value < 0 || value > 1)) return fail(RG_EINVAL, "%s is a flag (0 or 1)", name);
        *p = static_cast<int>(value);
        return RG_OK;
    }
    if (uint32_t* p = opt_u32(sim, name)) {
        if (value < 0 || value > 0xFFFFFFFFll) return fail(RG_EINVAL, "%s must be in [0, 2^32)", name);
        if (!strcmp(name, "run_ahead") && value > 64) return fail(RG_EINVAL, "run_ahead must be <= 64 events");
        if (!strcmp(name, "run_ahead") && value && sim->d.env_kind) return fail(RG_EINVAL, "env_kind 1 (reco-gym-v0) runs lock-step (run_ahead = 0)");
        if (!strcmp(name, "walk_search_batch") && value < 1) value = 1;
        if (!strcmp(name, "walk_handover")) sim->handover_auto = false;
        if (!strcmp(name, "walk_click_join") && value > 1) return fail(RG_EINVAL, "walk_click_join is a flag (0 or 1)");
        if (!strcmp(name, "walk_helpers") && value > kWalkHelpersMax) return fail(RG_EINVAL, "walk_helpers must be in [0, %u]", kWalkHelpersMax);
        if (!strcmp(name, "pipe_min_users") && value < 256) return fail(RG_EINVAL, "pipe_min_users must be >= 256");
        if (!strcmp(name, "lr_part_cap") && static_cast<uint64_t>(value) > sim->lr_part_rows)
            return fail(RG_EINVAL, "lr_part_cap can only be lowered (the workspace holds %u rows)", sim->lr_part_rows);
        *p = static_cast<uint32_t>(value);
        return RG_OK;
    }
    return fail(RG_EINVAL, "unknown option '%s'", name);
}

int rg_sim_get_option(rg_sim* sim, const char* name, int64_t* value) {
    if (!sim || !name || !value) return fail(RG_EINVAL, "NULL argument");
    if (!strcmp(name, "sweep_lds_kernel")) { *value = sim->tp_kernel ? 1 : 0; return RG_OK; }     // read-only: k_draw_tp serves the configuration
    if (int* p = opt_int(sim, name)) { *value = *p; return RG_OK; }
    if (uint32_t* p = opt_u32(sim, name)) { *value = *p; return RG_OK; }
    return fail(RG_EINVAL, "unknown option '%s'", name);
}

int rg_sim_set_tables(rg_sim* sim, const double* d_gamma, const double* d_mu_organic,
                      const double* d_beta, const double* d_mu_bandit, void* stream) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (!d_gamma || !d_mu_organic || !d_beta || !d_mu_bandit) return fail(RG_EINVAL, "table pointer is NULL");
    if (sim->d.env_kind) return fail(RG_ESTATE, "env_kind 1 (reco-gym-v0) takes rg_sim_set_env0_tables");
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
        if (sim->d.XNH) {
            hipLaunchKernelGGL(xh_table_kernel(), dim3(grid_for(static_cast<size_t>(sim->d.P_pad) * (sim->d.XRS / 2))),
                               dim3(kBlock), 0, static_cast<hipStream_t>(stream), sim->d);
            hipLaunchKernelGGL(xh_stats_kernel(), dim3(2 * sim->d.KH + 2), dim3(kBlock), 0, static_cast<hipStream_t>(stream), sim->d);
        }
    }
    HIP_TRY(hipGetLastError());
    sim->tables_set = true;
    return RG_OK;
}

int rg_env0_click_thresholds(const double* p, uint64_t n, double* qn, double* px1) {
    if (!p || !qn || !px1) return fail(RG_EINVAL, "NULL argument");
    for (uint64_t i = 0; i < n; ++i) {
        // random_binomial_inversion(n = 1, p'): q = 1 - p', qn = exp(n log q), second step px = ((n - X + 1) p' px) / (X q) at X = 1
        const double pe = p[i] <= 0.5 ? p[i] : 1.0 - p[i];
        const double q = 1.0 - pe;
        const double r = exp(1.0 * log(q));
        qn[i] = r;
        px1[i] = (1.0 * pe * r) / (1.0 * q);
    }
    return RG_OK;
}

int rg_sim_set_env0_tables(rg_sim* sim, const double* d_cdf_init, const double* d_cdf_cluster, uint32_t cluster_size,
                           const double* d_click_p, const double* d_click_qn, const double* d_click_px1, void* stream) {
    (void)stream;
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (!sim->d.env_kind) return fail(RG_ESTATE, "env_kind is not 1 (reco-gym-v0)");
    if (!d_cdf_init || !d_cdf_cluster || !d_click_p || !d_click_qn || !d_click_px1) return fail(RG_EINVAL, "table pointer is NULL");
    if (cluster_size == 0 || cluster_size > sim->d.P) return fail(RG_EINVAL, "cluster_size %u out of range", cluster_size);
    if (rg_device_count() <= 0) return fail(RG_ENODEV, "no HIP device");
    sim->d.e0_cdf_init = d_cdf_init; sim->d.e0_cdf_cluster = d_cdf_cluster; sim->d.e0_cluster = cluster_size;
    sim->d.e0_click_p = d_click_p; sim->d.e0_click_qn = d_click_qn; sim->d.e0_click_px1 = d_click_px1;
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
    if (sim->d.lr_sample && n_classes != sim->d.P)
        return fail(RG_EINVAL, "lr_select_randomly samples a PRODUCT from predict_proba: the model needs a class per product (%u classes, %u products)", n_classes, sim->d.P);
    sim->d.lr_coef_t = d_coef_t; sim->d.lr_intercept = d_intercept; sim->d.lr_classes = d_classes;
    sim->d.lr_n = n_classes;
    // a new model invalidates the optional copies of the old one (their shapes and bounds belong to it): set them again
    sim->d.lr_coef32_t = nullptr; sim->d.lr_intercept32 = nullptr; sim->d.lr_wmax = nullptr; sim->d.lr_bmax = 0.0f;
    sim->d.lr_coef16_t = nullptr; sim->d.lr_coef8_t = nullptr; sim->d.lr_scale8 = nullptr;
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
    sim->d.lr_coef8_t = nullptr; sim->d.lr_scale8 = nullptr;
    return RG_OK;
}

int rg_sim_set_logreg_fp16(rg_sim* sim, const uint16_t* d_coef16_t) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (sim->d.policy != RG_POLICY_LOGREG_FROZEN) return fail(RG_ESTATE, "policy is not RG_POLICY_LOGREG_FROZEN");
    if (d_coef16_t && !sim->d.lr_coef32_t) return fail(RG_ESTATE, "rg_sim_set_logreg_fp32 must be called first (intercept32, wmax, bmax)");
    if (d_coef16_t && sim->d.lr_n % 8u) return fail(RG_EINVAL, "the fp16 screening pass needs n_classes %% 8 == 0 (have %u)", sim->d.lr_n);
    sim->d.lr_coef16_t = d_coef16_t;
    sim->d.lr_coef8_t = nullptr; sim->d.lr_scale8 = nullptr;
    return RG_OK;
}

int rg_sim_set_logreg_int8(rg_sim* sim, const uint8_t* d_coef8_t, const float* d_scale8) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    if (sim->d.policy != RG_POLICY_LOGREG_FROZEN) return fail(RG_ESTATE, "policy is not RG_POLICY_LOGREG_FROZEN");
    if ((d_coef8_t != nullptr) != (d_scale8 != nullptr)) return fail(RG_EINVAL, "both arrays or none");
    if (d_coef8_t && !sim->d.lr_coef16_t) return fail(RG_ESTATE, "rg_sim_set_logreg_fp16 must be called first (the 8-bit copy replaces the rows the screening pass reads)");
    sim->d.lr_coef8_t = d_coef8_t; sim->d.lr_scale8 = d_scale8;
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
    d.run_ahead = 0;                           // rg_sim_run turns the rounds on for a run to the end
    // live lanes at which a draining wave hands its users to the next round: what it hands over also goes through the float64
    // batch — a sweep of P x K per user — while a later hand-over lengthens the round's drain.  16 below 2 M users where a
    // float64 sweep is expensive (P x K >= 10^5: a 1.25 M-user shard of C3 21.3 -> 20.0 ms, profiles/r6/ab_call19_handover.jsonl),
    // else 32 (C2, P x K = 2 10^4, 1 M users: 8.1 ms at 32, 8.9 at 16: profiles/r6/c3_bench_line_call22.json against _call9)
    if (sim->handover_auto) d.walk_handover = (n < 2000000ull && static_cast<uint64_t>(d.P) * d.K >= 100000ull) ? 16u : 32u;
    HIP_TRY(hipMemsetAsync(d.step_cnt, 0, sizeof(uint32_t) * 2 * (kMaxSteps + 2), st));
    HIP_TRY(hipMemsetAsync(d.exact_cnt, 0, sizeof(uint32_t) * (kMaxSteps + 2), st));
    HIP_TRY(hipMemsetAsync(d.exact_cnt_b, 0, sizeof(uint32_t) * (kMaxSteps + 2), st));
    if (d.lr_dirty) HIP_TRY(hipMemsetAsync(d.lr_cnt, 0, sizeof(uint32_t) * (kMaxSteps + 2), st));
    if (d.sigma_omega != 0.0) HIP_TRY(hipMemsetAsync(d.drift_cnt, 0, sizeof(uint32_t) * (kMaxSteps + 2), st));
    HIP_TRY(hipMemsetAsync(d.counters, 0, sizeof(unsigned long long) * RG_CNT_N, st));
    HIP_TRY(hipMemsetAsync(d.tp_hist, 0, sizeof(uint32_t) * (kTpBins * kTpShards + 4), st));
    if (d.debug_row_base && d.log) {       // test hook: the raw log starts at that row; what lies below is marked unused
        if (d.debug_row_base > d.log_cap) return fail(RG_EINVAL, "debug row base %llu beyond the log capacity %llu",
                                                      (unsigned long long)d.debug_row_base, (unsigned long long)d.log_cap);
        HIP_TRY(hipMemsetAsync(d.log, 0xFF, sizeof(rg_event) * d.debug_row_base, st));
    }
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
    // (-1 = no action: a user in the organic state ignores it; the kernel never indexes beta / mu_b with an action outside [0, P))
    if (action < -1 || action >= static_cast<int32_t>(sim->d.P)) return fail(RG_EINVAL, "action %d is outside [0, %u) (-1: none)", action, sim->d.P);
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
    // To the end from a fresh reset, nothing that moves omega inside a bandit run: run-ahead rounds (k_advance_run) — a user's whole
    // bandit run per round instead of a lock-step launch per event.  Everything else (partial runs, the step API, the sum cache of a
    // sigma_omega == 0 run without the walk) stays lock-step: event index == step number.
    if (sim->t == 0 && max_steps >= kMaxSteps && sim->run_ahead && !sim->d.change_omega_for_bandits && !sim->d.use_cache && !sim->d.u_override) {
        sim->d.run_ahead = sim->run_ahead;
        hipLaunchKernelGGL(round_rows_kernel(), dim3(1), dim3(1), 0, st, sim->d, 0u, 1u);
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

int rg_sim_debug_set_row_base(rg_sim* sim, uint64_t rows) {
    if (!sim) return fail(RG_EINVAL, "sim is NULL");
    sim->d.debug_row_base = rows;
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
    {
        static const bool plain = getenv("RECOGYM_SORT_PLAIN") != nullptr;      // (A/B: the scatter without the LDS tiles)
        const uint64_t tiles = (n_rows + kSortTile - 1) / kSortTile;
        const uint64_t cap = static_cast<uint64_t>(device_cus(sim)) * 12u;
        if (plain) hipLaunchKernelGGL(k_scatter_rows, dim3(grid_for(n_rows)), dim3(kBlock), 0, st, d, n_rows, d_row_offsets, d_sorted, sorted_capacity);
        else hipLaunchKernelGGL(k_scatter_rows_tiled, dim3(static_cast<unsigned>(tiles < cap ? (tiles ? tiles : 1) : cap)), dim3(kBlock), 0, st, d, n_rows,
                                d_row_offsets, d_sorted, sorted_capacity);
    }
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
