// rg_exact.hip — librecogym_hip.so, unit 2 of 7: the float64 resolve: k_exact_sums (K > 64), k_exact_sums_m (float64 matrix cores), k_exact_sums_h (the walk's batch), k_exact_ref, k_exact_pick.
// (see rg_common.hpp for the shared types and helpers, DESIGN.md for the data layout and the rooflines)

#include "rg_common.hpp"

namespace rgk {

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

// pure float64 mode: the reference of every user = its max logit (reco_env_v1.py:121)
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

}  // namespace rgk
