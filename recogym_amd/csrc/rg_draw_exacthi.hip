// rg_draw_exacthi.hip — librecogym_hip.so, unit 8 of 8: k_sweep_xh, the sigma_omega = 0 sweep whose leading part is ERROR-FREE.
// (see rg_common.hpp for the shared types and helpers, DESIGN.md §2 "round 5" for the derivation and the measured effect)
//
// What it replaces: k_draw_bf16p<.., F16> in its prefix form (sweep_only = 2) — every user's product sweep at t = 0, whose
// per-chunk exp-sums the user-major walk searches (reference: RecoEnv1.update_product_view, reco_env_v1.py:119-128).  The
// certificate's delta of that kernel is dominated by what it must ASSUME about the matrix pipe's fp32 accumulation:
// (K + 5) 2^-24 Ahat for K + 2 large terms added in an order the hardware chooses, + 12 x 2^-24 Ahat for the floating two-way
// split.  Here the logit  l = (mu_p + Gamma_p . omega) log2 e - q  is formed as  H + 2^-9 L  from TWO accumulators:
//
//   H (exact):   sum_k Ghi_pk whi_k + m1_p + m2_p - q, every operand a FIXED-POINT fp16 value: Ghi = Gamma' rounded to the grid
//                2^-8, whi = omega rounded to 2^-8, m1 + m2 = mu' rounded to 2^-16, q an integer.  Every product is a multiple
//                of 2^-16 and every partial sum of any subset of the terms, in any order, is below 2^24 x 2^-16 = 256 in
//                magnitude (checked per user: xh_eligible) — so every intermediate value is representable in fp32 and NO
//                rounding happens in the accumulator, whatever the adder tree inside the MFMA looks like.
//   L (small):   the residuals, scaled by 2^9: Ghi wmid + Ghi wlo + Glo whi + Glo wmid + (mu' - m1 - m2), 4K + 1 terms of
//                magnitude <= 2^-9 (sum |Gamma'| + sum |omega|) — their roundings are 2^-24 of THAT (E_lo below).
//
// One fp32 rounding (the fused multiply-add that joins them) and the exp's own ulp are what is left of the arithmetic error;
// the representation error is that of Gamma' = Ghi + Glo (measured per table column: xstats) — delta ~ 1.2e-5 at BASELINE
// config 3 against 1.1e-4.  Scales (2^9, 2^-6 x 2^15, 2^3 x 2^6) keep every fp16 piece in the NORMAL range, so nothing depends on
// how the matrix pipe treats fp16 subnormals.
//
//   A row (table, k_make_xh_table): [Ghi(K) 0.. m1 m2' 1 | Ghi(K) Ghi 2^-6(K) Glo 2^9(K) Glo 2^3(K) 0..]   (16 NH | 16 NL slots)
//   B row (per user, registers):    [whi(K) 0.. 1  2^-10 -q | wmid 2^9(K) wlo 2^15(K) whi(K) wmid 2^6(K) 0..]
//   seeds:                          H <- 0,  L <- 2^9 (mu' - m1 - m2)  (xmulo, -inf beyond P)
//
// Shape: a wave = 32 users x all P products in chunks of 32; one chunk's NH + NL MFMAs (v_mfma_f32_32x32x16_f16) are issued
// while the previous chunk's join + exp + sum run on the vector ALU and the next chunk's operand rows are read from the LDS
// tile; tiles of 128 products by buffer_load ... lds DMA, one tile ahead (two buffers), one barrier per tile.

#include "rg_common.hpp"
#include <type_traits>

// -DRG_XH_ABL=bits: timing experiments (results wrong by design; A/B builds only, loaded with RECOGYM_HIP_LIB).  1: no residual
// MFMAs beyond the first, 2: exps replaced by a move, 4: the next chunk's operand rows are not re-read, 8: no seed reads,
// 16: no tile DMA / barrier, 32: no scheduling pins, 64: no bookkeeping, 128: books without their stores, 1024: no super-chunk records
#ifndef RG_XH_ABL
#define RG_XH_ABL 0
#endif
#if (RG_XH_ABL & 32)
#define RG_XPIN() do {} while (0)
#else
#define RG_XPIN() RG_PIN()
#endif

namespace rgk {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x2v = __attribute__((ext_vector_type(2))) float;

struct XhPieces { _Float16 hi, mid9, lo15, mid6; double r1; };

// omega_k = hi + mid9 / 2^9 + lo15 / 2^15 up to ~2^-32: hi on the grid 2^-8 (an fp16 value: coarser above 8, still on the grid)
__device__ __forceinline__ XhPieces xh_split(double w) {
    XhPieces p;
    p.hi = static_cast<_Float16>(static_cast<float>(rint(w * 256.0) * 0.00390625));
    p.r1 = w - static_cast<double>(static_cast<float>(p.hi));
    p.mid9 = static_cast<_Float16>(static_cast<float>(p.r1 * 512.0));
    p.mid6 = static_cast<_Float16>(static_cast<float>(p.mid9) * 0.125f);
    const double r2 = p.r1 - static_cast<double>(static_cast<float>(p.mid9)) * 0.001953125;
    p.lo15 = static_cast<_Float16>(static_cast<float>(r2 * 32768.0));
    return p;
}

// The table: A rows of the two MFMA groups and the seed of the residual accumulator (see the header).  Also xstats:
//   [k < 2 KH]   max_p |Gamma'_pk - Ghi_pk - Glo_pk|   (log2 units, rounded up; block k)
//   [2 KH]       max_pk |Glo_pk|        [2 KH + 1]   max_p |mu'_p - m1_p - m2_p|
__global__ void __launch_bounds__(kBlock) k_make_xh_table(DevSim d) {
    const double log2e = 1.4426950408889634074;
    const size_t rs2 = d.XRS / 2;
    const size_t n = static_cast<size_t>(d.P_pad) * rs2;
    const uint32_t nh16 = 16u * d.XNH;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t p = i / rs2;
        const uint32_t ke = static_cast<uint32_t>(i % rs2);
        _Float16 v = static_cast<_Float16>(0.0f);
        if (p < d.P) {
            const double mu = d.mu_o[p] * log2e;
            const _Float16 m1 = static_cast<_Float16>(static_cast<float>(rint(mu * 32.0) * 0.03125));
            const double r = mu - static_cast<double>(static_cast<float>(m1));
            // m2 on the grid 2^-16, carried as m2 2^10 (the B row holds 2^-10): a normal fp16 value
            const _Float16 m2s = static_cast<_Float16>(static_cast<float>(rint(r * 65536.0) * 0.015625));
            if (ke < nh16) {
                if (ke < d.K) v = static_cast<_Float16>(static_cast<float>(rint(d.gamma[p * d.K + ke] * log2e * 256.0) * 0.00390625));
                else if (ke == nh16 - 3) v = m1;
                else if (ke == nh16 - 2) v = m2s;
                else if (ke == nh16 - 1) v = static_cast<_Float16>(1.0f);
            } else if (ke < 16u * (d.XNH + d.XNL)) {
                const uint32_t s = ke - nh16, grp = s / d.K, k = s % d.K;
                if (grp < 4) {
                    const double g = d.gamma[p * d.K + k] * log2e;
                    const _Float16 ghi = static_cast<_Float16>(static_cast<float>(rint(g * 256.0) * 0.00390625));
                    const double rg = g - static_cast<double>(static_cast<float>(ghi));
                    if (grp == 0) v = ghi;
                    else if (grp == 1) v = static_cast<_Float16>(static_cast<float>(ghi) * 0.015625f);
                    else if (grp == 2) v = static_cast<_Float16>(static_cast<float>(rg * 512.0));
                    else v = static_cast<_Float16>(static_cast<float>(rg * 8.0));
                }
            }
            if (ke == 0) {
                const double m2 = static_cast<double>(static_cast<float>(m2s)) * 0.0009765625;
                d.xmulo[p] = static_cast<float>((r - m2) * 512.0);
            }
        } else if (ke == 0) d.xmulo[p] = -INFINITY;
        d.xsplit[i] = __builtin_bit_cast(unsigned short, v);
    }
}

__global__ void __launch_bounds__(kBlock) k_xh_stats(DevSim d) {
    __shared__ double red[kBlock];
    const double log2e = 1.4426950408889634074;
    const uint32_t which = blockIdx.x;           // k < 2 KH: column k; 2 KH: max |Glo|; 2 KH + 1: max |mu' - m1 - m2|
    double m = 0.0;
    for (uint32_t p = threadIdx.x; p < d.P; p += kBlock) {
        if (which == 2 * d.KH + 1) {
            const double mu = d.mu_o[p] * log2e;
            const _Float16 m1 = static_cast<_Float16>(static_cast<float>(rint(mu * 32.0) * 0.03125));
            const double r = mu - static_cast<double>(static_cast<float>(m1));
            const _Float16 m2s = static_cast<_Float16>(static_cast<float>(rint(r * 65536.0) * 0.015625));
            m = fmax(m, fabs(r - static_cast<double>(static_cast<float>(m2s)) * 0.0009765625));
            continue;
        }
        for (uint32_t k = (which < 2 * d.KH ? which : 0u); k < (which < 2 * d.KH ? min(which + 1u, d.K) : d.K); ++k) {
            const double g = d.gamma[static_cast<size_t>(p) * d.K + k] * log2e;
            const _Float16 ghi = static_cast<_Float16>(static_cast<float>(rint(g * 256.0) * 0.00390625));
            const double rg = g - static_cast<double>(static_cast<float>(ghi));
            const double glo = static_cast<double>(static_cast<float>(static_cast<_Float16>(static_cast<float>(rg * 512.0)))) * 0.001953125;
            m = fmax(m, which < 2 * d.KH ? fabs(rg - glo) : fabs(glo));
        }
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = kBlock / 2; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0) d.xstats[which] = static_cast<float>(red[0] * (1.0 + 1e-6) + 1e-30);
}

// NW waves per block (4: two blocks per CU; 8: one — the two halves of the CU's users share one stream of table tiles)
template <int KH, int NH, int NL, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) k_sweep_xh(DevSim d, uint32_t t, uint32_t S) {
    constexpr int NM = NH + NL;
    constexpr uint32_t RSc = 32u * NM + 16u, TILE_B = 128u * RSc, NB = 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* g_buf = smem_raw;                                           // [NB][128][RSc]
    float* mu_buf = reinterpret_cast<float*>(g_buf + NB * TILE_B);    // [NB][128]
    float* scp_stage = mu_buf + NB * 128 + 64;                        // [NW waves][32 users][kMaxSC] super-chunk prefixes of the work item
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const int j = lane & 31, h = lane >> 5;
    const uint32_t pos0 = d.grp_lo, n_o = pos0 + d.grp_n;
    constexpr uint32_t UPB = 32u * NW;                               // users per block
    const uint32_t n_tiles_u = (d.grp_n + UPB - 1u) / UPB;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    const uint32_t n_pt = d.n_chunks / 4;                             // product tiles
    const uint32_t K = d.K;
    (void)S;

    for (uint32_t wk = blockIdx.x; wk < n_tiles_u; wk += gridDim.x) {
        const uint32_t pos = pos0 + wk * UPB + wave * 32 + j;
        const bool active = pos < n_o;
        const uint32_t slot = active ? cur[pos] : 0u;
        const size_t urow = active ? static_cast<size_t>(d.uid[slot]) : static_cast<size_t>(d.n_cap);   // inactive lanes: the dummy row
        float2* rec = d.cache_rec + urow * kMaxSC;
        float* chunkp = d.cache_chunk + urow * d.n_chunks;
        float* scp_row = d.walk_scp + urow * kMaxSC;
        __syncthreads();           // every wave is done with the LDS buffers (previous work item)
        // The prefix at the end of every super-chunk is gathered in LDS and leaves as ONE 128-byte row per user at the end of
        // the work item; the {sum, reference} records only exist for the rare users whose reference moves during the sweep
        // (k_cache_finalize / k_cache_prefix rescale those) or where the sweep does not leave the finalize output itself.
        // Written at every super-chunk end, as 4- and 8-byte scattered stores, the two cost 12 % of the kernel
        // (profiles/r5/ab_call5_xh_books.jsonl).  Only the lane h == 0 of a user touches its row: no synchronisation.
        float* sstage = scp_stage + static_cast<size_t>(wave * 32 + j) * kMaxSC;
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < static_cast<int>(kMaxSC) / 4; ++i)
                reinterpret_cast<float4*>(sstage)[i] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);     // unused entries: never counted
        }
        bool rec_on = !d.fin_in_sweep;
        const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
        const rg_v4i rs_g = raw_buffer_rsrc(d.xsplit), rs_m = raw_buffer_rsrc(d.xmulo);
        const uint32_t g_lds = lds_addr_of(g_buf), mu_lds = lds_addr_of(mu_buf);
        auto fetch_tile = [&](uint32_t ti) {
            for (uint32_t off = static_cast<uint32_t>(wave) * 1024u; off < TILE_B; off += NW * 1024u)
                dma_to_lds_b128(rs_g, g_lds + (ti % NB) * TILE_B + off, lane16, ti * TILE_B + off);
            if (wave == NW - 1 && lane < 32) dma_to_lds_b128(rs_m, mu_lds + (ti % NB) * 512u, lane16, ti * 512u);
        };
        fetch_tile(0);
        // ---- the user's bounds and the fp16 pieces of its omega (its own K / 2 coordinates per lane, joined across the two
        // lanes of the user); the pieces travel through LDS (tile buffer 1: its DMA goes out after the B rows are built) ----
        const double* om_row = d.omega + static_cast<size_t>(slot) * d.OMS;
        unsigned short* stage = reinterpret_cast<unsigned short*>(g_buf + TILE_B) + static_cast<size_t>(wave * 32 + j) * (2 * KH) * 3;   // (8 waves x 32 users x 20 x 6 B = one tile buffer)
        float absdot = 0.0f, sq = 0.0f, absw = 0.0f, egam = 0.0f, lob = 0.0f;
        const float glomax = d.xstats[2 * KH];
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            const uint32_t k = h * KH + s;
            double w = 0.0;
            if (active && k < K) w = om_row[k];
            const XhPieces pc = xh_split(w);
            stage[3 * k] = __builtin_bit_cast(unsigned short, pc.mid9);     // [k][wmid 2^9, wlo 2^15, whi]; wmid 2^6 = wmid 2^9 / 8
            stage[3 * k + 1] = __builtin_bit_cast(unsigned short, pc.lo15);
            stage[3 * k + 2] = __builtin_bit_cast(unsigned short, pc.hi);
            const float wf = fabsf(static_cast<float>(w)) * 1.0000002f;
            const float r1f = fabsf(static_cast<float>(pc.r1)) * 1.002f;
            absdot = fmaf(wf, d.stats[k], absdot);
            sq = fmaf(wf, wf, sq);
            absw += wf;
            egam = fmaf(wf, d.xstats[k], egam);
            // |residual terms| of coordinate k: |Ghi| (|wmid| + |wlo|) + |Glo| (|whi| + |wmid|)
            lob = fmaf(r1f, d.stats[k] * kLog2e * 1.000001f + 0.00390625f, lob);
            lob = fmaf(wf + r1f, glomax, lob);
        }
        absdot += swap32(absdot); sq += swap32(sq); absw += swap32(absw); egam += swap32(egam); lob += swap32(lob);
        lob += d.xstats[2 * KH + 1];               // the seed: |mu' - m1 - m2|
        const float Ahat = ahat_of(d, mumax, g2max, absdot, sq);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- B fragments ----
        f16x8 Bm[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned short v = 0;
                if (m < NH) {
                    const uint32_t ke = 16 * m + 8 * h + e;
                    if (ke < K) v = stage[3 * ke + 2];
                    else if (ke == 16 * NH - 3) v = 0x3C00;             // 1.0: the m1 column
                    else if (ke == 16 * NH - 2) v = 0x1400;             // 2^-10: the m2 2^10 column
                } else {
                    const uint32_t s = 16 * (m - NH) + 8 * h + e;
                    const uint32_t grp = (s >= K) + (s >= 2 * K) + (s >= 3 * K) + (s >= 4 * K);
                    if (grp < 3) v = stage[3 * (s - grp * K) + grp];
                    else if (grp == 3)       // wmid 2^6: the same fp16 significand three binades down (exact unless it leaves the normal range)
                        v = __builtin_bit_cast(unsigned short, static_cast<_Float16>(static_cast<float>(__builtin_bit_cast(_Float16, stage[3 * (s - 3 * K)])) * 0.125f));
                }
                Bm[m][e] = __builtin_bit_cast(_Float16, v);
            }
        float q = 0.0f, qabs_max = 0.0f;
        auto set_reference = [&](float qn) {       // an integer |q| <= 2047: one exact fp16 value
            qn = fminf(fmaxf(qn, -2047.0f), 2047.0f);
            q = qn;
            qabs_max = fmaxf(qabs_max, fabsf(qn));
            if (h == 1) Bm[NH - 1][7] = static_cast<_Float16>(-qn);
        };
        const char* a_lane = g_buf + j * RSc + 16 * h;
        const char* m_lane = reinterpret_cast<const char*>(mu_buf) + 16 * h;
        auto a_base = [&](uint32_t ci) { return a_lane + ((ci >> 2) % NB) * TILE_B + (ci & 3) * (32 * RSc); };
        auto m_base = [&](uint32_t ci) { return m_lane + ((ci >> 2) % NB) * 512u + (ci & 3) * 128u; };
        struct AOps { f16x8 a[NM]; };
        auto load_a = [&](AOps& o, const char* ab, int idx) { o.a[idx] = *reinterpret_cast<const f16x8*>(ab + 32 * idx); };
        auto load_seed = [&](f32x16& acc, const char* mb, int qq) {
            const float4 m = *reinterpret_cast<const float4*>(mb + 32 * qq);
            acc[4 * qq] = m.x; acc[4 * qq + 1] = m.y; acc[4 * qq + 2] = m.z; acc[4 * qq + 3] = m.w;
        };
        auto mm = [](const f16x8& a, const f16x8& b, const f32x16& c) -> f32x16 {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        };
        const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // MFMA m of a chunk: the sequence L0 H0 L1 H1 ... alternates the two accumulators while both have work
        auto issue = [&](const AOps& o, int m, f32x16& H, f32x16& L) {
            if (m < 2 * NH) {                  // (NH <= NL in every instantiation)
                const int i = m >> 1;
                if (m & 1) H = mm(o.a[i], Bm[i], i == 0 ? zero16 : H);
                else L = mm(o.a[NH + i], Bm[NH + i], L);
            } else if (!(RG_XH_ABL & 1)) L = mm(o.a[m], Bm[m], L);   // residual step m - NH sits at operand index NH + (m - NH)
        };
        // ---- per-chunk bookkeeping (prefix form: what k_walk2 searches).  The sum of chunk i is produced at the end of step i + 1
        // and BOOKED inside step i + 2, between that step's MFMAs: nothing waits for it (a wave's pace is its own chain of
        // dependent instructions — timing builds, profiles/r5/ab_call2_xh_ablation.jsonl: with the books closed at the end of
        // every step, behind branches on the chunk's position, they were a third of the kernel).  The chunk's position in its
        // tile is a compile-time constant of the step. ----
        double s_sc = 0.0;
        float wcmax = 0.0f;
        int n_resc = 0;
        float q_done = 0.0f, q_next = 0.0f;
        double run_pref = 0.0;
        float q_run = 0.0f;
        uint32_t sc_cur = 0;
        uint32_t sc_left = d.sc_chunks / 4;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f;
        auto book = [&](auto cpos, uint32_t ti, float s) {    // sum of chunk `cpos` of product tile ti
            constexpr int c = decltype(cpos)::value;
            if (RG_XH_ABL & 64) { wcmax += s; return; }
            s += swap32(s);
            if constexpr (c == 0) { w0 = s; return; }
            if constexpr (c == 1) { w1 = s; return; }
            if constexpr (c == 2) { w2 = s; return; }
            if constexpr (c == 3) {
                if (q_done != q_run) { run_pref *= static_cast<double>(__builtin_amdgcn_exp2f(q_run - q_done)); q_run = q_done; }
                // the tile's four prefixes in float64, each stored as ONE rounding of the float64 value (rho = 2^-24 per stored
                // prefix: the hot row's rho_rel)
                const double b1 = run_pref + static_cast<double>(w0), b2 = b1 + static_cast<double>(w1);
                const double b3 = b2 + static_cast<double>(w2), b4 = b3 + static_cast<double>(s);
                run_pref = b4;
                if (h == 0 && !(RG_XH_ABL & 128)) *reinterpret_cast<float4*>(chunkp + static_cast<size_t>(ti) * 4) =
                    make_float4(static_cast<float>(b1), static_cast<float>(b2), static_cast<float>(b3), static_cast<float>(b4));
                wcmax = fmaxf(fmaxf(wcmax, fmaxf(w0, w1)), fmaxf(w2, s));
                s_sc += static_cast<double>((w0 + w1) + (w2 + s));
                if (--sc_left == 0 && !(RG_XH_ABL & 1024)) {
                    if (h == 0 && !(RG_XH_ABL & 128)) {
                        sstage[sc_cur] = static_cast<float>(run_pref);
                        if (rec_on) rec[sc_cur] = make_float2(static_cast<float>(s_sc), q_done);
                    }
                    s_sc = 0.0;
                    // some logit is >= ~43 above the reference: re-reference from the next super-chunk that has not started
                    if (wcmax > 2.8e14f) q_next = fmaxf(q_next, q_done + floorf(__builtin_amdgcn_logf(wcmax)));
                    wcmax = 0.0f;
                    ++sc_cur;
                    sc_left = d.sc_chunks / 4;
                }
            }
        };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;

        // One step: MFMAs of a chunk (`co` operands) into (Ha, La: La holds its seed) | join + exp-sum of the chunk before, in
        // (Hb, Lb) -> sum | operand rows of the next chunk -> `no`, its seed -> Lb | `filler` (the books of the chunk two back)
        // (round 6, as in k_draw_tp: an exp's result is consumed a slot later, plain v_fma_f32 / v_add_f32 instead of the packed
        // forms — packed fp32 beside MFMAs costs ~13 cycles more per instruction, MI355X_MICROARCH.md — adds written as asm so
        // that the SLP pass cannot re-pack them)
#ifndef RG_XH_SCALAR
#define RG_XH_SCALAR 1
#endif
        auto fadd = [](float a, float b) -> float {
            float r;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            return r;
        };
        auto stream = [&](const AOps& co, AOps& no, uint32_t ci_next, f32x16& Ha, f32x16& La, f32x16& Hb, f32x16& Lb, float& sum, auto&& filler) {
            constexpr int ES = NM > 1 ? NM - 1 : 1;            // slots that carry exps
            auto e_lo = [](int m1) { return (m1 * 8 / ES) * 2; };
            f32x2v x[4];
            float xs[4], Y[16 / (ES > 4 ? 4 : ES) + 4];
            const char* ab = a_base(ci_next);
            const char* mb = m_base(ci_next);
            RG_XPIN();
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                issue(co, m, Ha, La);
                // the next chunk's operand rows: spread over the first slots
#pragma unroll
                for (int i = m * NM / (NM > 2 ? NM - 2 : 1); i < (m + 1) * NM / (NM > 2 ? NM - 2 : 1) && i < NM; ++i) if (!(RG_XH_ABL & 4)) load_a(no, ab, i);
                if (m == 1 || NM == 1) {
                    // join: logit = H + 2^-9 L (one rounding), in place
                    if (RG_XH_SCALAR) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) Hb[r] = fmaf(Lb[r], 0.001953125f, Hb[r]);
                    } else {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2v hv = {Hb[r], Hb[r + 1]}, lv = {Lb[r], Lb[r + 1]};
                        hv = lv * 0.001953125f + hv;
                        Hb[r] = hv[0]; Hb[r + 1] = hv[1];
                    }
                    }
                }
                if (m == 0) filler();          // (independent of this step's MFMAs and of the chunk being joined)
                if (m >= 1) {
                    asm volatile("" : "+v"(Hb));
                    const int m1 = NM > 1 ? m - 1 : 0;
                    if (RG_XH_SCALAR) {
                        if (m1 > 0) {                              // the exps of the slot before
#pragma unroll
                            for (int e = e_lo(m1 - 1); e < e_lo(m1); ++e) { if (e < 4) xs[e] = Y[e - e_lo(m1 - 1)]; else xs[e & 3] = fadd(xs[e & 3], Y[e - e_lo(m1 - 1)]); }
                        }
#pragma unroll
                        for (int e = e_lo(m1); e < e_lo(m1 + 1); e += 2) {
                            float ya = (RG_XH_ABL & 2) ? Hb[e] * 0.5f : __builtin_amdgcn_exp2f(Hb[e]), yb = (RG_XH_ABL & 2) ? Hb[e + 1] * 0.5f : __builtin_amdgcn_exp2f(Hb[e + 1]);
                            asm volatile("" : "+v"(ya), "+v"(yb));
                            Y[e - e_lo(m1)] = ya; Y[e - e_lo(m1) + 1] = yb;
                        }
                    } else {
#pragma unroll
                    for (int e = (m1 * 8 / ES) * 2; e < ((m1 + 1) * 8 / ES) * 2; e += 2) {
                        f32x2v y = {(RG_XH_ABL & 2) ? Hb[e] * 0.5f : __builtin_amdgcn_exp2f(Hb[e]), (RG_XH_ABL & 2) ? Hb[e + 1] * 0.5f : __builtin_amdgcn_exp2f(Hb[e + 1])};
                        asm volatile("" : "+v"(y));
                        if (e < 8) x[e / 2] = y; else x[(e / 2) & 3] += y;
                    }
                    }
                }
                if (m >= 2 && m - 2 < 4 && !(RG_XH_ABL & 8)) load_seed(Lb, mb, m - 2);
                RG_XPIN();
            }
            if (NM < 6) {
#pragma unroll
                for (int qq = (NM > 2 ? NM - 2 : 0); qq < 4; ++qq) load_seed(Lb, mb, qq);
            }
            if (RG_XH_SCALAR) {
#pragma unroll
                for (int e = e_lo(ES - 1); e < 16; ++e) { if (e < 4) xs[e] = Y[e - e_lo(ES - 1)]; else xs[e & 3] = fadd(xs[e & 3], Y[e - e_lo(ES - 1)]); }
                sum = fadd(fadd(xs[0], xs[2]), fadd(xs[1], xs[3]));
            } else {
            x[0] += x[2]; x[1] += x[3]; x[0] += x[1];
            sum = x[0][0] + x[0][1];
            }
            RG_XPIN();
        };

        AOps oa, ob;
        f32x16 H0, L0, H1, L1;
        RG_DMA_WAIT();
        __syncthreads();           // tile 0 landed (and every wave has built its B rows from the stage in tile buffer 1)
        if (n_pt > 1) fetch_tile(1);
#pragma unroll
        for (int i = 0; i < NM; ++i) load_a(oa, a_base(0), i);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { load_seed(L0, m_base(0), qq); load_seed(L1, m_base(0), qq); }
        RG_XPIN();
        {   // chunk 0 with reference 0: its largest logit, rounded up to an integer, becomes the reference
#pragma unroll
            for (int m = 0; m < NM; ++m) issue(oa, m, H1, L1);
            float cm = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) cm = fmaxf(cm, fmaf(L1[r], 0.001953125f, H1[r]));
            set_reference(fmaxf(ceilf(fmaxf(cm, swap32(cm))), -1.0e30f));
            q_done = q_next = q;
        }
        RG_XPIN();
        // head: chunk 0's MFMAs with nothing to exp yet; chunk 1's rows and seed arrive meanwhile
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            issue(oa, m, H0, L0);
            load_a(ob, a_base(1), m);
            RG_XPIN();
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) load_seed(L1, m_base(1), qq);
        RG_XPIN();
        const uint32_t n_ch = d.n_chunks;
        uint32_t sc_issue_left = d.sc_chunks / 4;       // tiles left in the super-chunk being ISSUED
        // Step i = MFMAs of chunk i | exp-sum of chunk i - 1 | books of chunk i - 2 | rows + seed of chunk i + 1.  Even chunks use
        // operand set `oa` and accumulators (H0, L0), odd ones `ob` and (H1, L1).
        float s_pend, s_new;
        stream(ob, oa, 2, H1, L1, H0, L0, s_pend, [] {});                   // step 1
        uint32_t T = 0;
        for (; T + 1 < n_pt; ++T) {        // steps 4T + 2 .. 4T + 5: the books of tile T
            const uint32_t c0 = 4 * T;
            stream(oa, ob, c0 + 3, H0, L0, H1, L1, s_new, [&] { book(C0{}, T, s_pend); });
            s_pend = s_new;
            // step 4T + 3 reads chunk 0 of tile T + 1: it has landed, and every wave is done reading tile T's buffer, which is
            // refilled with tile T + 2
            // (measured and not kept, profiles/r5/ab_call4_*, ab_call6_*: a counted vmcnt that leaves the newest store in flight;
            // the books' stores issued right behind this barrier instead of a step before it — neither moves the kernel: what
            // the stores cost, 17 % of it, is not this wait)
            if (!(RG_XH_ABL & 16)) RG_TILE_BARRIER(0);
            if (T + 2 < n_pt && !(RG_XH_ABL & 16)) fetch_tile(T + 2);
            stream(ob, oa, c0 + 4, H1, L1, H0, L0, s_new, [&] { book(C1{}, T, s_pend); });
            s_pend = s_new;
            // step 4T + 4 issues the first chunk of tile T + 1: a super-chunk may start there
            if (--sc_issue_left == 0) sc_issue_left = d.sc_chunks / 4;
            const bool sc_start = sc_issue_left == d.sc_chunks / 4;
            if (sc_start && q_next != q) {
                if (!rec_on) {     // the first move of this user's reference: the records of the super-chunks booked so far (all
                    rec_on = true; // on the reference in force until now; their sums from the staged prefixes)
                    if (h == 0)
                        for (uint32_t sc = 0; sc < sc_cur; ++sc)
                            rec[sc] = make_float2(sstage[sc] - (sc ? sstage[sc - 1] : 0.0f), q);
                }
                set_reference(q_next); n_resc += 1;
            }
            stream(oa, ob, c0 + 5, H0, L0, H1, L1, s_new, [&] { book(C2{}, T, s_pend); });
            s_pend = s_new;
            stream(ob, oa, min(c0 + 6, n_ch - 1), H1, L1, H0, L0, s_new, [&] { book(C3{}, T, s_pend); });   // (may flush the finished super-chunk with q_done)
            s_pend = s_new;
            if (sc_start) q_done = q;      // the sums pending from here on were taken with the new reference
        }
        {   // the last tile: steps n_ch - 2, n_ch - 1, the last chunk's own sums
            stream(oa, ob, n_ch - 1, H0, L0, H1, L1, s_new, [&] { book(C0{}, T, s_pend); });
            s_pend = s_new;
            stream(ob, oa, n_ch - 1, H1, L1, H0, L0, s_new, [&] { book(C1{}, T, s_pend); });
            book(C2{}, T, s_new);
            float sl = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sl += __builtin_amdgcn_exp2f(fmaf(L1[r], 0.001953125f, H1[r]));
            book(C3{}, T, sl);
        }
        if (sc_left != d.sc_chunks / 4 && h == 0) {            // partial last super-chunk
            if (rec_on) rec[sc_cur] = make_float2(static_cast<float>(s_sc), q_done);
            sstage[sc_cur] = static_cast<float>(run_pref);
        }
        if (h == 0) {              // the user's super-chunk prefixes: one row
#pragma unroll
            for (int i = 0; i < static_cast<int>(kMaxSC) / 4; ++i)
                reinterpret_cast<float4*>(scp_row)[i] = reinterpret_cast<const float4*>(sstage)[i];
        }
        if (active && h == 0) d.cache_resc[urow] = static_cast<uint8_t>(min(n_resc, 255));
        if (d.fin_in_sweep && active && h == 0 && n_resc == 0) {
            // what k_cache_finalize and k_cache_prefix would leave (one reference for the whole sweep): Q and delta in the cache
            // row, omega32 behind them, the unused super-chunk prefixes, the hot row {S~, delta, Q, empty memo | rho_rel}
            const double delta = xh_delta<NL>(d, static_cast<double>(Ahat), static_cast<double>(absw), static_cast<double>(egam),
                                              static_cast<double>(lob), static_cast<double>(qabs_max));
            const float dlt = static_cast<float>(delta * 1.000001);          // rounded up: the budget must not shrink
            float4* row4 = reinterpret_cast<float4*>(d.cache_row + urow * d.cache_row_f);
            row4[8] = make_float4(q, dlt, 0.0f, 0.0f);
            float ou[2 * KH];
#pragma unroll
            for (int k = 0; k < 2 * KH; ++k) ou[k] = static_cast<uint32_t>(k) < K ? static_cast<float>(om_row[k]) : 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < (2 * KH) / 4; ++k4) row4[11 + k4] = make_float4(ou[4 * k4], ou[4 * k4 + 1], ou[4 * k4 + 2], ou[4 * k4 + 3]);
#pragma unroll
            for (int k = ((2 * KH) / 4) * 4; k < 2 * KH; ++k) reinterpret_cast<float*>(row4)[44 + k] = ou[k];
            float* hot = d.walk_hot + urow * 32;
            *reinterpret_cast<float4*>(hot) = make_float4(static_cast<float>(run_pref), dlt * 1.000001f, q, __builtin_bit_cast(float, 0u));
            // float 31: the in-chunk budget of the walk's fp32 recompute (hot_budgets: rho_rel = 2^-23 with it), or rho_rel = 2^-20
            // for a user whose delta is the loose one anyway
            const float dcf = fmaxf(static_cast<float>(xh_delta_chunk(d, static_cast<double>(Ahat), static_cast<double>(absw)) * 1.000001), kHotDcMin);
            hot[31] = xh_eligible(static_cast<double>(Ahat), static_cast<double>(qabs_max)) ? dcf : kRhoLoose;
        }
    }
}

draw_kernel_t xh_kernel_for(const DevSim& d, int waves) {
    if (d.XNH == 1 && d.XNL == 2 && d.KH == 4) return waves == 8 ? k_sweep_xh<4, 1, 2, 8> : k_sweep_xh<4, 1, 2, 4>;
    if (d.XNH == 2 && d.XNL == 5 && d.KH == 10) return waves == 8 ? k_sweep_xh<10, 2, 5, 8> : k_sweep_xh<10, 2, 5, 4>;
    return nullptr;
}
void (*xh_table_kernel())(DevSim) { return k_make_xh_table; }
void (*xh_stats_kernel())(DevSim) { return k_xh_stats; }

}  // namespace rgk
