// rg_draw_lds.hip — librecogym_hip.so, unit 9 of 9: k_draw_tp, the sigma_omega > 0 sweep whose search never leaves the CU.
// (see rg_common.hpp for the shared types and helpers, DESIGN.md §4 "Round 6" for the measurements)
//
// What it replaces: k_draw_bf16p<.., F16> + search_and_emit where every organic draw needs its own product sweep (omega
// drifts: reco_env_v1.py:85-100, the draw itself: update_product_view, reco_env_v1.py:119-128).  That kernel writes the
// exp-sum of every 32-product chunk (1.25 KB per draw at P = 10^4) and 32 super-chunk records to a global scratch and reads
// them back in a search that is a chain of ~12 dependent memory round trips per user tile — 29 % of its time, 16x the
// algorithmic HBM bytes (profiles/r5).  Here
//   * the sweep (same tiles, same MFMA / exp stream, same arithmetic and certificate budget) keeps ONE number per 128-product
//     tile and user: the running prefix of the exp-sums at the tile's end, float64 in a register, stored as ONE fp32 rounding
//     in LDS ([wave][user][tile]: 316 B per user at P = 10^4).  Nothing goes to global memory during the sweep;
//   * the search counts the user's tile prefixes <= u S in LDS (two lanes per user), then RECOMPUTES the 128 products of that
//     tile in fp32 from the chunk-major copy of Gamma (eight users per pass, eight lanes per user, 16 products per lane) — the
//     arithmetic of search_and_emit's chunk recompute on four chunks — and takes the certificate of cert_correlated with
//     A = the tile's starting prefix: two round trips (the Gamma tile in two batches) instead of twelve;
//   * no re-referencing: the reference is the first chunk's largest logit; a user whose sums overflow fp32 (a logit > 2^127
//     above it) fails the certificate and is drawn in float64 like every uncertified draw.
// LDS: tiles 2 x 18 KB + prefixes 40 KB per block of 4 waves x 32 users (P = 10^4, K = 20): two blocks per CU.  Served: the
// two-way fp16 split classes with K <= 20 (KH <= 10), unsliced sweeps (S = 1), no per-user cache; everything else keeps
// k_draw_bf16p.

#include "rg_common.hpp"

namespace rgk {

template <int KH, int N1>
__global__ void __launch_bounds__(kBlock, 2) k_draw_tp(DevSim d, uint32_t t, uint32_t NTs) {
    constexpr uint32_t NB = 2;
    constexpr int NM = N1;
    constexpr int EXS = NM > 1 ? NM - 1 : 1;
    constexpr int K2 = 2 * KH;
    constexpr uint32_t RSc = 32 * N1 + 16, TILE_B = 128 * RSc;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* g_buf = smem_raw;                                           // [NB][128][RSc]: the tile in use and the one in flight
    float* mu_buf = reinterpret_cast<float*>(g_buf + NB * TILE_B);    // [NB][128] (+ pad)
    float* tpref = mu_buf + NB * 128 + 64;                            // [4 waves][32 users][NTs] tile prefixes
    // omega32 of the block's users [4][32][K2]: in tile buffer 1 while the B rows are built (its first DMA goes out after
    // them), in tile buffer 0 again for the search (behind a barrier: every wave is done with the tiles)
    float* om_pre = reinterpret_cast<float*>(g_buf + TILE_B);
    float* om_post = reinterpret_cast<float*>(g_buf);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const int j = lane & 31, h = lane >> 5;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_tiles = (n_o + 127) / 128;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    const uint32_t n_pt = d.n_chunks / 4;                             // product tiles
    const uint32_t np = 2 * n_pt;                                     // pairs of chunks
    float* trow = tpref + static_cast<size_t>(wave * 32 + j) * NTs;   // this lane's user's prefixes

    struct PairOps { bf16x8 A0[N1], A1[N1]; };

    for (uint32_t tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
        const uint32_t pos = tb * 128 + wave * 32 + j;
        const bool active = pos < n_o;
        const uint32_t slot = active ? cur[pos] : 0u;
        __syncthreads();           // every wave is done with the LDS buffers (previous work item's search)
        const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
        const rg_v4i rs_g = raw_buffer_rsrc(d.gsplit), rs_m = raw_buffer_rsrc(d.mu32s);
        const uint32_t g_lds = lds_addr_of(g_buf), mu_lds = lds_addr_of(mu_buf);
        auto fetch_tile = [&](uint32_t ti) {
            for (uint32_t off = static_cast<uint32_t>(wave) * 1024u; off < TILE_B; off += 4096u)
                dma_to_lds_b128(rs_g, g_lds + (ti % NB) * TILE_B + off, lane16, ti * TILE_B + off);
            if (wave == 3 && lane < 32) dma_to_lds_b128(rs_m, mu_lds + (ti % NB) * 512u, lane16, ti * 512u);
        };
        fetch_tile(0);
        // ---- omega32 of the user -> LDS stage (also the logit error bound) ----
        float* omu = om_pre + (wave * 32 + j) * K2;
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
        const double delta_fixed = kDeltaFixedBf16 + f16_extra_delta(d, Ahat, absw);
        if (h == 0) for (uint32_t i = n_pt; i < NTs; ++i) trow[i] = INFINITY;        // (row padding: never counted)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- B fragments: [w1 | w1 | w2 | 0 .. | -q] ----
        bf16x8 Bm[NM];
        {
            const uint32_t K = d.K;
#pragma unroll
            for (int s = 0; s < N1; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t ke = 16 * s + 8 * h + e;
                    unsigned short sp[2] = {0, 0};
                    if (ke < 3 * K) f16_split2(omu[ke % K], sp);
                    Bm[s][e] = static_cast<short>(ke < 2 * K ? sp[0] : sp[1]);
                }
        }
        float q = 0.0f;            // reference (log2 units, an integer): one exact fp16 piece
        constexpr uint32_t TB = TILE_B;
        const char* a_lane = g_buf + j * RSc + 16 * h;
        const char* m_lane = reinterpret_cast<const char*>(mu_buf) + 16 * h;
        auto a_base = [&](uint32_t pi) { return a_lane + ((pi >> 1) % NB) * TB + (pi & 1) * (64 * RSc); };
        auto m_base = [&](uint32_t pi) { return m_lane + ((pi >> 1) % NB) * (128 * 4) + (pi & 1) * (64 * 4); };
        auto load_a = [&](PairOps& o, const char* ab, int idx) {        // A row block idx of the pair's chunk 0 / 1
            if (idx < N1) o.A0[idx < N1 ? idx : 0] = *reinterpret_cast<const bf16x8*>(ab + 32 * idx);
            else o.A1[idx - N1 < N1 ? idx - N1 : 0] = *reinterpret_cast<const bf16x8*>(ab + 32 * RSc + 32 * (idx - N1));
        };
        auto load_mu = [&](f32x16& acc, const char* mb, int which, int qq) {   // mu quad qq, into the accumulator it seeds
            const float4 m = *reinterpret_cast<const float4*>(mb + 128 * which + 32 * qq);
            acc[4 * qq] = m.x; acc[4 * qq + 1] = m.y; acc[4 * qq + 2] = m.z; acc[4 * qq + 3] = m.w;
        };
        auto mm = [](const bf16x8& a, const bf16x8& b, const f32x16& c) -> f32x16 {
            using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        };
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        // One pair (k_draw_bf16p's stream): MFMAs of (co) into (a0, a1), which already hold the pair's mu | exp-sum of
        // (p0, p1) -> (s0, s1) | A rows of pair pi_next -> no, its mu -> (p0, p1) once their exps are done
        auto stream = [&](const PairOps& co, PairOps& no, uint32_t pi_next, f32x16& a0, f32x16& a1,
                          f32x16& p0, f32x16& p1, float& s0, float& s1) {
            f32x2 x0[4], x1[4];
            const char* ab = a_base(pi_next);
            const char* mb = m_base(pi_next);
            RG_PIN();
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                a0 = mm(co.A0[m], Bm[m], a0);
                if (m < EXS) {
                    asm volatile("" : "+v"(p0));                        // (exps may not float above this slot)
#pragma unroll
                    for (int i = (2 * m) * (2 * N1) / (2 * EXS); i < (2 * m + 1) * (2 * N1) / (2 * EXS); ++i) load_a(no, ab, i);
#pragma unroll
                    for (int e = (m * 8 / EXS) * 2; e < ((m + 1) * 8 / EXS) * 2; e += 2) {       // exps in pairs
                        f32x2 y = {__builtin_amdgcn_exp2f(p0[e]), __builtin_amdgcn_exp2f(p0[e + 1])};
                        asm volatile("" : "+v"(y));
                        if (e < 8) x0[e / 2] = y; else x0[(e / 2) & 3] += y;
                    }
                } else {
#pragma unroll
                    for (int qq = (m - EXS) * 4 / (NM > EXS ? NM - EXS : 1); qq < (m - EXS + 1) * 4 / (NM > EXS ? NM - EXS : 1); ++qq) load_mu(p0, mb, 0, qq);
                }
                RG_PIN();
                a1 = mm(co.A1[m], Bm[m], a1);
                if (m < EXS) {
                    asm volatile("" : "+v"(p1));
#pragma unroll
                    for (int i = (2 * m + 1) * (2 * N1) / (2 * EXS); i < (2 * m + 2) * (2 * N1) / (2 * EXS); ++i) load_a(no, ab, i);
#pragma unroll
                    for (int e = (m * 8 / EXS) * 2; e < ((m + 1) * 8 / EXS) * 2; e += 2) {
                        f32x2 y = {__builtin_amdgcn_exp2f(p1[e]), __builtin_amdgcn_exp2f(p1[e + 1])};
                        asm volatile("" : "+v"(y));
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

        // ---- the books: one float64 add and one 4-byte LDS store per 128-product tile ----
        double run_pref = 0.0;     // running prefix of the exp-sums (every lane of the user holds it)
        float wlo = 0.0f;
        auto book = [&](uint32_t pe, float s0, float s1) {    // sums of pair pe (chunks 2 pe, 2 pe + 1)
            float s = s0 + s1;
            s += swap32(s);
            if (!(pe & 1)) { wlo = s; return; }
            // (fp32 inside the tile: <= 10 roundings per term from the exp to here, part of the fixed budget; the prefix itself
            // float64, stored as one fp32 rounding: rho)
            run_pref += static_cast<double>(wlo + s);
            if (h == 0) trow[pe >> 1] = static_cast<float>(run_pref);
        };

        PairOps oa, ob;
        f32x16 a0, a1, p0, p1;
        RG_DMA_WAIT();
        __syncthreads();           // tile 0 landed; every wave has built its B rows from the stage in tile buffer 1
        if (1 < n_pt) fetch_tile(1);
#pragma unroll
        for (int i = 0; i < 2 * N1; ++i) load_a(oa, a_base(0), i);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { load_mu(a0, m_base(0), 0, qq); load_mu(a1, m_base(0), 1, qq); load_mu(p0, m_base(0), 0, qq); }
        RG_PIN();
        {   // first chunk with reference 0: its max (an integer after ceil: exact in one fp16 piece and in exp2 differences)
            // becomes the reference
#pragma unroll
            for (int m = 0; m < NM; ++m) p0 = mm(oa.A0[m], Bm[m], p0);
            float cm = p0[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) cm = fmaxf(cm, p0[r]);
            float qn = fmaxf(ceilf(fmaxf(cm, swap32(cm))), -1.0e30f);
            qn = fminf(fmaxf(qn, -2047.0f), 2047.0f);
            q = qn;
            if (h == 1) Bm[N1 - 1][7] = static_cast<short>(__builtin_bit_cast(unsigned short, static_cast<_Float16>(-qn)));
        }
        // head: pair 0's MFMAs with nothing to exp yet; pair 1's A rows and mu arrive meanwhile
        RG_PIN();
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            a0 = mm(oa.A0[m], Bm[m], a0);
            if (m < N1) load_a(ob, a_base(1), m);
            else if (m < N1 + 4) load_mu(p0, m_base(1), 0, m - N1);
            RG_PIN();
            a1 = mm(oa.A1[m], Bm[m], a1);
            if (m < N1) load_a(ob, a_base(1), N1 + m);
            else if (m < N1 + 4) load_mu(p1, m_base(1), 1, m - N1);
            RG_PIN();
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { load_mu(p0, m_base(1), 0, qq); load_mu(p1, m_base(1), 1, qq); }
        RG_PIN();
        // Steady state, straight-line: [second pair of tile T | first pair of tile T + 1] per iteration
        uint32_t pi = 1;
        for (; pi + 1 < np; pi += 2) {
            float s0, s1;
            const uint32_t T = pi >> 1;
            // tile barrier: every wave holds tile T's operands (its buffer is refilled with tile T + 2); tile T + 1, whose DMA
            // went out at the last barrier, has landed (no other vector-memory operation is in flight in this loop)
            RG_TILE_BARRIER(0);
            if (T + 2 < n_pt) fetch_tile(T + 2);
            stream(ob, oa, pi + 1, p0, p1, a0, a1, s0, s1);                  // MFMAs of pair pi | sums of pair pi - 1
            book(pi - 1, s0, s1);
            stream(oa, ob, pi + 2, a0, a1, p0, p1, s0, s1);                  // MFMAs of pair pi + 1 | sums of pair pi
            book(pi, s0, s1);
        }
        {   // the last pair (second pair of the last tile), then its own sums
            float s0, s1;
            stream(ob, oa, pi, p0, p1, a0, a1, s0, s1);                      // (operand fetch of a "next" pair: this one again, unused)
            book(pi - 1, s0, s1);
#pragma unroll
            for (int r = 0; r < 16; ++r) { p0[r] = __builtin_amdgcn_exp2f(p0[r]); p1[r] = __builtin_amdgcn_exp2f(p1[r]); }
            book(pi, tree(p0), tree(p1));
        }

        // =========================== the search: LDS, then one tile of Gamma ===========================
        // omega32 again (the stage was overwritten by tile 1): requested now, staged behind the barrier below
        float wre[KH];
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            const uint32_t k = h * KH + s;
            wre[s] = (active && k < d.K) ? static_cast<float>(d.omega[static_cast<size_t>(slot) * d.OMS + k]) : 0.0f;
        }
        const uint32_t uidx = d.uid[slot];
        const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
        const double u_draw = organic_uniform(d, uidx, user, t);
        const float Sf = static_cast<float>(run_pref);
        const double S = static_cast<double>(Sf);
        const double tau = u_draw * S;
        // tile: the number of tile prefixes <= tau (they ascend); a float x is <= tau iff x <= the largest float <= tau
        float tf = static_cast<float>(tau);
        if (static_cast<double>(tf) > tau) tf = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, tf) - 1u);   // (tau >= 0)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // trow: written by the lanes h == 0, read by both
        __builtin_amdgcn_wave_barrier();
        uint32_t cnt = 0;
        {
            const float4* r4 = reinterpret_cast<const float4*>(trow);
            const uint32_t n4 = NTs / 4;
            for (uint32_t i = h; i < n4; i += 2) {
                const float4 x = r4[i];
                cnt += (x.x <= tf) + (x.y <= tf) + (x.z <= tf) + (x.w <= tf);
            }
        }
        cnt += static_cast<uint32_t>(__shfl_xor(static_cast<int>(cnt), 32));
        const bool found_t = cnt < n_pt && S > 0.0 && S < 3.0e38;
        const uint32_t ti_star = cnt < n_pt ? cnt : n_pt - 1u;
        const double pb = ti_star ? static_cast<double>(trow[ti_star - 1u]) : 0.0;    // A: the prefix at the tile's start
        const float remf = static_cast<float>(tau - pb);
        const double delta = static_cast<double>(d.K + 5) * 5.9604644775390625e-08 * static_cast<double>(Ahat) + delta_fixed;
        __syncthreads();           // every wave is done with the tile buffers
        {
            float* o = om_post + (wave * 32 + j) * K2 + h * KH;
#pragma unroll
            for (int s = 0; s < KH; ++s) o[s] = wre[s];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- the 128 products of tile ti_star: eight users per pass, eight lanes per user, per lane four consecutive products
        // of each of the tile's four chunks (every load a 128-byte run per k and user) ----
        int r_idx = -1;
        float r_a = 0.0f, r_b = 0.0f;
        {
            const int lane_w = 32 * h + j, grp = lane_w >> 3, gl = lane_w & 7;
#pragma unroll 1
            for (int ps = 0; ps < 4; ++ps) {
                const int u = 8 * ps + grp;                        // the user this group works for (its h = 0 lane)
                const uint32_t ts = static_cast<uint32_t>(__shfl(static_cast<int>(ti_star), u));
                const float Qs = __shfl(q, u);
                const float rems = __shfl(remf, u);
                const float* ou = om_post + (wave * 32 + u) * K2;
                float wv[K2];
#pragma unroll
                for (int k4 = 0; k4 < K2 / 4; ++k4) {
                    const float4 w4 = *reinterpret_cast<const float4*>(ou + 4 * k4);
                    wv[4 * k4] = w4.x; wv[4 * k4 + 1] = w4.y; wv[4 * k4 + 2] = w4.z; wv[4 * k4 + 3] = w4.w;
                }
#pragma unroll
                for (int k = (K2 / 4) * 4; k < K2; ++k) wv[k] = ou[k];
                float qx[4][4];                                    // in-lane inclusive prefixes of every chunk
                float exl[4], tot[4];
#pragma unroll
                for (int c2 = 0; c2 < 4; c2 += 2) {                // two chunks' rows in flight
                    float4 gk[2][K2], l[2];
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const uint32_t cs = ts * 4 + c2 + cc;
                        const float4* gp = reinterpret_cast<const float4*>(d.gamma32t + (static_cast<size_t>(cs) * K2) * 32) + gl;
                        l[cc] = *(reinterpret_cast<const float4*>(d.mu32 + cs * 32) + gl);
#pragma unroll
                        for (int k = 0; k < K2; ++k) gk[cc][k] = gp[k * 8];
                    }
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
                        for (int k = 0; k < K2; ++k) {
                            l[cc].x = fmaf(gk[cc][k].x, wv[k], l[cc].x); l[cc].y = fmaf(gk[cc][k].y, wv[k], l[cc].y);
                            l[cc].z = fmaf(gk[cc][k].z, wv[k], l[cc].z); l[cc].w = fmaf(gk[cc][k].w, wv[k], l[cc].w);
                        }
                        const float e0 = __builtin_amdgcn_exp2f(fmaf(l[cc].x, kLog2e, -Qs)), e1 = __builtin_amdgcn_exp2f(fmaf(l[cc].y, kLog2e, -Qs));
                        const float e2 = __builtin_amdgcn_exp2f(fmaf(l[cc].z, kLog2e, -Qs)), e3 = __builtin_amdgcn_exp2f(fmaf(l[cc].w, kLog2e, -Qs));
                        const int c = c2 + cc;
                        qx[c][0] = e0; qx[c][1] = qx[c][0] + e1; qx[c][2] = qx[c][1] + e2; qx[c][3] = qx[c][2] + e3;
                        float inc = qx[c][3];
#pragma unroll
                        for (int o2 = 1; o2 < 8; o2 <<= 1) {
                            const float y = __shfl_up(inc, o2, 8);
                            if (gl >= o2) inc += y;
                        }
                        float ex = __shfl_up(inc, 1, 8);
                        if (gl == 0) ex = 0.0f;
                        exl[c] = ex;
                        tot[c] = __shfl(inc, 8 * grp + 7);
                    }
                }
                // the proposal: the first product whose in-tile prefix exceeds the remainder (fp32: only a proposal — the
                // certificate is taken from the two prefixes around it)
                float f_idx = -1.0f, f_a = 0.0f, f_b = 0.0f;
                bool g_done = false;
                float off = 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float base = off + exl[c];
                    const float x0 = base + qx[c][0], x1 = base + qx[c][1], x2 = base + qx[c][2], x3 = base + qx[c][3];
                    const int j0 = x0 > rems ? 0 : x1 > rems ? 1 : x2 > rems ? 2 : x3 > rems ? 3 : -1;
                    const unsigned long long hits = __ballot(j0 >= 0);
                    const uint32_t gmask = static_cast<uint32_t>(hits >> (8 * grp)) & 0xFFu;
                    const int win = 8 * grp + (gmask ? __builtin_ctz(gmask) : 7);
                    const float c_idx = j0 >= 0 ? static_cast<float>(32 * c + 4 * gl + j0) : -1.0f;
                    const float c_a = j0 <= 0 ? (j0 == 0 ? base : x3) : j0 == 1 ? x0 : j0 == 2 ? x1 : x2;
                    const float c_b = j0 < 0 ? x3 : j0 == 0 ? x0 : j0 == 1 ? x1 : j0 == 2 ? x2 : x3;
                    const float w_idx = __shfl(c_idx, win), w_a = __shfl(c_a, win), w_b = __shfl(c_b, win);
                    if (!g_done && gmask) { f_idx = w_idx; f_a = w_a; f_b = w_b; g_done = true; }
                    off += tot[c];
                }
                // back to the user's own lanes (both halves): user u' is served in pass u' >> 3 by group u' & 7
                const int from = 8 * (j & 7);
                const float o_idx = __shfl(f_idx, from), o_a = __shfl(f_a, from), o_b = __shfl(f_b, from);
                if ((j >> 3) == ps) { r_idx = static_cast<int>(o_idx); r_a = o_a; r_b = o_b; }
            }
        }
        const uint32_t v = ti_star * 128u + static_cast<uint32_t>(max(r_idx, 0));
        // (S, pb: fp32 roundings of the float64 running prefix; a, b: fp32 sums of the tile's recomputed terms)
        const CertLin ct = cert_correlated(S, pb, static_cast<double>(r_a), static_cast<double>(r_b), delta);
        const bool ok = found_t && r_idx >= 0 && v < d.P && ct.valid &&
                        (v == 0 || u_draw * ct.den_lo > ct.num_lo) &&
                        (v == d.P - 1 || u_draw * ct.den_hi < ct.num_hi);
        if (active && h == 0) {
            if (ok) {
                write_organic_row(d, t, pos, slot, user, v);
                if (d.hist_cap) history_add(d, slot, v);
            } else {
                const uint32_t xi = atomicAdd(&d.exact_cnt[t], 1u);
                d.exact_list[xi] = pos;
                d.exact_ref[xi] = q;
            }
        }
    }
}

draw_kernel_t tp_kernel_for(const DevSim& d) {
    if (!d.f16 || d.wide) return nullptr;
#define RG_CASE(kh, a) if (d.KH == kh && d.N1 == a) return k_draw_tp<kh, a>;
    RG_CASE(4, 1) RG_CASE(4, 2) RG_CASE(10, 2) RG_CASE(10, 3) RG_CASE(10, 4)
#undef RG_CASE
    return nullptr;
}

}  // namespace rgk
