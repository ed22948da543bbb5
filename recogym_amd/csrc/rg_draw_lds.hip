// rg_draw_lds.hip — librecogym_hip.so, unit 9 of 9: k_draw_tp + k_pick, the sigma_omega > 0 sweep without a scratch and its search on the matrix cores.
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
//   * at the end of its sweep a user's tile prefixes <= u S are COUNTED in LDS (two lanes per user): the tile of the draw and the
//     prefix A at its start; those, the total, the reference and the certificate's budget go to k_pick (32 bytes per draw),
//     which finds the product inside the tile on the matrix cores, 32 draws of one tile at a time (below);
//   * no re-referencing: the reference is the first chunk's largest logit; a user whose sums overflow fp32 (a logit > 2^127
//     above it) has no tile and is drawn in float64 like every uncertified draw.
// LDS: tiles 2 x 18 KB + prefixes 40 KB per block of 4 waves x 32 users (P = 10^4, K = 20): two blocks per CU.  Served: the
// two-way fp16 split classes with K <= 20 (KH <= 10), unsliced sweeps (S = 1), no per-user cache; everything else keeps
// k_draw_bf16p.

#include "rg_common.hpp"
#include <type_traits>

// -DRG_TP_ABL=bits: timing experiments on k_draw_tp (results wrong by design; A/B builds only, loaded with RECOGYM_HIP_LIB).
// 1: no epilogue (uniform, tile count, record, list entry), 2: the first 8 product tiles only (what a work item costs besides its
// tile loop), 4: exps replaced by a multiply, 8: no MFMAs, 16: no tile barrier / DMA beyond the first two tiles, 32: no books,
// 64: no mu loads.  -DRG_PICK_ABL=bits on k_pick: 1: no row / history, 2: no chunk loop
// k_draw_tpw's exp sums with plain v_add_f32 (the same sums in the same order): at one wave per SIMD a packed fp32 add beside MFMAs
// costs more than the two adds it replaces (MI355X_MICROARCH.md): 3 393 -> 3 247 cycles per tile, but the clock under this load falls
// with it (2.06 -> 1.99 GHz): 1.2 % per tile, C4's sweep 1 265 -> 1 245 ms (profiles/r6/ab_call39_*)
#ifndef RG_TPW_SCALAR_ADD
#define RG_TPW_SCALAR_ADD 1
#endif
#ifndef RG_TP_ABL
#define RG_TP_ABL 0
#endif
#ifndef RG_PICK_ABL
#define RG_PICK_ABL 0
#endif
// -DRG_TPW_ABL=bits on k_draw_tpw: 1: exps replaced by a multiply, 2: no MFMAs, 4: no tile barrier / DMA beyond the first three
// tiles, 8: no books, 16: no mu seeds, 32: the A ring is not re-read (one fragment for every k-step), 64: print the shader clock
// under the kernel's load (s_memtime against the 100 MHz s_memrealtime over one work item)
#ifndef RG_TPW_ABL
#define RG_TPW_ABL 0
#endif

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
    // omega32 of the block's users [4][32][K2]: in tile buffer 1 while the B rows are built (its first DMA goes out after them)
    float* om_pre = reinterpret_cast<float*>(g_buf + TILE_B);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const int j = lane & 31, h = lane >> 5;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_tiles = (n_o + 127) / 128;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    const uint32_t n_pt = (RG_TP_ABL & 2) ? min(d.n_chunks / 4, 8u) : d.n_chunks / 4;   // product tiles
    const uint32_t np = 2 * n_pt;                                     // pairs of chunks
    float* trow = tpref + static_cast<size_t>(wave * 32 + j) * NTs;   // this lane's user's prefixes

    struct PairOps { bf16x8 A0[N1], A1[N1]; };

    for (uint32_t tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
        const uint32_t pos = tb * 128 + wave * 32 + j;
        const bool active = pos < n_o;
        const uint32_t slot = active ? cur[pos] : 0u;
        __syncthreads();           // every wave is done with the LDS buffers (previous work item)
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
        char* recp = d.tp_rec + static_cast<size_t>(active ? pos : 0u) * tp_rec_stride(KH);
        if (active) {              // omega32 behind the record's header: k_pick's B operands (each lane its half)
            float* rw = reinterpret_cast<float*>(recp + sizeof(TpRec)) + h * KH;
#pragma unroll
            for (int s = 0; s < KH; ++s) rw[s] = omu[h * KH + s];
        }
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
            if (RG_TP_ABL & 64) return;
            const float4 m = *reinterpret_cast<const float4*>(mb + 128 * which + 32 * qq);
            acc[4 * qq] = m.x; acc[4 * qq + 1] = m.y; acc[4 * qq + 2] = m.z; acc[4 * qq + 3] = m.w;
        };
        auto mm = [](const bf16x8& a, const bf16x8& b, const f32x16& c) -> f32x16 {
            using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
            if (RG_TP_ABL & 8) return c;
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        };
        auto ex2 = [](float x) -> float { return (RG_TP_ABL & 4) ? x * 0.5f : __builtin_amdgcn_exp2f(x); };
        // ---- the books: one float64 add and one 4-byte LDS store per 128-product tile.  The sums of a pair are produced at the
        // end of a stream() and booked inside the next one, behind its first MFMA (`filler`).  (Measured: the same time as closing
        // them right where they are produced, profiles/r6/ab_call7_tp_probe.jsonl; so is consuming an exp a slot later, and plain
        // v_add_f32 instead of v_pk_add_f32 is 4 % SLOWER — 16 more vector instructions per pair: ab_call8_tp_probe.jsonl.  The
        // loop is bound by its vector instruction count: 32 v_exp_f32 at ~8 cycles + ~24 others per pair and wave next to 8 MFMAs
        // of 32 cycles, the two waves of a SIMD taking turns.) ----
        double run_pref = 0.0;     // running prefix of the exp-sums (every lane of the user holds it)
        float wlo = 0.0f;
        auto book_even = [&](float s0, float s1) {            // first pair of a tile
            if (RG_TP_ABL & 32) { wlo += s0 + s1; return; }
            const float s = s0 + s1;
            wlo = s + swap32(s);
        };
        auto book_odd = [&](uint32_t ti, float s0, float s1) {    // second pair of tile ti: the tile is complete
            if (RG_TP_ABL & 32) { wlo += s0 + s1; return; }
            float s = s0 + s1;
            s += swap32(s);
            // (fp32 inside the tile: <= 10 roundings per term from the exp to here, part of the fixed budget; the prefix itself
            // float64, stored as one fp32 rounding: rho)
            run_pref += static_cast<double>(wlo + s);
            if (h == 0) trow[ti] = static_cast<float>(run_pref);
        };
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        // One pair (k_draw_bf16p's stream): MFMAs of (co) into (a0, a1), which already hold the pair's mu | exp-sum of
        // (p0, p1) -> (s0, s1) | A rows of pair pi_next -> no, its mu -> (p0, p1) once their exps are done
        auto stream = [&](const PairOps& co, PairOps& no, uint32_t pi_next, f32x16& a0, f32x16& a1,
                          f32x16& p0, f32x16& p1, float& s0, float& s1, auto&& filler) {
            f32x2 x0[4], x1[4];
            const char* ab = a_base(pi_next);
            const char* mb = m_base(pi_next);
            RG_PIN();
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                a0 = mm(co.A0[m], Bm[m], a0);
                if (m == 0) filler();          // (the books of the pair before: independent of this pair's MFMAs and exps)
                if (m < EXS) {
                    asm volatile("" : "+v"(p0));                        // (exps may not float above this slot)
#pragma unroll
                    for (int i = (2 * m) * (2 * N1) / (2 * EXS); i < (2 * m + 1) * (2 * N1) / (2 * EXS); ++i) load_a(no, ab, i);
#pragma unroll
                    for (int e = (m * 8 / EXS) * 2; e < ((m + 1) * 8 / EXS) * 2; e += 2) {       // exps in pairs
                        f32x2 y = {ex2(p0[e]), ex2(p0[e + 1])};
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
                        f32x2 y = {ex2(p1[e]), ex2(p1[e + 1])};
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
        float q0s = 0.0f, q1s = 0.0f;       // sums of the pair before, waiting for their books
        for (; pi + 1 < np; pi += 2) {
            float s0, s1;
            const uint32_t T = pi >> 1;
            // tile barrier: every wave holds tile T's operands (its buffer is refilled with tile T + 2); tile T + 1, whose DMA
            // went out at the last barrier, has landed (no other vector-memory operation is in flight in this loop)
            if (!(RG_TP_ABL & 16)) RG_TILE_BARRIER(0);
            if (T + 2 < n_pt && !(RG_TP_ABL & 16)) fetch_tile(T + 2);
            stream(ob, oa, pi + 1, p0, p1, a0, a1, s0, s1, [&] { if (pi > 1) book_odd((pi - 2) >> 1, q0s, q1s); });   // MFMAs of pair pi | sums of pair pi - 1 | books of pair pi - 2
            q0s = s0; q1s = s1;
            stream(oa, ob, pi + 2, a0, a1, p0, p1, s0, s1, [&] { book_even(q0s, q1s); });                           // MFMAs of pair pi + 1 | sums of pair pi | books of pair pi - 1
            q0s = s0; q1s = s1;
        }
        {   // the last pair (second pair of the last tile), then its own sums
            float s0, s1;
            stream(ob, oa, pi, p0, p1, a0, a1, s0, s1, [&] { if (pi > 1) book_odd((pi - 2) >> 1, q0s, q1s); });      // (operand fetch of a "next" pair: this one again, unused)
            book_even(s0, s1);
#pragma unroll
            for (int r = 0; r < 16; ++r) { p0[r] = ex2(p0[r]); p1[r] = ex2(p1[r]); }
            book_odd(pi >> 1, tree(p0), tree(p1));
        }

        // ---- what the search needs from here: the tile of the draw (a count in LDS) and five numbers; k_pick does the rest ----
        if (RG_TP_ABL & 1) { if (active && h == 0 && run_pref + wlo == 12345.0) d.tp_rec[0] = 1; continue; }
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
        const float pbf = (found_t && cnt) ? trow[cnt - 1u] : 0.0f;                   // A: the prefix at the tile's start
        if (active && h == 0) {
            const double delta = static_cast<double>(d.K + 5) * 5.9604644775390625e-08 * static_cast<double>(Ahat) + delta_fixed;
            TpRec r;
            r.u = u_draw; r.S = Sf; r.pb = pbf; r.q = q; r.dlt = static_cast<float>(delta * 1.000001);      // (rounded up: the budget must not shrink)
            r.ti = found_t ? cnt : 0xFFFFFFFFu; r.pad = 0u;
            *reinterpret_cast<TpRec*>(recp) = r;
            if (found_t) {         // the step's draws BY TILE (and shard: the work item's): k_pick takes 32 of one list at a time
                const uint32_t bs = cnt * d.tp_shards + (tb % d.tp_shards);
                const uint32_t k = atomicAdd(&d.tp_hist[bs], 1u);
                d.tp_order[static_cast<size_t>(bs) * d.tp_cap + k] = pos;
            } else {               // no tile (sums that overflowed, u S beyond the last prefix): float64
                const uint32_t xi = atomicAdd(&d.exact_cnt[t], 1u);
                d.exact_list[xi] = pos;
                d.exact_ref[xi] = q;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_pick — the second half of an unsliced step's organic draws: product, certificate, row.
//
// k_draw_tp left, per draw, the 128-product tile u S falls into and the prefix A at its start.  What is left is the position
// inside the tile: the tile's 128 terms once more.  Recomputed per user from the fp32 table that is 10 KB of L2 traffic and
// 2 560 fused multiply-adds per draw on an eighth of a wave (measured inside the sweep kernel: as slow as the search it replaced,
// profiles/r6/ab_call1_tp.jsonl).  Here the draws of a segment of the organic list are first GROUPED BY TILE (a counting sort in
// LDS: 79 bins at P = 10^4), and a wave takes 32 draws of ONE tile: the tile's split rows are the A operands of all of them
// (18 KB from L2 per 32 draws), the users' omega the B operands — the sweep's own MFMA, bit for bit the sweep's logits — and the
// exps, the in-tile prefix and the first product beyond the remainder are two lanes per user in the accumulator layout
// (rows 8 g + 4 h + r of a chunk in register 4 g + r of lane (user, h)).  The certificate is search_and_emit's
// (cert_correlated) with A = the tile's start.  ~1 400 vector instructions per 32 draws, no scratch.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kPickLists = kTpLists;      // (tile, shard) lists of a step's draws (tile * d.tp_shards + shard)

#ifndef RG_PICK_OCC
#define RG_PICK_OCC 3
#endif
template <int KH, int N1>
__global__ void __launch_bounds__(kBlock, RG_PICK_OCC) k_pick(DevSim d, uint32_t t, uint32_t unused) {
    constexpr int K2 = 2 * KH;
    constexpr uint32_t RSc = 32 * N1 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem_raw);            // [kPickLists] draws per (tile, shard)
    uint32_t* gstart = hist + kPickLists;                              // [kPickLists + 1] first group of the list
    float* om_stage = reinterpret_cast<float*>(gstart + kPickLists + 4);   // [4 waves][32][K2]
    __shared__ uint32_t wave_tot[kBlock / 64];
    __shared__ uint32_t last_flag;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const int j = lane & 31, h = lane >> 5;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    (void)unused;
    // ---- the step's draws per list (k_draw_tp counted and listed them): groups of 32 of one list; exclusive scan of the groups ----
    {
        constexpr uint32_t PER = kPickLists / kBlock;                  // consecutive lists per thread
        uint32_t c[PER], gsum = 0;
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i) { c[i] = d.tp_hist[threadIdx.x * PER + i]; hist[threadIdx.x * PER + i] = c[i]; gsum += (c[i] + 31u) >> 5; }
        uint32_t sc = gsum;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(sc, o); if (lane >= o) sc += y; }
        if (lane == 63) wave_tot[wave] = sc;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += wave_tot[w];
        uint32_t run = base + sc - gsum;
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i) { gstart[threadIdx.x * PER + i] = run; run += (c[i] + 31u) >> 5; }
        if (threadIdx.x == kBlock - 1) gstart[kPickLists] = run;
    }
    __syncthreads();
    const uint32_t n_groups = gstart[kPickLists];
    float* omw = om_stage + wave * 32 * K2;
    // (the next group's list entry / record / omega requested a group ahead: measured, no gain — the kernel is bound by its
    // scattered 128-byte accesses, not by a wave's chain of round trips: profiles/r6/ab_call6_tp_ablation.jsonl, pkabl4)
    for (uint32_t g = blockIdx.x * (kBlock / 64) + wave; g < n_groups; g += gridDim.x * (kBlock / 64)) {
        // the group's list: the last one whose first group is <= g (wave-uniform; lists with no draws have no groups)
        uint32_t lo = 0, hi = kPickLists;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (gstart[mid] <= g) lo = mid; else hi = mid; }
        const uint32_t lst = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lo)));
        const uint32_t tile = lst / d.tp_shards;
        const uint32_t gi = g - gstart[lst];
        const uint32_t cnt = min(32u, hist[lst] - 32u * gi);
        const bool active = static_cast<uint32_t>(j) < cnt;
        const uint32_t pos = d.tp_order[static_cast<size_t>(lst) * d.tp_cap + 32u * gi + (active ? j : 0)];
        const uint32_t slot = cur[pos];
        const char* recp = d.tp_rec + static_cast<size_t>(pos) * tp_rec_stride(KH);
        const TpRec rec = *reinterpret_cast<const TpRec*>(recp);
        // ---- omega32 of the 32 users (behind the record's header) -> the wave's stage -> B fragments [w1 | w1 | w2 | 0 .. | -q] (as
        // the sweep built them) ----
        {
            float* o = omw + j * K2 + h * KH;
            const float* rw = reinterpret_cast<const float*>(recp + sizeof(TpRec)) + h * KH;
#pragma unroll
            for (int s = 0; s < KH; ++s) o[s] = rw[s];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        bf16x8 Bm[N1];
        {
            const uint32_t K = d.K;
            const float* omu = omw + j * K2;
#pragma unroll
            for (int s = 0; s < N1; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t ke = 16 * s + 8 * h + e;
                    unsigned short sp[2] = {0, 0};
                    if (ke < 3 * K) f16_split2(omu[ke % K], sp);
                    Bm[s][e] = static_cast<short>(ke < 2 * K ? sp[0] : sp[1]);
                }
            if (h == 1) Bm[N1 - 1][7] = static_cast<short>(__builtin_bit_cast(unsigned short, static_cast<_Float16>(-rec.q)));
        }
        __builtin_amdgcn_wave_barrier();                                   // (the stage is rewritten by the next group)
        const double S = static_cast<double>(rec.S), pb = static_cast<double>(rec.pb);
        const float rems = static_cast<float>(rec.u * S - pb);
        // ---- the tile's four chunks: logits on the matrix cores, then two lanes per user in the accumulator layout ----
        const uint32_t cpt = d.tp_cpt;                                     // chunks of a list tile (4; the wide sweep's super-tiles: more)
        const uint32_t c_first = tile * cpt;
        const int c_n = static_cast<int>(min(cpt, d.n_chunks - c_first));
        const char* a_lane = reinterpret_cast<const char*>(d.gsplit) + (static_cast<size_t>(c_first) * 32 + j) * RSc + 16 * h;
        const float* mu_lane = d.mu32s + static_cast<size_t>(c_first) * 32 + 4 * h;
        int r_idx = -1;
        float r_a = 0.0f, r_b = 0.0f, off = 0.0f;
        // (a chunk's operands are requested a chunk ahead: the loop is a chain of L2 round trips otherwise)
        using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
        f32x16 acc_n;
        bf16x8 A_n[N1];
        auto fetch_chunk = [&](int c) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 m = *reinterpret_cast<const float4*>(mu_lane + 32 * c + 8 * gq);
                acc_n[4 * gq] = m.x; acc_n[4 * gq + 1] = m.y; acc_n[4 * gq + 2] = m.z; acc_n[4 * gq + 3] = m.w;
            }
#pragma unroll
            for (int m = 0; m < N1; ++m) A_n[m] = *reinterpret_cast<const bf16x8*>(a_lane + static_cast<size_t>(c) * 32 * RSc + 32 * m);
        };
        fetch_chunk(0);
#pragma unroll 1
        for (int c = 0; c < ((RG_PICK_ABL & 2) ? 0 : c_n); ++c) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A_n[0]), __builtin_bit_cast(f16x8, Bm[0]), acc_n, 0, 0, 0);
#pragma unroll
            for (int m = 1; m < N1; ++m)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A_n[m]), __builtin_bit_cast(f16x8, Bm[m]), acc, 0, 0, 0);
            if (c + 1 < c_n) fetch_chunk(c + 1);   // (the MFMAs above have read their operands: the registers take the next chunk's)
            // in-group inclusive prefixes (a group = 4 consecutive products: rows 8 gq + 4 h .. + 3), the groups' sums
            float p[16], sg[4], so[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                p[4 * gq] = __builtin_amdgcn_exp2f(acc[4 * gq]);
                p[4 * gq + 1] = p[4 * gq] + __builtin_amdgcn_exp2f(acc[4 * gq + 1]);
                p[4 * gq + 2] = p[4 * gq + 1] + __builtin_amdgcn_exp2f(acc[4 * gq + 2]);
                p[4 * gq + 3] = p[4 * gq + 2] + __builtin_amdgcn_exp2f(acc[4 * gq + 3]);
                sg[gq] = p[4 * gq + 3];
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) so[gq] = swap32(sg[gq]);
            // product order alternates the two lanes: group 2 gq + h; `run` = the sum of the groups before this lane's group gq
            float run = off;
            int ci = -1;
            float ca = 0.0f, cb = 0.0f;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float base = h ? run + so[gq] : run;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = base + p[4 * gq + r];
                    if (ci < 0 && x > rems) { ci = 32 * c + 8 * gq + 4 * h + r; cb = x; ca = r ? base + p[4 * gq + r - 1] : base; }
                }
                run += h ? so[gq] + sg[gq] : sg[gq] + so[gq];                 // (the same value on both lanes: lane 0's group first)
            }
            off = run;
            // the user's first hit: the smaller product index of its two lanes'
            const int oi = __builtin_bit_cast(int, swap32(__builtin_bit_cast(float, ci)));
            const float oa = swap32(ca), ob = swap32(cb);
            if (r_idx < 0) {
                if (ci >= 0 && (oi < 0 || ci < oi)) { r_idx = ci; r_a = ca; r_b = cb; }
                else if (oi >= 0) { r_idx = oi; r_a = oa; r_b = ob; }
            }
            if (__ballot(active && r_idx < 0) == 0ull) break;                // every user of the group has its product
        }
        const uint32_t v = c_first * 32u + static_cast<uint32_t>(max(r_idx, 0));
        // (S, pb: fp32 roundings of the sweep's float64 running prefix; a, b: fp32 sums of the tile's terms before / with v)
        const CertLin ct = cert_correlated(S, pb, static_cast<double>(r_a), static_cast<double>(r_b), static_cast<double>(rec.dlt));
        const bool ok = r_idx >= 0 && v < d.P && ct.valid &&
                        (v == 0 || rec.u * ct.den_lo > ct.num_lo) &&
                        (v == d.P - 1 || rec.u * ct.den_hi < ct.num_hi);
        if (active && h == 0 && !((RG_PICK_ABL & 1) && v != 0x7fffffffu)) {
            if (ok) {
                const uint32_t user = static_cast<uint32_t>(d.first_user + d.uid[slot]);
                write_organic_row(d, t, pos, slot, user, v);
                if (d.hist_cap) history_add(d, slot, v);
            } else {
                const uint32_t xi = atomicAdd(&d.exact_cnt[t], 1u);
                d.exact_list[xi] = pos;
                d.exact_ref[xi] = rec.q;
            }
        }
    }
    // the last block to finish leaves the counters empty for the next step's sweep (every block read them before its loop)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last_flag = atomicAdd(&d.tp_hist[kPickLists], 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (last_flag) {
        for (uint32_t i = threadIdx.x; i <= kPickLists; i += kBlock) d.tp_hist[i] = 0u;
        __threadfence();
    }
}

draw_kernel_t tp_kernel_for(const DevSim& d) {
    if (!d.f16 || d.wide) return nullptr;
#define RG_CASE(kh, a) if (d.KH == kh && d.N1 == a) return k_draw_tp<kh, a>;
    RG_CASE(4, 1) RG_CASE(4, 2) RG_CASE(10, 2) RG_CASE(10, 3) RG_CASE(10, 4)
#undef RG_CASE
    return nullptr;
}
draw_kernel_t pick_kernel_for(const DevSim& d) {
    if (!d.f16) return nullptr;
#define RG_CASE(kh, a) if (d.KH == kh && d.N1 == a) return k_pick<kh, a>;
    RG_CASE(4, 1) RG_CASE(4, 2) RG_CASE(10, 2) RG_CASE(10, 3) RG_CASE(10, 4)
    RG_CASE(16, 7) RG_CASE(32, 7) RG_CASE(32, 10) RG_CASE(32, 13)
#undef RG_CASE
    return nullptr;
}


// ------------------------------------------------------------------------------------------
// k_draw_tpw — the same sweep for WIDE embeddings (21 < K <= 64: BASELINE config 4's K = 64; replaces k_draw_f16w on unsliced
// steps without a per-user cache).
//
// At N1 = 13 k-steps per chunk a 32x32x16 MFMA needs 1 KB of A operand from LDS per 32 matrix-pipe cycles and SIMD: exactly the
// CU's LDS bandwidth, so k_draw_f16w's 8 waves x 32 users are LDS-bound as much as MFMA-bound (and measured at the SUM of its
// MFMA, LDS and vector time: 4 100 cycles per 64-product tile against 1 664 of MFMA).  Its variant with two user groups per wave
// — every A fragment feeds two MFMAs, half the LDS traffic — lost because the per-chunk scratch stores kept spilled pointers in the
// loop, and every reload waits for the tile DMA in flight (profiles/r6/ab_call10_wide_step0.txt).  Here the loop has no global
// memory operation but the tile DMA: ONE wave per SIMD, 64 users per wave (2 groups of 32: an A fragment from the LDS ring feeds
// two MFMAs), 256 users per block; the books are a float64 running prefix per user, stored to LDS once per SUPER-TILE of
// `tp_cpt / 2` tiles (a user's <= ~70 prefixes: what fits beside three 27 KB tile buffers); the draw's super-tile is a count in
// LDS at the end, k_pick finds the product inside it on the matrix cores (tp_cpt chunks instead of 4).  Two accumulator sets
// alternate between the tile being multiplied and the tile being exp-summed: no copies.
// ------------------------------------------------------------------------------------------
template <int KH, int N1>
__global__ void __launch_bounds__(kBlock, 1) k_draw_tpw(DevSim d, uint32_t t, uint32_t NTs) {
    constexpr uint32_t RSc = 32 * N1 + 16, TILE_B = 64 * RSc, NT = (TILE_B + 1023) / 1024, NB = 3;
    constexpr int NW = 4, UG = 2, K2 = 2 * KH;
    using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // [tile buffer 0][mu x 3][tile buffers 1, 2][prefixes]; omega32 of the block's 256 users is staged over buffers 1, 2 and the
    // prefix rows while the B rows are built (before their first use)
    char* g_buf0 = smem_raw;
    float* mu_buf = reinterpret_cast<float*>(smem_raw + TILE_B);          // [3][64]
    char* g_buf12 = smem_raw + TILE_B + 1024;
    float* tpref = reinterpret_cast<float*>(g_buf12 + 2 * TILE_B);        // [256 users][NTs]
    float* om_stage = reinterpret_cast<float*>(g_buf12);                  // [256][K2]
    auto buf_of = [&](uint32_t b) -> char* { return b == 0 ? g_buf0 : g_buf12 + (b - 1) * TILE_B; };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const int j = lane & 31, h = lane >> 5;
    const uint32_t n_o = d.step_cnt[2 * t + RG_STATE_ORGANIC];
    const uint32_t n_items = (n_o + 255) / 256;
    const uint32_t* cur = list_ptr(d, t & 1, RG_STATE_ORGANIC);
    const float mumax = d.stats[2 * KH + 1], g2max = d.stats[2 * KH];
    const uint32_t n_t = d.n_chunks / 2;                                  // 64-product tiles
    const uint32_t G = d.tp_cpt / 2;                                      // tiles per stored prefix
    const uint32_t n_s = (n_t + G - 1) / G;                               // super-tiles
    const int my_dma = static_cast<int>((NT - wave + NW - 1) / NW) + (wave == NW - 1 ? 1 : 0);
    float* trow_own = tpref + static_cast<size_t>((wave * UG + h) * 32 + j) * NTs;     // lane (j, h) keeps the books of group h's user j

    for (uint32_t tb = blockIdx.x; tb < n_items; tb += gridDim.x) {
        uint32_t pos[UG], slot[UG];
        bool active[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            pos[g] = tb * 256 + (wave * UG + g) * 32 + j;
            active[g] = pos[g] < n_o;
            slot[g] = active[g] ? cur[pos[g]] : 0u;
        }
        __syncthreads();           // every wave is done with the LDS buffers (previous work item)
        const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
        const rg_v4i rs_g = raw_buffer_rsrc(d.gsplit), rs_m = raw_buffer_rsrc(d.mu32s);
        const uint32_t lds0 = lds_addr_of(g_buf0), lds12 = lds_addr_of(g_buf12), mu_lds = lds_addr_of(mu_buf);
        auto fetch_tile = [&](uint32_t ti) {
            const uint32_t b = ti % NB;
            const uint32_t base = b == 0 ? lds0 : lds12 + (b - 1) * TILE_B;
            for (uint32_t off = static_cast<uint32_t>(wave) * 1024u; off < TILE_B; off += NW * 1024u)
                dma_to_lds_b128(rs_g, base + off, lane16, ti * TILE_B + off);
            if (wave == NW - 1 && lane < 16) dma_to_lds_b128(rs_m, mu_lds + b * 256u, lane16, ti * 256u);
        };
        fetch_tile(0);
#if (RG_TPW_ABL & 64)
        const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- omega32 of the users -> LDS stage (also the logit error bound) and behind the draw's record (k_pick's B operands) ----
        float Ahat[UG];
        double delta_fixed[UG];
        float* omu[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            omu[g] = om_stage + ((wave * UG + g) * 32 + j) * K2;
            float* rw = reinterpret_cast<float*>(d.tp_rec + static_cast<size_t>(active[g] ? pos[g] : 0u) * tp_rec_stride(KH) + sizeof(TpRec)) + h * KH;
            float absdot = 0.0f, sq = 0.0f, absw = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < KH; ++s2) {
                const uint32_t k = h * KH + s2;
                float w = 0.0f;
                if (active[g] && k < d.K) w = static_cast<float>(d.omega[static_cast<size_t>(slot[g]) * d.OMS + k]);
                omu[g][k] = w;
                if (active[g]) rw[s2] = w;
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
        float q[UG] = {0.0f, 0.0f};
        // Register plan (512 per lane at one wave per SIMD, but the vector ALU only reads the 256 architectural ones): both
        // accumulator sets in VGPRs (the exps read them), the B rows and the A ring — MFMA operands only — in AGPRs.  The
        // allocator does not find that split by itself (228 spilled registers, a third of them inside the tile loop): pinned here.
        auto mm = [](const bf16x8& a, const bf16x8& b, const f32x16& c) -> f32x16 {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        };
        auto load_mu = [&](f32x16& acc, const char* mb, int which) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float4 m = *reinterpret_cast<const float4*>(mb + 128 * which + 32 * qq);
                acc[4 * qq] = m.x; acc[4 * qq + 1] = m.y; acc[4 * qq + 2] = m.z; acc[4 * qq + 3] = m.w;
            }
        };
        // ---- the books: a float64 running prefix per user, one fp32 rounding of it to LDS per super-tile ----
        double run_pref[UG] = {0.0, 0.0};
        const uint32_t n_c = d.n_chunks, cps = d.tp_cpt;                  // chunks; chunks per super-tile
        uint32_t st_left = cps, st_idx = 0;                              // chunks left in the super-tile being summed; its index
        auto book = [&](uint32_t ci_done, float s0, float s1) {          // s_g: group g's exp-sum of chunk ci_done (this lane's half)
            if (RG_TPW_ABL & 8) { run_pref[0] += static_cast<double>(s0 + s1); return; }
            s0 += swap32(s0);
            s1 += swap32(s1);
            run_pref[0] += static_cast<double>(s0);
            run_pref[1] += static_cast<double>(s1);
            if (--st_left == 0 || ci_done + 1 == n_c) {                  // (counters, not ci_done % cps: an integer division per chunk is ~20 scalar instructions in a one-wave stream)
                trow_own[st_idx] = static_cast<float>(h ? run_pref[1] : run_pref[0]);
                ++st_idx; st_left = cps;
            }
        };
        RG_DMA_WAIT();
        __syncthreads();           // tile 0 landed; every wave has built its B rows from the stage (which tiles 1, 2 and the prefix rows now overwrite)
        for (uint32_t i = n_s; i < NTs; ++i) trow_own[i] = INFINITY;     // (row padding: never counted)
        if (1 < n_t) fetch_tile(1);                      // (tile 2 goes out at the barrier inside chunk 1)
        const char* a_lane0 = g_buf0 + j * RSc + 16 * h;
        {   // first chunk with reference 0: its max (an integer after ceil, exact in one fp16 piece) becomes the reference
#pragma unroll
            for (int g = 0; g < UG; ++g) {
                f32x16 y;
                load_mu(y, reinterpret_cast<const char*>(mu_buf) + 16 * h, 0);
#pragma unroll
                for (int s2 = 0; s2 < N1; ++s2) y = mm(*reinterpret_cast<const bf16x8*>(a_lane0 + 32 * s2), Bm[g][s2], y);
                float cm = y[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) cm = fmaxf(cm, y[r]);
                float qn = fmaxf(ceilf(fmaxf(cm, swap32(cm))), -1.0e30f);
                qn = fminf(fmaxf(qn, -2047.0f), 2047.0f);
                q[g] = qn;
                if (h == 1) Bm[g][N1 - 1][7] = static_cast<short>(__builtin_bit_cast(unsigned short, static_cast<_Float16>(-qn)));
            }
        }
        // One CHUNK (32 products x the wave's 64 users): its 2 N1 MFMAs into `ac` — every A fragment of the ring feeds both user
        // groups — and, in the issue slots behind those MFMAs, everything else: the exps and sums of the chunk before (in `pv`), its
        // books, then the NEXT chunk's mu seeds into `pv` (whose exps are done) and the next chunk's first A fragments into the
        // ring; the tile barrier sits in the middle of a tile's second chunk.  Nothing is left outside the MFMA stream: with one
        // wave per SIMD whatever a chunk step did before its first or after its last MFMA ran with the matrix pipe idle — 40 % of
        // the first form's time (profiles/r6/ab_call13_tpw_ablation.txt: MFMA 0.66 us of 1.96 per tile, the rest additive).
        // The pipeline is a chunk deep, not a tile: two accumulator sets of 2 x 16 registers, so that B rows (8 N1 registers),
        // accumulators, ring and sums fit the 256 architectural registers.
        constexpr int NSLOT = UG * N1, NEP = 8 * UG, EPS = NSLOT >= NEP + 9 ? 1 : (NSLOT >= NEP / 2 + 9 ? 2 : 4), ESL = NEP / EPS;     // exp PAIRS per slot; slots that carry exps
        static_assert(ESL + 9 <= NSLOT, "a chunk's slots must hold its exps, books and the next chunk's seeds");
        constexpr int RD = 4;                                          // A ring: k-steps read ahead of the MFMAs
#ifdef RG_TEST_TPW_LATE_BARRIER     // (the bug put back, for tests/test_hip_parity.py::test_two_runs_of_a_bench_shape_give_the_same_log)
        constexpr int BAR_S2 = N1 - RD + 1;
#else
        constexpr int BAR_S2 = ((ESL + 1) / UG) < (N1 - RD + 1) ? ((ESL + 1) / UG) : (N1 - RD + 1);     // the tile barrier's k-step in a tile's second chunk
#endif
        auto load_mu_q = [&](f32x16& acc, const char* mb, int qq) {
            const float4 m = *reinterpret_cast<const float4*>(mb + 32 * qq);
            acc[4 * qq] = m.x; acc[4 * qq + 1] = m.y; acc[4 * qq + 2] = m.z; acc[4 * qq + 3] = m.w;
        };
        bf16x8 Ar[RD];
        auto chunk_step = [&](auto roff_, auto odd_, auto hp_, auto hn_, uint32_t ci, uint32_t b, f32x16 (&ac)[UG], f32x16 (&pv)[UG]) {
            constexpr int ROFF = decltype(roff_)::value;               // ring slot of this chunk's k-step 0
            constexpr bool ODD = decltype(odd_)::value;                // second chunk of its tile
            constexpr bool have_p = decltype(hp_)::value;              // there is a chunk before (its logits in `pv`)
            constexpr bool has_next = decltype(hn_)::value;            // ... and one behind
            const uint32_t ti = ci >> 1;
            // (the lane's row / half are REBUILT here from the lane id: kept across the loop they were spilled, and a reload inside
            // the loop is followed by s_waitcnt vmcnt(0), which also waits for the tile DMA in flight)
            const int ln = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
            int jl = ln & 31, hl = ln >> 5;
            asm volatile("" : "+v"(jl), "+v"(hl));
            const uint32_t bn = ODD ? (b == NB - 1 ? 0u : b + 1u) : b;           // (b = ti % NB, carried by the caller)
            const char* ab = buf_of(b) + ((ODD ? 32 : 0) + jl) * RSc + 16 * hl;
            const char* abn = buf_of(bn) + ((ODD ? 0 : 32) + jl) * RSc + 16 * hl;            // the next chunk's rows
            const char* mbn = reinterpret_cast<const char*>(mu_buf) + 16 * hl + bn * 256u + (ODD ? 0u : 128u);
            f32x2 x[UG][4];
#if RG_TPW_SCALAR_ADD
            float xs[UG][8];
            auto addf = [](float a, float b2) -> float { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b2)); return r; };
#endif
            auto slot = [&](int i) {
                if (i < ESL) {
                    if (have_p) {
#pragma unroll
                        for (int e = i * EPS; e < (i + 1) * EPS; ++e) {
                            const int g = e >> 3, r = e & 7;              // accumulator of group g, register pair r
                            asm volatile("" : "+v"(pv[g]));
                            f32x2 y = {(RG_TPW_ABL & 1) ? pv[g][2 * r] * 0.5f : __builtin_amdgcn_exp2f(pv[g][2 * r]),
                                       (RG_TPW_ABL & 1) ? pv[g][2 * r + 1] * 0.5f : __builtin_amdgcn_exp2f(pv[g][2 * r + 1])};
                            asm volatile("" : "+v"(y));
#if RG_TPW_SCALAR_ADD
                            // (same sums, same order per component as the packed form: x[r & 3] component-wise)
                            if (r < 4) { xs[g][2 * r] = y[0]; xs[g][2 * r + 1] = y[1]; }
                            else { xs[g][2 * (r & 3)] = addf(xs[g][2 * (r & 3)], y[0]); xs[g][2 * (r & 3) + 1] = addf(xs[g][2 * (r & 3) + 1], y[1]); }
#else
                            if (r < 4) x[g][r] = y; else x[g][r & 3] += y;
#endif
                        }
                    }
                } else if (i == ESL) {
                    if (have_p) {
                        float sm[UG];
#pragma unroll
                        for (int g = 0; g < UG; ++g) {
#if RG_TPW_SCALAR_ADD
                            const float z0 = addf(addf(xs[g][0], xs[g][4]), addf(xs[g][2], xs[g][6]));
                            const float z1 = addf(addf(xs[g][1], xs[g][5]), addf(xs[g][3], xs[g][7]));
                            sm[g] = addf(z0, z1);
#else
                            const f32x2 z = (x[g][0] + x[g][2]) + (x[g][1] + x[g][3]);
                            sm[g] = z[0] + z[1];
#endif
                        }
                        book(ci - 1, sm[0], sm[1]);
                    }
                } else if (i <= ESL + 4) {
                    if (has_next && !(RG_TPW_ABL & 16)) load_mu_q(pv[0], mbn, i - ESL - 1);
                } else if (i <= ESL + 8) {
                    if (has_next && !(RG_TPW_ABL & 16)) load_mu_q(pv[1], mbn, i - ESL - 5);
                }
            };
            RG_PIN();
#pragma unroll
            for (int s2 = 0; s2 < N1; ++s2) {
                if (ODD && s2 == BAR_S2 && has_next && !(RG_TPW_ABL & 4)) {
                    // the next chunk opens tile ti + 1: it has landed once this wave's DMA (issued a tile ago) is done; every wave is
                    // past tile ti - 1, whose buffer takes tile ti + 2.  The barrier sits BEFORE the first slot that touches tile
                    // ti + 1 — its mu seeds (slot ESL + 1), then its first A fragments (k-step N1 - RD + 1).  (Until call 53 it sat
                    // at k-step N1 - RD + 1 only: the first mu quads of the next tile — DMA'd by the block's LAST wave — were read
                    // up to three slots before it; almost always landed, ~300 wrong draws in the 2.8 10^7 of a C4 shard when not:
                    // profiles/r6/determinism_call51.jsonl)
                    RG_TILE_BARRIER(0);
                    if (ti + 2 < n_t) fetch_tile(ti + 2);
                }
                {
                    constexpr int dummy = 0; (void)dummy;
                    const int qn = s2 + RD - 1;
                    if (!(RG_TPW_ABL & 32)) {
                        if (qn < N1) Ar[(ROFF + qn) % RD] = *reinterpret_cast<const bf16x8*>(ab + 32 * qn);
                        else if (has_next) Ar[(ROFF + qn) % RD] = *reinterpret_cast<const bf16x8*>(abn + 32 * (qn - N1));
                    }
                }
#pragma unroll
                for (int g = 0; g < UG; ++g) {
                    if (!(RG_TPW_ABL & 2)) ac[g] = mm(Ar[(ROFF + s2) % RD], Bm[g][s2], ac[g]);
                    slot(s2 * UG + g);
                    RG_PIN();
                }
            }
        };
        f32x16 accA[UG], accB[UG];
        {   // chunk 0's seeds and first A fragments
            const char* mb0 = reinterpret_cast<const char*>(mu_buf) + 16 * h;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) { load_mu_q(accA[0], mb0, qq); load_mu_q(accA[1], mb0, qq); }
#pragma unroll
            for (int s2 = 0; s2 < RD - 1 && s2 < N1; ++s2) Ar[s2] = *reinterpret_cast<const bf16x8*>(a_lane0 + 32 * s2);
        }
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, N1 % RD>;
        using I2 = std::integral_constant<int, (2 * N1) % RD>; using I3 = std::integral_constant<int, (3 * N1) % RD>;
        // (n_chunks is a multiple of 4: four chunk steps bring the ring back to slot 0; first / last group: compile-time flags, so
        // that no chunk step carries a branch per slot)
        using T_ = std::true_type; using F_ = std::false_type;
        uint32_t b0 = 0;                                 // LDS buffer of the group's first tile (tile index mod NB)
        auto group = [&](uint32_t ci, auto first_, auto last_) {
            constexpr bool FIRST = decltype(first_)::value, LAST = decltype(last_)::value;
            const uint32_t b1 = b0 == NB - 1 ? 0u : b0 + 1u;
            chunk_step(I0{}, F_{}, std::bool_constant<!FIRST>{}, T_{}, ci, b0, accA, accB);
            chunk_step(I1{}, T_{}, T_{}, T_{}, ci + 1, b0, accB, accA);
            chunk_step(I2{}, F_{}, T_{}, T_{}, ci + 2, b1, accA, accB);
            chunk_step(I3{}, T_{}, T_{}, std::bool_constant<!LAST>{}, ci + 3, b1, accB, accA);
            b0 = b1 == NB - 1 ? 0u : b1 + 1u;
        };
        if (n_c == 4) group(0, T_{}, T_{});
        else {
            group(0, T_{}, F_{});
            uint32_t ci = 4;
            for (; ci + 4 < n_c; ci += 4) group(ci, F_{}, F_{});
            group(ci, F_{}, T_{});
        }
        {   // the last chunk's own sums
            float sm[UG];
#pragma unroll
            for (int g = 0; g < UG; ++g) {
                float e0 = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) e0 += __builtin_amdgcn_exp2f(accB[g][r]);
                sm[g] = e0;
            }
            book(n_c - 1, sm[0], sm[1]);
        }
        // ---- the draw's super-tile (a count in LDS) and what k_pick needs ----
#if (RG_TPW_ABL & 64)
        // the shader clock under this kernel's load: cycles (s_memtime) per 100 MHz tick (s_memrealtime) over one work item's sweep
        if (blockIdx.x == 7 && threadIdx.x == 0 && tb == blockIdx.x) {
            const unsigned long long c1 = __builtin_amdgcn_s_memtime() - clk0, r1 = __builtin_amdgcn_s_memrealtime() - rt0;
            printf("k_draw_tpw work item: %llu shader cycles in %llu ticks of 100 MHz = %.0f MHz; %u tiles: %.0f cycles per tile\n", c1, r1,
                   100.0 * static_cast<double>(c1) / static_cast<double>(r1), n_t, static_cast<double>(c1) / n_t);
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            const uint32_t uidx = d.uid[slot[g]];
            const uint32_t user = static_cast<uint32_t>(d.first_user + uidx);
            const double u_draw = organic_uniform(d, uidx, user, t);
            const float Sf = static_cast<float>(run_pref[g]);
            const double S = static_cast<double>(Sf);
            const double tau = u_draw * S;
            float tf = static_cast<float>(tau);
            if (static_cast<double>(tf) > tau) tf = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, tf) - 1u);   // (tau >= 0)
            const float* trow = tpref + static_cast<size_t>((wave * UG + g) * 32 + j) * NTs;
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
            const bool found_t = cnt < n_s && S > 0.0 && S < 3.0e38;
            const float pbf = (found_t && cnt) ? trow[cnt - 1u] : 0.0f;
            if (active[g] && h == 0) {
                const double delta = static_cast<double>(d.K + 5) * 5.9604644775390625e-08 * static_cast<double>(Ahat[g]) + delta_fixed[g];
                TpRec r;
                r.u = u_draw; r.S = Sf; r.pb = pbf; r.q = q[g]; r.dlt = static_cast<float>(delta * 1.000001);
                r.ti = found_t ? cnt : 0xFFFFFFFFu; r.pad = 0u;
                *reinterpret_cast<TpRec*>(d.tp_rec + static_cast<size_t>(pos[g]) * tp_rec_stride(KH)) = r;
                if (found_t) {
                    const uint32_t bs = cnt * d.tp_shards + (tb % d.tp_shards);
                    const uint32_t k = atomicAdd(&d.tp_hist[bs], 1u);
                    d.tp_order[static_cast<size_t>(bs) * d.tp_cap + k] = pos[g];
                } else {
                    const uint32_t xi = atomicAdd(&d.exact_cnt[t], 1u);
                    d.exact_list[xi] = pos[g];
                    d.exact_ref[xi] = q[g];
                }
            }
        }
    }
}

draw_kernel_t tpw_kernel_for(const DevSim& d) {
    if (!d.f16 || !d.wide) return nullptr;
#define RG_CASE(kh, a) if (d.KH == kh && d.N1 == a) return k_draw_tpw<kh, a>;
    RG_CASE(16, 7) RG_CASE(32, 7) RG_CASE(32, 10) RG_CASE(32, 13)
#undef RG_CASE
    return nullptr;
}

}  // namespace rgk
