// rg_draw_pipelined.hip — librecogym_hip.so, unit 4 of 7: the pipelined 16-bit sweep (k_draw_bf16p) and the per-user cache kernels (k_cache_finalize, k_draw_cached).
// (see rg_common.hpp for the shared types and helpers, DESIGN.md for the data layout and the rooflines)

#include "rg_common.hpp"

namespace rgk {

template <int KH, int N1, int N2, int N3, bool F16>
__global__ void __launch_bounds__(kBlock, RG_SWEEP_OCC) k_draw_bf16p(DevSim d, uint32_t t, uint32_t S) {
    constexpr uint32_t NB = RG_SWEEP_NB;       // product tiles in LDS: the one in use and NB - 1 in flight
    constexpr int NM = F16 ? N1 : N1 + N2 + N3;   // MFMAs per chunk (fp16 two-way split: one group)
    // MFMA slots that carry the exps (and the A loads); the rest carry the mu loads.  The fp16 form is VALU-bound:
    // its exps spread over all slots but the last
    constexpr int EXS = F16 ? (NM > 1 ? NM - 1 : 1) : (NM > 3 ? NM - 3 : 1);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t tile_b = d.TPB * d.RS;                             // bytes per split tile
    char* g_buf = smem_raw;                                           // [NB][TPB][RS]: the tile in use and two (NB = 3) in flight
    float* mu_buf = reinterpret_cast<float*>(g_buf + NB * tile_b);    // [NB][TPB] (+ pad)
    float* om_stage = mu_buf + NB * d.TPB + 64;                       // [4 waves][32 users][2KH] omega32
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
                dma_to_lds_b128(rs_g, g_lds + ((ti - pt_lo) % NB) * TB + off, lane16, ti * TB + off);
            if (wave == 3 && lane < 32) dma_to_lds_b128(rs_m, mu_lds + ((ti - pt_lo) % NB) * 512u, lane16, ti * 512u);
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
        auto a_base = [&](uint32_t pi) { return a_lane + ((pi >> 1) % NB) * TILE_B + (pi & 1) * (64 * RSc); };
        auto m_base = [&](uint32_t pi) { return m_lane + ((pi >> 1) % NB) * (128 * 4) + (pi & 1) * (64 * 4); };
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
        if (NB > 2 && pt_lo + 2 < pt_hi) fetch_tile(pt_lo + 2);
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
            if (NB > 2) {
                if (T + 2 >= pt_hi) RG_TILE_BARRIER(0);            // nothing was issued behind tile T + 1 but stores
                else if (pi == 1) RG_TILE_BARRIER(4);               // DMA(T + 2)
                else if (pi == 3) RG_TILE_BARRIER(5);               // + one store
                else RG_TILE_BARRIER(6);                            // + two stores
                if (T + 3 < pt_hi && !RG_SWEEP_ABL(32u)) fetch_tile(T + 3);
            } else {
                // two buffers: tile T + 1's DMA went out at the last barrier (tile T's buffer is refilled with tile T + 2 now);
                // behind it this wave issued the scratch store of the tile finished since, nothing else
                if (T + 1 >= pt_hi || pi == 1) RG_TILE_BARRIER(0);
                else RG_TILE_BARRIER(1);
                if (T + 2 < pt_hi && !RG_SWEEP_ABL(32u)) fetch_tile(T + 2);
            }
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
            d.walk_hot[urow * 32 + 31] = kRhoLoose;
        }
        if (S == 1 && !RG_SWEEP_ABL(128u) && !d.sweep_only) search_and_emit<KH>(d, t, scr, scr_chunk, omu, Ahat, n_resc, active, pos, slot, j, h, true, delta_fixed, &view);
    }
}

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
        float Q = -INFINITY, qabs = 0.0f;
#pragma unroll
        for (uint32_t sc = 0; sc < kMaxSC; ++sc) if (sc < d.n_sc) { Q = fmaxf(Q, rec[sc].y); qabs = fmaxf(qabs, fabsf(rec[sc].y)); }
        if (d.XNH) {
            // the sums are k_sweep_xh's (a user whose reference moved, or a run with the finalize output left to this kernel): its own
            // representation / residual / join terms on top of the accumulation budget above (xh_delta's ineligible branch), with the
            // residual bound taken at its largest (|omega - whi| <= 2^-9) — ADVICE round 5
            const float glomax = d.xstats[2 * KH];
            float egam = 0.0f, lob = d.xstats[2 * KH + 1];
#pragma unroll
            for (int k = 0; k < K2; ++k) {
                const float wf = fabsf(om[k]) * 1.0000002f, r1f = 0.001953125f * 1.002f;
                egam = fmaf(wf, d.xstats[k], egam);
                lob = fmaf(r1f, d.stats[k] * kLog2e * 1.000001f + 0.00390625f, lob);
                lob = fmaf(wf + r1f, glomax, lob);
            }
            const double e24 = 5.9604644775390625e-08;
            const double e_lo = (16.0 * d.XNL + 4.0) * e24 * (static_cast<double>(lob) + 1.6e-5);
            const double e_x = e24 * (static_cast<double>(Ahat) * 1.4426950408889634 + static_cast<double>(qabs)) * 1.01;
            delta += 0.6931471805599453 * (static_cast<double>(egam) + static_cast<double>(d.K) * 4.0e-9 + e_lo + e_x);
        }
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

}  // namespace rgk
