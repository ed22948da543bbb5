// rg_draw_wide.hip — librecogym_hip.so, unit 5 of 7: the wide-K sweep (k_draw_f16w, 21 < K <= 64).
// (see rg_common.hpp for the shared types and helpers, DESIGN.md for the data layout and the rooflines)

#include "rg_common.hpp"

namespace rgk {

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

draw_kernel_t f16w_kernel_for(const DevSim& d) {
    const int ug = f16w_ug();
#define RG_CASE(kh, a) if (d.KH == kh && d.N1 == a) return ug == 2 ? k_draw_f16w<kh, a, 2> : k_draw_f16w<kh, a, 1>;
    RG_CASE(16, 7) RG_CASE(32, 7) RG_CASE(32, 10) RG_CASE(32, 13)
#undef RG_CASE
    return nullptr;
}

}  // namespace rgk
