// rg_draw_fp32.hip — librecogym_hip.so, unit 3 of 7: the fp32-MFMA sweep (k_draw_mfma), the one-accumulator 16-bit sweep (k_draw_bf16) and the search of the product-sliced form (k_draw_search).
// (see rg_common.hpp for the shared types and helpers, DESIGN.md for the data layout and the rooflines)

#include "rg_common.hpp"

namespace rgk {

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

// second kernel of the sliced mode: the search over the sums all slices of a user tile left
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

// kernel selection by (KH, N1, N2, N3)
search_kernel_t search_kernel_for(const DevSim& d) {
    switch (d.KH) {
        case 4: return k_draw_search<4>;
        case 10: return k_draw_search<10>;
        case 16: return k_draw_search<16>;
        case 32: return k_draw_search<32>;
        default: return k_draw_search<64>;
    }
}

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

}  // namespace rgk
