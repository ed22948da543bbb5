// rg_walk.hip — librecogym_hip.so, unit 7 of 7: the user-major walk of sigma_omega = 0 runs: k_walk, k_walk2, k_walk_solo, k_cache_prefix, k_exact_prefix.
// (see rg_common.hpp for the shared types and helpers, DESIGN.md for the data layout and the rooflines)

#include "rg_common.hpp"

namespace rgk {

template <int KH, int OCC, bool DENSE>
__global__ void __launch_bounds__(kBlock, OCC) k_walk(DevSim d_arg, uint32_t n_work, int round, uint32_t chunk_rows,
                                                       uint32_t in_base, uint32_t out_base) {
    // The ~60 fields of DevSim this kernel uses do not fit the scalar registers next to its own state: kept live across
    // the loop they were spilled into VGPR lanes (v_writelane / v_readlane: ~10 % of the kernel's VALU instructions, the
    // unit that bounds it).  They are read from the kernel-argument segment instead — scalar loads, at the point of use:
    // the pointer is laundered once per iteration so that the loads are not hoisted out of the loop again.
    (void)d_arg;
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    const DevSim& d = *(const DevSim*)kargs;
    constexpr int K2 = 2 * KH;
    // (the draw's chunk recomputed as a float64 dot; -DRG_WALK_PRECISE_CHUNK=0: in fp32 with its own budget delta_c — a tie: rg_common.hpp)
    constexpr bool kPreciseChunk = RG_WALK_PRECISE_CHUNK && KH <= 10;
    constexpr double kLog2e64 = 1.4426950408889634074;
    constexpr int kEmpty = 3;
    // a user that stops still owes its phantom row (one more policy act, abstract.py:311-316): it takes it on its lane's
    // NEXT step, through the one policy_act call site of the loop (a second inlined copy of the policy cost ~15 % of
    // the kernel's instructions and was executed on half of the steps)
    constexpr int kPhantom = 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    // A lane holds kWalkUsers users and, each iteration, advances the first of them that is in the state the wave
    // processes (see below): with one user per lane ~45 % of the lanes had nothing to do in an iteration.
    constexpr int kIdle = 5;                                       // this iteration: none of the lane's users takes part
    // per wave: omega32 of the lanes' users [entry][K2][64] (k-major: conflict-free; a lane keeps a user to its end, so
    // omega is fetched once per USER; the recompute and the fp32 click decision read it) | mailbox [64] {idx, A, B}
    char* wbase = smem_raw + static_cast<size_t>(wave) * walk_wave_lds(KH);
    float* om32 = reinterpret_cast<float*>(wbase);
    double* mbox = reinterpret_cast<double*>(om32 + kWalkUsers * K2 * 64);   // [64][3]
    uint32_t* slots = reinterpret_cast<uint32_t*>(mbox + 64 * 3);  // [64] lanes of the searching users, by rank
    const uint32_t n_cc = d.PT / 64;
    uint32_t slotA[kWalkUsers], tA[kWalkUsers];
    int stA[kWalkUsers];
    bool pendA[kWalkUsers];                                        // round 2: the parked draw, to be picked in float64
#pragma unroll
    for (int e = 0; e < kWalkUsers; ++e) { slotA[e] = 0; tA[e] = 0; stA[e] = kEmpty; pendA[e] = false; }
    uint32_t res_next = 0, res_end = 0;                            // this wave's reservoir of queue tickets
    uint64_t row_next = 0, row_end = 0;                            // this wave's reserved raw-log rows
    uint32_t park_next = 0, park_end = 0;                          // this wave's reserved park_list entries
    bool exhausted = false;
    unsigned long long c_org = 0, c_ban = 0, c_clicks = 0, c_ph = 0, c_pick = 0, c_sweeps = 0;
    uint32_t c_maxt = 0, c_limit = 0;

    for (;;) {
        asm volatile("" : "+s"(kargs));
        const DevSim& d = *(const DevSim*)kargs;
        // ---- refill the entries whose user has stopped (or was parked) ----
#pragma unroll
        for (int e = 0; e < kWalkUsers; ++e) {
        unsigned long long dead = __ballot(stA[e] == kEmpty);
        if (dead && !exhausted && (static_cast<uint32_t>(__popcll(dead)) >= d.walk_refill || dead == ~0ull)) {
            for (int pass = 0; pass < 2 && dead; ++pass) {
                if (res_next == res_end) {
                    if (exhausted) break;
                    uint32_t base = 0;
                    if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_ticket, 64ull));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= n_work) { exhausted = true; break; }
                    res_next = base; res_end = min(base + 64u, n_work);
                }
                const uint32_t take = min(static_cast<uint32_t>(__popcll(dead)), res_end - res_next);
                const uint32_t r = prefix_in_mask(dead);
                const bool mine = ((dead >> lane) & 1ull) != 0 && r < take;   // (a lane that drew an unused entry in pass 1 is not in `dead`)
                if (mine) {
                    const uint32_t idx = res_next + r;
                    uint32_t s2 = idx;
                    if (round >= 2) s2 = d.park_list[in_base + idx];
                    if (s2 != 0xFFFFFFFFu) {
                        slotA[e] = s2;
                        stA[e] = RG_STATE_ORGANIC;                   // every user starts organic
                        tA[e] = 0u;
                        pendA[e] = false;
                        if (round >= 2) {
                            // a parked user sits at an organic draw to be picked in float64; a handed-over one anywhere
                            const uint32_t pt = d.park_t[s2];
                            tA[e] = pt & 0xFFFFFFu; stA[e] = static_cast<int>((pt >> 24) & 7u); pendA[e] = (pt >> 27) & 1u;
                            if (round == 2) { d.f64_valid[s2] = 1; c_sweeps += pendA[e] ? 1 : 0; }   // the batch between the rounds took its sums (counted for the parked)
                        }
                        // omega32 = float(omega), as k_cache_finalize left it in the user's cache row (floats 44 .. 44 + K2)
                        const float4* rp = reinterpret_cast<const float4*>(d.cache_row + static_cast<size_t>(s2) * d.cache_row_f);
                        float* o = om32 + e * (K2 * 64) + lane;
#pragma unroll
                        for (int k4 = 0; k4 < K2 / 4; ++k4) {
                            const float4 x = rp[11 + k4];
                            o[(4 * k4) * 64] = x.x; o[(4 * k4 + 1) * 64] = x.y; o[(4 * k4 + 2) * 64] = x.z; o[(4 * k4 + 3) * 64] = x.w;
                        }
#pragma unroll
                        for (int k = (K2 / 4) * 4; k < K2; ++k) o[k * 64] = reinterpret_cast<const float*>(rp)[44 + k];
                    }
                }
                res_next += take;
                dead = __ballot(stA[e] == kEmpty && !mine);          // lanes that drew an unused entry wait for the next refill
            }
        }
        }
        // (sorting the block's users by state through LDS so that waves are all-organic or all-bandit was measured
        // SLOWER, 352 vs 301 ms on C3: its two barriers per step serialise the block on its organic wave's latency chain)
        bool any_user = false, has_org = false, has_ban = false;
#pragma unroll
        for (int e = 0; e < kWalkUsers; ++e) {
            any_user = any_user || stA[e] != kEmpty;
            has_org = has_org || stA[e] == RG_STATE_ORGANIC;
            has_ban = has_ban || stA[e] == RG_STATE_BANDIT || stA[e] == kPhantom;
        }
        const unsigned long long live = __ballot(any_user);
        if (!live) { if (exhausted) break; else continue; }
        // ---- hand-over: once the queue is empty a wave would drain for several user lifetimes with ever fewer live
        // lanes (an iteration costs the same whatever their number).  With few left it passes its users on — they are
        // appended to the list the next round reads, with their time, state and pending-pick flag — and ends; the
        // last round walks everyone to the end. ----
        if (exhausted && round < 3 && d.walk_handover && static_cast<uint32_t>(__popcll(live)) <= d.walk_handover) {
#pragma unroll
            for (int e = 0; e < kWalkUsers; ++e) {
                const bool give = stA[e] != kEmpty;
                const unsigned long long gm = __ballot(give);
                if (!gm) continue;
                const uint32_t np = static_cast<uint32_t>(__popcll(gm));
                if (park_next + np > park_end) {
                    for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
                    uint32_t base = 0;
                    if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_park, 64ull));
                    base = __builtin_amdgcn_readfirstlane(base);
                    park_next = base; park_end = base + 64;
                }
                if (give) {
                    d.park_list[out_base + park_next + prefix_in_mask(gm)] = slotA[e];
                    d.park_t[slotA[e]] = tA[e] | (static_cast<uint32_t>(stA[e]) << 24) | (pendA[e] ? 1u << 27 : 0u);
                    // round 1: the float64 batch takes the sums of every listed user with the reference found here — the
                    // user's common reference, as a parked draw would have left it (float 32 of its cache row)
                    if (round == 1) d.exact_ref[slotA[e]] = d.cache_row[static_cast<size_t>(slotA[e]) * d.cache_row_f + 32];
                    stA[e] = kEmpty;
                }
                park_next += np;
            }
            break;
        }
        {
        // ---- ONE kind of event per iteration: the organic draw and the bandit event are different code, and a wave whose
        // lanes are in both states executes both for every step at ~25 active lanes each.  The users are independent and
        // every draw is addressed by (user, t), so the lanes in the minority state simply wait an iteration: the wave runs
        // the path more of its lanes are ready for (organic weighted by walk_bias / 4: its path is the longer one). ----
        bool run_org = true, run_ban = true;
        if (d.walk_bias) {
            const uint32_t n_ro = static_cast<uint32_t>(__popcll(__ballot(has_org)));
            const uint32_t n_rb = static_cast<uint32_t>(__popcll(__ballot(has_ban)));
            run_org = n_ro != 0 && n_ro * d.walk_bias >= n_rb * 4u;
            run_ban = !run_org;
        }
        // this iteration's user of the lane: its first one in a state that is processed
        int sel = -1;
#pragma unroll
        for (int e = kWalkUsers - 1; e >= 0; --e)
            if ((run_org && stA[e] == RG_STATE_ORGANIC) || (run_ban && (stA[e] == RG_STATE_BANDIT || stA[e] == kPhantom))) sel = e;
        uint32_t slot = slotA[0], t = tA[0];
        int st = sel == 0 ? stA[0] : kIdle;
        bool pending = pendA[0];
#pragma unroll
        for (int e = 1; e < kWalkUsers; ++e)
            if (sel == e) { slot = slotA[e]; t = tA[e]; st = stA[e]; pending = pendA[e]; }
        const uint32_t user = static_cast<uint32_t>(d.first_user + slot);
        const float* om_sel = om32 + max(sel, 0) * (K2 * 64);     // [K2][64]
        // ---- one raw-log row per lane that emits an event (not for the pending phantom rows: they have their own array) ----
        const unsigned long long rowm = __ballot((run_org && st == RG_STATE_ORGANIC) || (run_ban && st == RG_STATE_BANDIT));
        const uint32_t nlive = static_cast<uint32_t>(__popcll(rowm));
        if (row_next + nlive > row_end) {
            for (uint64_t r = row_next + lane; r < row_end; r += 64)
                if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&d.counters[kCntTailRows], static_cast<unsigned long long>(chunk_rows));
            base = readfirstlane_u64(base);
            row_next = base; row_end = base + chunk_rows;
        }
        const uint64_t my_row = row_next + prefix_in_mask(rowm);
        row_next += nlive;
        const bool alive = (run_org && st == RG_STATE_ORGANIC) || (run_ban && st == RG_STATE_BANDIT);
        const rg_u32x4 w = rg_draw(d.seed, user, t, 0, RG_DRAW_EVENT);
        const bool is_org = alive && st == RG_STATE_ORGANIC;
        bool parked = false;
        // =========================== organic product draw ===========================
        const unsigned long long org_mask = __ballot(is_org);
        if (org_mask && RG_WALK_ABL(20)) {          // timing experiment: no draw at all
            if (is_org) {
                if (d.log && my_row < d.log_cap) { rg_event e; e.u = user; e.t = t; e.code = (user + t) % d.P; e.ps = __builtin_nanf(""); d.log[my_row] = e; }
                c_org += 1;
            }
        } else
        if (org_mask) {
            const bool search = is_org && !pending;
            const size_t row = search ? slot : d.n_cap;
            // ---- the user's cache row (k_draw_cached phase 1) ----
            const float4* rp = reinterpret_cast<const float4*>(d.cache_row + row * d.cache_row_f);
            float W[kMaxSC];
#pragma unroll
            for (int i = 0; i < kMaxSC / 4; ++i) {
                const float4 x = rp[i];
                W[4 * i] = x.x; W[4 * i + 1] = x.y; W[4 * i + 2] = x.z; W[4 * i + 3] = x.w;
            }
            const float4 hdr = rp[8];
            const float4 of0 = rp[9], of1 = rp[10];
            const float Q = hdr.x;
            const double delta = static_cast<double>(hdr.y);
            double S = 0.0;
#pragma unroll
            for (uint32_t sc = 0; sc < kMaxSC; ++sc) S += static_cast<double>(W[sc]);
            const double u_org = d.u_override ? d.u_override[slot] : rg_uniform(w.w[0], w.w[1]);
            const double tau = u_org * S;
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
            uint32_t offw;
            {
                const uint32_t q = sc_star >> 2;
                const float4 o4 = q < 4 ? of0 : of1;
                const float ow = (q & 3) == 0 ? o4.x : (q & 3) == 1 ? o4.y : (q & 3) == 2 ? o4.z : o4.w;
                offw = (__builtin_bit_cast(uint32_t, ow) >> (8 * (sc_star & 3))) & 0xFFu;
            }
            if (offw >= 127u) found_sc = false;
            const float f_star = found_sc ? __builtin_amdgcn_exp2f(-static_cast<float>(offw)) : 1.0f;
            // ---- the chunk inside that super-chunk (phase 2) ----
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
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // ---- the 32 products of the chosen chunk (phase 3): eight searching users per pass, eight lanes per
            // user, four products per lane.  A pass is ONE latency chain (user parameters -> 21 coalesced
            // 16-byte loads -> 80 fma -> 4 exp -> 3-step prefix across the user's lanes -> compare); two users
            // per pass of 32-lane prefixes cost a chain per pair and made the walk 4x slower ----
            const int grp = lane >> 3, gl = lane & 7;
            unsigned long long todo = __ballot(search);
            if RG_WALK_ABL(16) { todo = 0; if (search) { mbox[lane * 3] = 0.0; mbox[lane * 3 + 1] = 0.0; mbox[lane * 3 + 2] = 1e300; } }
            while (todo) {
                int src = -1;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int bit = todo ? __builtin_ctzll(todo) : -1;
                    if (g == grp) src = bit;
                    if (todo) todo &= todo - 1;
                }
                const bool has = src >= 0;
                const int s2 = has ? src : 0;
                const uint32_t cs = static_cast<uint32_t>(__shfl(static_cast<int>(c_star), s2));
                const float Qs = __shfl(Q, s2);
                const double pbs = __shfl(pb, s2), taus = __shfl(tau, s2);
                const float4* gp = reinterpret_cast<const float4*>(d.gamma32t + (static_cast<size_t>(cs) * K2) * 32) + gl;
                float4 l = *(reinterpret_cast<const float4*>(d.mu32 + cs * 32) + gl);
                const float* o = om32 + __shfl(max(sel, 0), s2) * (K2 * 64) + s2;
#pragma unroll
                for (int kh = 0; kh < K2; kh += KH) {          // two halves: KH 16-byte loads in flight, then their fmas
                    float4 gk[KH];
#pragma unroll
                    for (int k = 0; k < KH; ++k) gk[k] = gp[(kh + k) * 8];
#pragma unroll
                    for (int k = 0; k < KH; ++k) {
                        const float wk = o[(kh + k) * 64];
                        l.x = fmaf(gk[k].x, wk, l.x); l.y = fmaf(gk[k].y, wk, l.y);
                        l.z = fmaf(gk[k].z, wk, l.z); l.w = fmaf(gk[k].w, wk, l.w);
                    }
                    asm volatile("" : "+v"(l.x), "+v"(l.y), "+v"(l.z), "+v"(l.w));   // keeps the second half's loads behind these
                }
                const float e0 = __builtin_amdgcn_exp2f(fmaf(l.x, kLog2e, -Qs)), e1 = __builtin_amdgcn_exp2f(fmaf(l.y, kLog2e, -Qs));
                const float e2 = __builtin_amdgcn_exp2f(fmaf(l.z, kLog2e, -Qs)), e3 = __builtin_amdgcn_exp2f(fmaf(l.w, kLog2e, -Qs));
                const float q0 = e0, q1 = q0 + e1, q2 = q1 + e2, q3 = q2 + e3;      // prefix inside the lane
                float inc = q3;                                                     // ... and across the user's 8 lanes
#pragma unroll
                for (int o2 = 1; o2 < 8; o2 <<= 1) {
                    const float y = __shfl_up(inc, o2, 8);
                    if (gl >= o2) inc += y;
                }
                float ex = __shfl_up(inc, 1, 8);                                    // prefix before this lane's products
                if (gl == 0) ex = 0.0f;
                const double pxb = pbs + static_cast<double>(ex);
                const double px0 = pbs + static_cast<double>(ex + q0), px1 = pbs + static_cast<double>(ex + q1);
                const double px2 = pbs + static_cast<double>(ex + q2), px3 = pbs + static_cast<double>(ex + q3);
                const int j0 = px0 > taus ? 0 : px1 > taus ? 1 : px2 > taus ? 2 : px3 > taus ? 3 : -1;
                const unsigned long long hits = __ballot(has && j0 >= 0);
                const uint32_t gmask = static_cast<uint32_t>(hits >> (8 * grp)) & 0xFFu;
                if (has) {
                    if (gmask) {
                        if (gl == __builtin_ctz(gmask)) {
                            mbox[src * 3] = static_cast<double>(4 * gl + j0);
                            mbox[src * 3 + 1] = j0 == 0 ? pxb : j0 == 1 ? px0 : j0 == 2 ? px1 : px2;
                            mbox[src * 3 + 2] = j0 == 0 ? px0 : j0 == 1 ? px1 : j0 == 2 ? px2 : px3;
                        }
                    } else if (gl == 0) { mbox[src * 3] = -1.0; mbox[src * 3 + 1] = pbs; mbox[src * 3 + 2] = pbs; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            uint32_t v = 0;
            bool ok = false;
            if (search) {
                const int idx = static_cast<int>(mbox[lane * 3]);
                const double Av = mbox[lane * 3 + 1], Bv = mbox[lane * 3 + 2];
                v = c_star * 32 + static_cast<uint32_t>(max(idx, 0));
                const CertLin ct = cert_correlated(S, pb, Av - pb, Bv - pb, delta);
                ok = found_c && idx >= 0 && v < d.P && ct.valid &&
                     (v == 0 || u_org * ct.den_lo > ct.num_lo) &&
                     (v == d.P - 1 || u_org * ct.den_hi < ct.num_hi);
            }
            // ---- uncertified: float64 pick from the user's stored sums, or park the user until they exist ----
            const bool need64 = is_org && !ok;
            const bool have64 = need64 && d.f64_valid[slot] != 0;
            parked = need64 && !have64;
            unsigned long long picks = __ballot(have64);
            while (picks) {
                const int L = __builtin_ctzll(picks);
                picks &= picks - 1;
                const uint32_t s_slot = static_cast<uint32_t>(__shfl(static_cast<int>(slot), L));
                const double s_u = __shfl(u_org, L);
                const double M = static_cast<double>(d.exact_ref[s_slot]) * 0.69314718055994530942;
                const uint32_t pv = exact_pick_wave(d, d.exact_sums + static_cast<size_t>(s_slot) * n_cc,
                                                    d.omega + static_cast<size_t>(s_slot) * d.OMS, M, s_u, 1u, lane);
                if (lane == L) { v = pv; c_pick += 1; }
                __builtin_amdgcn_wave_barrier();
            }
            if (parked) {
                d.park_t[slot] = t | (static_cast<uint32_t>(RG_STATE_ORGANIC) << 24) | (1u << 27);
                d.exact_ref[slot] = Q;
            }
            if (is_org && !parked) {
                if (d.log && my_row < d.log_cap) {
                    rg_event e;
                    e.u = user; e.t = t; e.code = v; e.ps = __builtin_nanf("");
                    d.log[my_row] = e;
                }
                if (d.lpv) d.lpv[slot] = v;
                if (d.hist_cap && !RG_WALK_ABL(17)) history_add(d, slot, v);
                c_org += 1;
                pending = false;
            }
        }
        // ---- park list entries for the users parked in this step ----
        const unsigned long long pmask = __ballot(parked);
        if (pmask) {
            const uint32_t np = static_cast<uint32_t>(__popcll(pmask));
            if (park_next + np > park_end) {
                for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
                uint32_t base = 0;
                if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_park, 64ull));
                base = __builtin_amdgcn_readfirstlane(base);
                park_next = base; park_end = base + 64;
            }
            if (parked) {
                d.park_list[out_base + park_next + prefix_in_mask(pmask)] = slot;
                if (d.log && my_row < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[my_row] = e; }
                st = kEmpty;
            }
            park_next += np;
        }
        // =========================== bandit event + transition (k_advance's arithmetic) ===========================
        const bool is_ban = run_ban && st == RG_STATE_BANDIT, is_ph = run_ban && st == kPhantom;
        double ps = 1.0;
        uint32_t a = 0;
        if (is_ban || is_ph) a = RG_WALK_ABL(18) ? (user + t) % d.P : policy_act<DENSE>(d, slot, user, t, &ps);
        if (is_ph) {       // final step_offline(done = True): the act above, reward 0 (abstract.py:223-233,311-316); t is already the row's time
            rg_event e;
            e.u = user; e.t = t; e.code = RG_EV_BANDIT | RG_EV_PHANTOM | a;
            e.ps = static_cast<float>(ps);
            d.phantom[slot] = e;
            d.phantom_ps[slot] = ps;
            d.has_phantom[slot] = 1;
            c_ph += 1;
            st = kEmpty;
        }
        if (alive && !parked) {
            const double u_trans = rg_uniform(w.w[2], w.w[3]);
            bool click = false;
            // The click is a Bernoulli draw against ff(beta[a].omega + mu_b[a]) (three nested sigmoids: three float64
            // exps and four divisions).  Its outcome is decided in fp32 wherever the fp32 value of 1 - ff is further from
            // the uniform than the fp32 error bound (fp32 dot: (K + 2) 2^-24 sum|beta_k omega_k|, damped by the chain's
            // slope <= 0.05; three v_exp / v_rcp at ~1e-6); the float64 evaluation below is for the lanes inside that
            // band (~4e-5 of the acts) and for runs that export the click probability.
            bool click_known = false;
            if (is_ban && !d.aux_pclick && rg_uniform(w.w[0], w.w[1]) < kNoClickBelow) click_known = true;     // (click = false)
            else
            if (is_ban && !d.aux_pclick && !RG_WALK_ABL(21)) {
                const float* om_l = om_sel + lane;
                const int dec = click_decide32<((K2 + 3) / 4) * 4>(d.beta32 + static_cast<size_t>(a) * d.KB4, [&](int k) { return om_l[k * 64]; },
                                                                   d.K, d.KB4, static_cast<float>(d.mu_b[a]), rg_uniform(w.w[0], w.w[1]));
                if (dec >= 0) { click = dec != 0; click_known = true; }
            }
            if (is_ban && click_known) {
                c_clicks += click;
                c_ban += 1;
                if (d.log && my_row < d.log_cap) {
                    rg_event e;
                    e.u = user; e.t = t;
                    e.code = RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a;
                    e.ps = static_cast<float>(ps);
                    d.log[my_row] = e;
                    if (d.aux_ps) d.aux_ps[my_row] = ps;
                }
            }
            if (is_ban && !click_known) {
                const double* b = d.beta + static_cast<size_t>(a) * d.K;
                const double* om = d.omega + static_cast<size_t>(slot) * d.OMS;
                double x = 0.0;
                if RG_WALK_ABL(19) {}
                else if (!(d.K & 1)) {
                    // rows of K even are 16-byte aligned: half as many (scattered) load requests as 8-byte loads
                    for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {
                        double2 wv[4], bv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t k = min(k0 + 2 * i, d.K - 2);
                            wv[i] = *reinterpret_cast<const double2*>(om + k);
                            bv[i] = *reinterpret_cast<const double2*>(b + k);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (k0 + 2 * i < d.K) { x += bv[i].x * wv[i].x; x += bv[i].y * wv[i].y; }
                    }
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
                const double ctr = RG_WALK_ABL(19) ? 0.01 : ff64(x + d.mu_b[a]);
                const double p0 = 1.0 - ctr;
                click = (p0 / (p0 + ctr)) <= rg_uniform(w.w[0], w.w[1]);
                c_clicks += click;
                c_ban += 1;
                if (d.log && my_row < d.log_cap) {
                    rg_event e;
                    e.u = user; e.t = t;
                    e.code = RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a;
                    e.ps = static_cast<float>(ps);
                    d.log[my_row] = e;
                    if (d.aux_ps) d.aux_ps[my_row] = ps;
                    if (d.aux_pclick) d.aux_pclick[my_row] = ctr;
                }
            }
            const double c0 = is_org ? d.cdf_o0 : d.cdf_b0, c1 = is_org ? d.cdf_o1 : d.cdf_b1;
            int ns = (c0 <= u_trans) + (c1 <= u_trans);
            if (click) ns = RG_STATE_ORGANIC;                  // abstract.py:180-181 (sigma_omega == 0: no drift to apply)
            const bool organic_only = (d.first_user + slot) < d.organic_only_below;
            if (organic_only && ns != RG_STATE_ORGANIC) {
                ns = RG_STATE_STOP;
                d.n_events[slot] = t + 1;
            } else if (ns == RG_STATE_STOP) {
                d.n_events[slot] = t + 1;
                ns = kPhantom;                                 // the phantom row's act: this lane's next step
            } else if (t + 2 >= kMaxSteps) {
                ns = RG_STATE_STOP;
                d.n_events[slot] = t + 1;
                c_limit += 1;
            }
            if (ns == RG_STATE_STOP || ns == kPhantom) c_maxt = max(c_maxt, t + 1);
            if (ns == RG_STATE_STOP) st = kEmpty;
            else { st = ns; t += 1; }
        }
#pragma unroll
        for (int e = 0; e < kWalkUsers; ++e)
            if (sel == e) { stA[e] = st; tA[e] = t; pendA[e] = pending; }
        }   // if (live)
    }
    // ---- leftovers of the reserved chunks, counters ----
    for (uint64_t r = row_next + lane; r < row_end; r += 64)
        if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
    for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
    for (int o = 32; o > 0; o >>= 1) {
        c_org += __shfl_xor(c_org, o); c_ban += __shfl_xor(c_ban, o); c_clicks += __shfl_xor(c_clicks, o);
        c_ph += __shfl_xor(c_ph, o); c_pick += __shfl_xor(c_pick, o); c_sweeps += __shfl_xor(c_sweeps, o);
        c_maxt = max(c_maxt, static_cast<uint32_t>(__shfl_xor(static_cast<int>(c_maxt), o)));
        c_limit += static_cast<uint32_t>(__shfl_xor(static_cast<int>(c_limit), o));
    }
    if (lane == 0) {
        if (c_org) atomicAdd(&d.counters[kCntTailOrganic], c_org);
        if (c_ban) atomicAdd(&d.counters[kCntTailBandit], c_ban);
        if (c_clicks) atomicAdd(&d.counters[RG_CNT_CLICKS], c_clicks);
        if (c_ph) atomicAdd(&d.counters[RG_CNT_PHANTOM], c_ph);
        if (c_pick) atomicAdd(&d.counters[RG_CNT_EXACT_DRAWS], c_pick);
        if (c_sweeps) atomicAdd(&d.counters[RG_CNT_EXACT_SWEEPS], c_sweeps);
        if (c_maxt) atomicMax(&d.counters[kCntTailMaxT], static_cast<unsigned long long>(c_maxt));
        if (c_limit) atomicAdd(&d.counters[kCntTailLimit], static_cast<unsigned long long>(c_limit));
    }
}


// The prefix form.  Eight lanes per user, 32 chunks per pass (one 128-byte line of the user's chunk sums): scaled to the
// user's common reference Q (exact powers of two), summed in float64 in chunk order, stored back in place as fp32
// prefixes; the prefix at the end of every super-chunk also goes to the user's scp row, the total into its hot row.
// fused = 1: k_draw_bf16p already stored prefixes (sweep_only = 2), each on the reference of its super-chunk: what is left is
// the hot row's header and, for the users whose reference moved during the sweep (cache_resc != 0: rare), the exact rescaling
// (powers of two) of their entries to the common reference.
__global__ void __launch_bounds__(kBlock) k_cache_prefix(DevSim d, int fused) {
    const int lane = lane_id(), grp = lane >> 3, gl = lane & 7;
    const uint32_t n_groups = (d.grp_n + 7) / 8;
    const uint32_t waves = gridDim.x * (kBlock / 64);
    for (uint32_t ug = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); ug < n_groups; ug += waves) {
        const uint32_t i = d.grp_lo + ug * 8 + grp;
        // (fin_in_sweep: only the users whose reference moved during the sweep — the others' prefixes need no rescaling and the
        // sweep left their hot rows)
        const bool act = i < d.grp_lo + d.grp_n && !(d.fin_in_sweep && d.cache_resc[i < d.grp_lo + d.grp_n ? i : d.n_cap] == 0);
        const size_t row = act ? i : d.n_cap;
        const float4* r4 = reinterpret_cast<const float4*>(d.cache_row + row * d.cache_row_f);
        const float4 hdr = r4[8], of0 = r4[9], of1 = r4[10];
        float* cp = d.cache_chunk + row * d.n_chunks;
        float* scp = d.walk_scp + row * kMaxSC;
        double run = 0.0;
        const bool moved = fused && act && d.cache_resc[row] != 0;
        if (fused) {
            if (moved) {
                for (uint32_t c0 = 0; c0 < d.n_chunks; c0 += 32) {
                    const uint32_t c = c0 + 4 * gl;
                    if (c >= d.n_chunks) continue;
                    const uint32_t sc = min(c / d.sc_chunks, kMaxSC - 1u);
                    const uint32_t q = sc >> 2;
                    const float4 o4 = q < 4 ? of0 : of1;
                    const float ow = (q & 3) == 0 ? o4.x : (q & 3) == 1 ? o4.y : (q & 3) == 2 ? o4.z : o4.w;
                    const uint32_t off = (__builtin_bit_cast(uint32_t, ow) >> (8 * (sc & 3))) & 0xFFu;
                    const float f = off >= 127u ? 0.0f : __builtin_amdgcn_exp2f(-static_cast<float>(off));
                    float4 w = *reinterpret_cast<const float4*>(cp + c);
                    w.x *= f; w.y *= f; w.z *= f; w.w *= f;
                    *reinterpret_cast<float4*>(cp + c) = w;
                    if ((c + 4) % d.sc_chunks == 0 || c + 4 == d.n_chunks) scp[sc] = w.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            run = act ? static_cast<double>(scp[d.n_sc - 1]) : 0.0;
        } else
        for (uint32_t c0 = 0; c0 < d.n_chunks; c0 += 32) {
            const uint32_t c = c0 + 4 * gl;
            const bool in = c < d.n_chunks;
            const float4 w = in ? *reinterpret_cast<const float4*>(cp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const uint32_t sc = min(c / d.sc_chunks, kMaxSC - 1u);       // (sc_chunks % 4 == 0: one super-chunk per float4)
            const uint32_t q = sc >> 2;
            const float4 o4 = q < 4 ? of0 : of1;
            const float ow = (q & 3) == 0 ? o4.x : (q & 3) == 1 ? o4.y : (q & 3) == 2 ? o4.z : o4.w;
            const uint32_t off = (__builtin_bit_cast(uint32_t, ow) >> (8 * (sc & 3))) & 0xFFu;
            const float f = off >= 127u ? 0.0f : __builtin_amdgcn_exp2f(-static_cast<float>(off));
            const double p0 = static_cast<double>(w.x * f), p1 = p0 + static_cast<double>(w.y * f);
            const double p2 = p1 + static_cast<double>(w.z * f), p3 = p2 + static_cast<double>(w.w * f);
            double inc = p3;
#pragma unroll
            for (int o2 = 1; o2 < 8; o2 <<= 1) {
                const double y = __shfl_up(inc, o2, 8);
                if (gl >= o2) inc += y;
            }
            const double base = run + (inc - p3);
            const float4 out = make_float4(static_cast<float>(base + p0), static_cast<float>(base + p1),
                                           static_cast<float>(base + p2), static_cast<float>(base + p3));
            if (in && act) {
                *reinterpret_cast<float4*>(cp + c) = out;
                if ((c + 4) % d.sc_chunks == 0 || c + 4 == d.n_chunks) scp[sc] = out.w;
            }
            run += __shfl(inc, (grp << 3) | 7);
        }
        if (act) {
            for (uint32_t sc = d.n_sc + gl; sc < kMaxSC; sc += 8) scp[sc] = INFINITY;     // never counted
            if (gl == 0) {
                // S~ as the search sees it (the last prefix), the certificate's delta + 2^-21 for the roundings of the stored
                // prefixes (<= 5 of 2^-24 each, relative to the prefix: the same kind of error the budget is made of), Q, an empty memo
                float4* hot = reinterpret_cast<float4*>(d.walk_hot + row * 32);
                hot[0] = make_float4(static_cast<float>(run), hdr.y * 1.000001f + 4.8e-7f, hdr.x, __builtin_bit_cast(float, 0u));
                // (these prefixes went through several fp32 / rescaling roundings: rho = 2^-20.  Behind k_sweep_xh the records of a
                // user whose reference moved are fp32 DIFFERENCES of staged prefixes — each off by ~2 x 2^-24 of the prefix, up to
                // kMaxSC = 32 of them summed again here: 2^-19 at worst, so such rows get 2^-18.  ADVICE round 5)
                reinterpret_cast<float*>(hot)[31] = (d.XNH && d.cache_resc[row]) ? 4.0f * kRhoLoose : kRhoLoose;
            }
        }
    }
}

// next float above / below (finite x; the roundings of the memo's interval bounds and of the uniform go INWARDS)
__device__ __forceinline__ float f32_up(float x) {
    const uint32_t b = __builtin_bit_cast(uint32_t, x);
    return x == 0.0f ? __builtin_bit_cast(float, 1u) : __builtin_bit_cast(float, x > 0.0f ? b + 1u : b - 1u);
}
__device__ __forceinline__ float f32_down(float x) {
    const uint32_t b = __builtin_bit_cast(uint32_t, x);
    return x == 0.0f ? __builtin_bit_cast(float, 0x80000001u) : __builtin_bit_cast(float, x > 0.0f ? b - 1u : b + 1u);
}

// The float64 pick on sums stored as PREFIXES (k_exact_prefix): the 64-product chunk by counting the prefixes <= u total
// (three ballots instead of three wave scans), then its products walked in product order as exact_pick_wave does.
__device__ __forceinline__ uint32_t exact_pick_pfx(const DevSim& d, const double* pfx, const double* om, double M, double u, int lane) {
    const uint32_t n_cc = d.PT / 64;
    const double total = pfx[n_cc - 1];
    const double target = u * total;
    uint32_t cnt = 0;
    for (uint32_t c0 = 0; c0 < n_cc; c0 += 64) {
        const uint32_t c = c0 + lane;
        cnt += static_cast<uint32_t>(__popcll(__ballot(c < n_cc && pfx[c] <= target)));
    }
    const uint32_t ccstar = min(cnt, n_cc - 1u);            // (u * total rounded up to total: the last chunk)
    double acc = ccstar ? pfx[ccstar - 1] : 0.0;
    uint32_t v = min(ccstar * 64 + 63, d.P - 1);           // if rounding leaves no hit: the chunk's last product
    const uint32_t p = ccstar * 64 + lane;
    const double* g = d.gammaT + p;                        // PT columns: always in range
    double lg = 0.0;
    for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {             // same association as the oracle (k ascending)
        double gv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = g[static_cast<size_t>(min(k0 + j, d.K - 1)) * d.PT];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k0 + j < d.K) lg += gv[j] * om[k0 + j];
    }
    lg = p < d.P ? lg + d.mu_o[p] : -INFINITY;
    const double inc = wave_scan(exp64(lg - M), lane);
    const unsigned long long hit = __ballot(p < d.P && acc + inc > target);
    if (hit) v = ccstar * 64 + static_cast<uint32_t>(__builtin_ctzll(hit));
    return v;
}

// exact_sums rows of the listed users (the float64 batch between rounds 1 and 2 just took them) -> inclusive prefixes, in
// place, in exact_pick_wave's association (a wave scan per block of 64 sums, the blocks in order).  A wave per user.
__global__ void __launch_bounds__(kBlock) k_exact_prefix(DevSim d, uint32_t n_list) {
    const int lane = lane_id();
    const uint32_t n_cc = d.PT / 64;
    const uint32_t waves = gridDim.x * (kBlock / 64);
    if (d.q_count) n_list = static_cast<uint32_t>(*d.q_count);
    for (uint32_t w = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w < n_list; w += waves) {
        const uint32_t slot = d.park_list[d.list_in + w];
        if (slot == 0xFFFFFFFFu) continue;
        double* row = d.exact_sums + static_cast<size_t>(slot) * n_cc;
        double run = 0.0;
        for (uint32_t c0 = 0; c0 < n_cc; c0 += 64) {
            const uint32_t c = c0 + lane;
            const double incl = wave_scan(c < n_cc ? row[c] : 0.0, lane);
            if (c < n_cc) row[c] = run + incl;
            run += __shfl(incl, 63);
        }
    }
}

// (the float64 pick as a noinline CALL freed ~40 registers of the walk's loop, but at four blocks per CU — 128 VGPRs, the loop
// spilling around the call — the kernel no longer reproduced the oracle: measured, dropped; inlined at three blocks per CU
// the loop holds everything in 168 registers)
__device__ __forceinline__ uint32_t exact_pick_call(const DevSim& d, const double* sums, const double* om, double M,
                                                   double u, int lane) {
    return exact_pick_wave(d, sums, om, M, u, 1u, lane);
}

// k_walk2's COMPACT history line (HIST == 2: products < 65 535, hist_cap <= 32 768): the same 128 bytes of LDS per lane hold
// 32 words instead of 16 64-bit entries — word 0 the header (views << 15 | distinct), words 1 .. 31 the 31 smallest products as
// (PREFIX << 16 | product): the running view count up to and including the product in the high half (a user has < 65 536
// events), so the words ascend with the index, an unused word is 0xFFFFFFFF, and the policy's act — first product whose
// cumulative count exceeds u x views — is a COUNT of words below a key: two LDS round trips (the last word of every 8-word
// segment, then the segment) and ~40 vector instructions instead of a 10-instruction step per entry, and all but ~1 % of C3's
// events find their whole history in the line (15 products in 64-bit entries: 11.6 % beyond).  Word w of the line is half
// (w & 1) of the 64-bit LDS entry hl[(w >> 1) * 64]; the user's ROW keeps the (product, count) form every other kernel reads.
constexpr uint32_t kHcLine = 32;     // words of the compact line (header + 31 products)

// Three blocks per CU (168 VGPRs, no spills).  Four (128 VGPRs) were measured in two forms — omega32 re-read from the cache
// row instead of held in registers, and the Gamma rows of the chunk pass in two batches — and did not pay: the extra loads and
// spills cost what the fourth wave brought (C3 walk 130.9 vs 133.8 ms, C2 13.2 vs 12.1 ms: profiles/r3/ab_walk_call3.jsonl).
// The DevSim fields are read from the kernel-argument segment where they are used (as in k_walk); pinning the 25 or 38 of
// the main path in registers instead (104 -> 70 / 60 scalar loads in the code, 200 / 259 scalar registers in VGPR lanes)
// measured the same to 0.3 % (profiles/r3/ab_call11_pinned_fields_shard_sizes.jsonl): the waits are not the scalar loads'.
template <int KH, int HIST>
__global__ void __launch_bounds__(kBlock, (KH <= 10 ? 3 : 2)) k_walk2(DevSim d_arg, uint32_t n_work, int round, uint32_t chunk_rows,
                                                        uint32_t in_base, uint32_t out_base) {
    (void)d_arg;       // read from the kernel-argument segment at the point of use (see k_walk)
    const __attribute__((address_space(4))) char* kargs =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int K2 = 2 * KH;
    // (the draw's chunk recomputed as a float64 dot; -DRG_WALK_PRECISE_CHUNK=0: in fp32 with its own budget delta_c — a tie: rg_common.hpp)
    constexpr bool kPreciseChunk = RG_WALK_PRECISE_CHUNK && KH <= 10;
    constexpr double kLog2e64 = 1.4426950408889634074;
    constexpr int KC = ((K2 + 3) / 4) * 4;
    constexpr int kEmpty = 3, kPhantom = 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    char* wbase = smem_raw + static_cast<size_t>(wave) * walk2_wave_lds(HIST);
    hent_t* hl = reinterpret_cast<hent_t*>(wbase) + lane;               // HIST: [16][64] entry-major: hl[i * 64]
    float* mboxf = reinterpret_cast<float*>(wbase + (HIST ? 16 * 64 * 8 : 0));   // [64][3]: the search's result per lane
    uint32_t slot = 0, t = 0;
    int st = kEmpty;
    bool hdirty = false;                                               // the history line in LDS is newer than the user's row
    bool pend = false;                                                 // rounds >= 2: the parked draw, to be picked in float64
    float om[KC];                                                      // omega32 of the lane's user
#pragma unroll
    for (int k = 0; k < KC; ++k) om[k] = 0.0f;
    uint32_t res_next = 0, res_end = 0;
    uint64_t row_next = 0, row_end = 0;
    uint32_t park_next = 0, park_end = 0;
    bool exhausted = false;
    // wave-uniform tallies (scalar registers): events are counted by ballots
    uint32_t c_org = 0, c_ban = 0, c_clicks = 0, c_ph = 0, c_pick = 0, c_sweeps = 0, c_maxt = 0, c_limit = 0, c_hit = 0, c_anch = 0;
    if (const unsigned long long* qc = ((const DevSim*)kargs)->q_count) n_work = static_cast<uint32_t>(*qc);   // (pipeline: the list's length is on the device)
#ifdef RG_WALK_TIMING
    unsigned long long wk_it[4] = {0, 0, 0, 0}, wk_ln[4] = {0, 0, 0, 0}, wk_cy[4] = {0, 0, 0, 0}, wk_hv = 0, wk_hd = 0, wk_t0 = __builtin_amdgcn_s_memtime();
    int wk_kind = -1;
#endif

    for (;;) {
        asm volatile("" : "+s"(kargs));
        const DevSim& d = *(const DevSim*)kargs;
        const uint32_t n_cc = d.PT / 64;
#ifdef RG_WALK_TIMING
        {
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();
            if (wk_kind >= 0) wk_cy[wk_kind] += now_ - wk_t0;
            wk_t0 = now_; wk_kind = -1;
        }
#endif
        // the view history is written back when the lane lets go of the user (stop, park, hand-over) or needs the row
        auto flush_hist = [&](bool c) {
            if (HIST == 2 && c) {
                // (product, count) entries from the prefixes; the pairs that hold entries <= nd (what lies behind them in the row
                // is don't-care)
                ulonglong2* hw = reinterpret_cast<ulonglong2*>(hist_row(d, slot));
                const uint32_t h0 = static_cast<uint32_t>(hl[0]);
                const uint32_t nd = h0 & 0x7FFFu;
                uint32_t prev = 0u;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const hent_t x = hl[i * 64];
                    const uint32_t w0 = static_cast<uint32_t>(x), w1 = static_cast<uint32_t>(x >> 32);
                    hent_t e0, e1;
                    if (i == 0) e0 = (static_cast<hent_t>(h0 >> 15) << 32) | nd;
                    else { e0 = (static_cast<hent_t>(w0 & 0xFFFFu) << 32) | ((w0 >> 16) - prev); prev = w0 >> 16; }
                    e1 = (static_cast<hent_t>(w1 & 0xFFFFu) << 32) | ((w1 >> 16) - prev); prev = w1 >> 16;
                    if (static_cast<uint32_t>(2 * i) <= nd) hw[i] = make_ulonglong2(e0, e1);
                }
            } else
            if (HIST && c) {
                ulonglong2* hw = reinterpret_cast<ulonglong2*>(hist_row(d, slot));
#pragma unroll
                for (int i = 0; i < 8; ++i) hw[i] = make_ulonglong2(hl[(2 * i) * 64], hl[(2 * i + 1) * 64]);
            }
        };
        // the compact line from the user's row (its first 32 entries): running prefixes of the counts, unused words all ones
        auto load_compact = [&](uint32_t s_row) {
            const ulonglong2* hr2 = reinterpret_cast<const ulonglong2*>(hist_row(d, s_row));
            uint32_t run = 0u, nd = 0u;
#pragma unroll
            for (int b = 0; b < 16; b += 8) {           // (two batches of eight 16-byte loads: 32 registers in flight, not 64)
                ulonglong2 x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = hr2[b + i];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uint32_t w0, w1;
                    const bool in0 = static_cast<uint32_t>(2 * (b + i)) <= nd || b + i == 0;
                    if (b + i == 0) { nd = h_cnt(x[0].x); w0 = (h_prod(x[0].x) << 15) | nd; }
                    else { run += in0 ? h_cnt(x[i].x) : 0u; w0 = in0 ? ((run << 16) | h_prod(x[i].x)) : 0xFFFFFFFFu; }
                    const bool in1 = static_cast<uint32_t>(2 * (b + i) + 1) <= nd;
                    run += in1 ? h_cnt(x[i].y) : 0u;
                    w1 = in1 ? ((run << 16) | h_prod(x[i].y)) : 0xFFFFFFFFu;
                    hl[(b + i) * 64] = static_cast<hent_t>(w0) | (static_cast<hent_t>(w1) << 32);
                }
                asm volatile("" ::: "memory");
            }
        };
        // ---- refill the lanes whose user has stopped (or was parked) ----
        {
            unsigned long long dead = __ballot(st == kEmpty);
            if (dead && !exhausted && (static_cast<uint32_t>(__popcll(dead)) >= d.walk_refill || dead == ~0ull)) {
                // the lines of the users these lanes let go of (stop, phantom row, park), all at once: a write-back per stop
                // ran in ~40 % of the iterations for one or two lanes
                flush_hist(st == kEmpty && hdirty);
                hdirty = hdirty && st != kEmpty;
                for (int pass = 0; pass < 2 && dead; ++pass) {
                    if (res_next == res_end) {
                        if (exhausted) break;
                        uint32_t base = 0;
                        if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_ticket, 64ull));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base >= n_work) { exhausted = true; break; }
                        res_next = base; res_end = min(base + 64u, n_work);
                    }
                    const uint32_t take = min(static_cast<uint32_t>(__popcll(dead)), res_end - res_next);
                    const uint32_t r = prefix_in_mask(dead);
                    const bool mine = ((dead >> lane) & 1ull) != 0 && r < take;
                    if (mine) {
                        const uint32_t idx = res_next + r;
                        uint32_t s2 = d.grp_lo + idx;
                        if (round >= 2) s2 = d.park_list[in_base + idx];
                        if (s2 != 0xFFFFFFFFu) {
                            slot = s2; st = RG_STATE_ORGANIC; t = 0u; pend = false; hdirty = false;
                            if (round >= 2) {
                                const uint32_t pt = d.park_t[s2];
                                t = pt & 0xFFFFFFu; st = static_cast<int>((pt >> 24) & 7u); pend = (pt >> 27) & 1u;
                                if (round == 2) d.f64_valid[s2] = 1;         // the batch between the rounds took its sums
                                if (pend) st = kWSlow;                       // its draw goes straight to the float64 pick
                            }
                            {   // omega32 = float(omega), as k_cache_finalize left it in the user's cache row (floats 44 ..)
                                const float4* rp = reinterpret_cast<const float4*>(d.cache_row + static_cast<size_t>(s2) * d.cache_row_f);
#pragma unroll
                                for (int k4 = 0; k4 < K2 / 4; ++k4) {
                                    const float4 x = rp[11 + k4];
                                    om[4 * k4] = x.x; om[4 * k4 + 1] = x.y; om[4 * k4 + 2] = x.z; om[4 * k4 + 3] = x.w;
                                }
#pragma unroll
                                for (int k = (K2 / 4) * 4; k < K2; ++k) om[k] = reinterpret_cast<const float*>(rp)[44 + k];
                            }
                            if (HIST == 2) load_compact(s2);
                            else
                            if (HIST) {
                                const ulonglong2* hr2 = reinterpret_cast<const ulonglong2*>(hist_row(d, s2));
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    const ulonglong2 x = hr2[i];
                                    hl[(2 * i) * 64] = x.x; hl[(2 * i + 1) * 64] = x.y;
                                }
                            }
                        }
                    }
                    // (the float64 batch between the rounds swept for every parked user: counted where round 2 takes them)
                    if (round == 2) c_sweeps += static_cast<uint32_t>(__popcll(__ballot(mine && st != kEmpty && pend)));
                    res_next += take;
                    dead = __ballot(st == kEmpty && !mine);
                }
            }
        }
        const unsigned long long live = __ballot(st != kEmpty);
        if (!live && !exhausted) continue;
        // ---- hand-over (see k_walk); the wave's end: the lines still waiting for their write-back (one copy of that code) ----
        if (!live || (exhausted && round < 3 && d.walk_handover && static_cast<uint32_t>(__popcll(live)) <= d.walk_handover)) {
            const bool give = st != kEmpty;
            const unsigned long long gm = __ballot(give);
            const uint32_t np = static_cast<uint32_t>(__popcll(gm));
            if (park_next + np > park_end) {
                for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
                uint32_t base = 0;
                if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_park, 64ull));
                base = __builtin_amdgcn_readfirstlane(base);
                park_next = base; park_end = base + 64;
            }
            flush_hist(hdirty);                  // (the users handed over and the lines of lanes that are already empty)
            if (give) {
                d.park_list[out_base + park_next + prefix_in_mask(gm)] = slot;
                const int st_out = st == kWSlow ? RG_STATE_ORGANIC : st == kWClick ? RG_STATE_BANDIT : st;     // (they restart at the memo check / the act)
                d.park_t[slot] = t | (static_cast<uint32_t>(st_out) << 24) | (pend ? 1u << 27 : 0u);
                if (round == 1) d.exact_ref[slot] = d.cache_row[static_cast<size_t>(slot) * d.cache_row_f + 32];
                st = kEmpty;
            }
            park_next += np;
            break;
        }
        // ---- ONE kind of event per iteration: memo-answered organic draws, searching organic draws, bandit events ----
        const uint32_t n_o = static_cast<uint32_t>(__popcll(__ballot(st == RG_STATE_ORGANIC)));
        const uint32_t n_s = static_cast<uint32_t>(__popcll(__ballot(st == kWSlow)));
        const uint32_t n_b = static_cast<uint32_t>(__popcll(__ballot(st == RG_STATE_BANDIT || st == kPhantom)));
        const uint32_t n_c = static_cast<uint32_t>(__popcll(__ballot(st == kWClick)));
        // (walk_bias == 0: the memo-answered draws AND the bandit events of the wave in the same iteration)
        bool do_org = false, do_srch = false, do_ban = false, do_clk = false;
        if (n_s >= d.walk_search_batch || (n_s && !n_o && !n_b)) do_srch = true;
        else if (n_c && (n_c >= d.walk_click_batch || (!n_o && !n_b))) do_ban = do_clk = true;   // the bandit events that need ctr
        else if (d.walk_bias == 0u) { do_org = n_o != 0u; do_ban = n_b != 0u; }
        else if (n_o && (n_o * d.walk_bias >= n_b * 4u)) do_org = true;
        else if (n_b) do_ban = true;
        else if (n_o) do_org = true;
        else do_srch = true;
        // ---- helpers.  Inside a bandit run nothing an event reads moves (the view history, the last view, omega: sigma_omega = 0)
        // and its draws are addressed by (user, t): the lanes that sit this bandit iteration out — empty, waiting for an organic
        // iteration, a search or a click batch — take events t + 1 .. t + walk_helpers (<= kWalkHelpersMax) of the runs of the lanes that are in it
        // ("owners"), without side effects.  A helper's event counts (its row is written, the owner moves past it) iff it is
        // PLAIN — no click possible (uniform below kNoClickBelow), next state organic or bandit — and every event of the run
        // before it, the owner's own included, was plain and stayed in the run; whatever else it finds is dropped and met again
        // by the owner itself.  e_slot / e_t / e_lane: whose event this lane evaluates (its own unless it helps) ----
#ifdef RG_WALK_TIMING
        wk_kind = do_srch ? 1 : do_clk ? 3 : do_ban ? 2 : 0;
        wk_it[wk_kind] += 1;
        wk_ln[wk_kind] += do_srch ? n_s : do_clk ? n_c : do_ban ? n_b : n_o;
#endif
        bool helper = false, owner = false;
        uint32_t e_slot = slot, e_t = t, h_p = 0, h_e = 0;        // h_p: this lane's place among the dealt events (rank + h_e n_own)
        int e_lane = lane;
        uint32_t n_own = 0, n_help = 0;
        // (walk_click_join: the click batch rides along in a bandit iteration instead of taking one of its own)
        const bool ban_all = do_ban && (!do_clk || d.walk_click_join);      // the bandit lanes take part
        if (ban_all && d.walk_helpers && d.walk_click_batch && !d.aux_pclick) {
            owner = st == RG_STATE_BANDIT;
            const bool idle = !(owner || st == kPhantom || (do_org && st == RG_STATE_ORGANIC) || (do_clk && st == kWClick));
            const unsigned long long om_mask = __ballot(owner), id_mask = __ballot(idle);
            n_own = static_cast<uint32_t>(__popcll(om_mask));
            if (n_own && id_mask) {
                n_help = min(static_cast<uint32_t>(__popcll(id_mask)), d.walk_helpers * n_own);
                uint32_t* tab = reinterpret_cast<uint32_t*>(mboxf);             // [rank]{slot, t | lane << 24} (the search's mailbox: idle now)
                if (owner) {
                    h_p = prefix_in_mask(om_mask);
                    tab[2 * h_p] = slot;
                    tab[2 * h_p + 1] = t | (static_cast<uint32_t>(lane) << 24);
                }
                // (LDS instructions of a wave execute in order: only the compiler has to keep the order)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t i = prefix_in_mask(id_mask);
                if (idle && i < n_help) {
                    helper = true;
                    h_e = 1u;
#pragma unroll
                    for (uint32_t e2 = 1; e2 < kWalkHelpersMax; ++e2) h_e += i >= e2 * n_own ? 1u : 0u;
                    const uint32_t r = i - (h_e - 1u) * n_own;
                    h_p = i + n_own;
                    e_slot = tab[2 * r];
                    const uint32_t tl = tab[2 * r + 1];
                    e_t = (tl & 0xFFFFFFu) + h_e;
                    e_lane = static_cast<int>(tl >> 24);
                }
            }
        }
        hent_t* const hle = reinterpret_cast<hent_t*>(wbase) + e_lane;      // that user's history line
        const uint32_t user = static_cast<uint32_t>(d.first_user + e_slot);
        const rg_u32x4 w = rg_draw(d.seed, user, e_t, 0, RG_DRAW_EVENT);
        bool have_v = false, parked = false;
        uint32_t v = 0;
        if (do_org) {
            // =========================== organic draw, answered by the user's memo ===========================
            const bool is_o = st == RG_STATE_ORGANIC;
            const size_t row = is_o ? slot : d.n_cap;
            const float4* hp = reinterpret_cast<const float4*>(d.walk_hot + row * 32);
            const float4 h0 = hp[0];
            const uint32_t n_hot = __builtin_bit_cast(uint32_t, h0.w);
            const double u_org = d.u_override ? d.u_override[slot] : rg_uniform(w.w[0], w.w[1]);
            float uf = static_cast<float>(u_org), u_dn = uf, u_up = uf;
            if (static_cast<double>(uf) > u_org) u_dn = f32_down(uf);
            if (static_cast<double>(uf) < u_org) u_up = f32_up(uf);
            bool hit = false;
            if RG_WALK_ABL(23) { hit = true; v = (user + (t & 7u)) % d.P; }     // timing experiment: every draw a memo hit, no row read
            else {
                // the whole line in one round trip (the entries behind n_hot are not looked at), selects only
                float e[28];
#pragma unroll
                for (int i = 1; i < 8; ++i) {
                    const float4 x = hp[i];
                    e[4 * i - 4] = x.x; e[4 * i - 3] = x.y; e[4 * i - 2] = x.z; e[4 * i - 1] = x.w;
                }
#pragma unroll
                for (int j = 0; j < kHotEntries; ++j) {
                    const bool in = static_cast<uint32_t>(j) < n_hot && e[3 * j + 1] < u_dn && u_up < e[3 * j + 2];
                    hit = hit || in;
                    v = in ? __builtin_bit_cast(uint32_t, e[3 * j]) : v;
                }
            }
            have_v = is_o && hit;
            if (is_o && !hit) st = kWSlow;
            c_hit += static_cast<uint32_t>(__popcll(__ballot(have_v)));
        }
        if (do_srch) {
            // =========================== organic draw by the search over the user's prefix sums ===========================
            const bool is_s = st == kWSlow;
            const bool search = is_s;            // (a parked draw too: its chunk is where the anchored certificate starts)
            const size_t row = search ? slot : d.n_cap;
            const float4* hp = reinterpret_cast<const float4*>(d.walk_hot + row * 32);
            const float4 h0 = hp[0];
            const double S = static_cast<double>(h0.x), delta = static_cast<double>(h0.y);
            const float Q = h0.z;
            const uint32_t n_hot = __builtin_bit_cast(uint32_t, h0.w);
            double rho_rel, delta_c;                                        // the stored prefixes' roundings, the in-chunk budget (cert_correlated)
            hot_budgets(hp[7].w, delta, &rho_rel, &delta_c);
            if (kPreciseChunk) delta_c = delta;
            const double u_org = d.u_override ? d.u_override[slot] : rg_uniform(w.w[0], w.w[1]);
            const double tau = u_org * S;
            const float tauf = static_cast<float>(tau);
            // ---- super-chunk: the prefixes <= tau (an unused entry is +inf) ----
            uint32_t sc_star = 0;
            float pbf = 0.0f;
            {
                const float4* sp = reinterpret_cast<const float4*>(d.walk_scp + row * kMaxSC);
#pragma unroll
                for (int i = 0; i < kMaxSC / 4; ++i) {
                    const float4 x = sp[i];
                    const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        { const bool le = xs[q] <= tauf; sc_star += le ? 1u : 0u; pbf = le ? xs[q] : pbf; }      // (prefixes ascend: the last one <= tau is the largest)
                }
            }
            bool found = sc_star < d.n_sc;
            sc_star = min(sc_star, d.n_sc - 1);
            // ---- chunk inside it ----
            uint32_t c_star;
            {
                const uint32_t c0 = sc_star * d.sc_chunks, c1 = min(c0 + d.sc_chunks, d.n_chunks);
                const float* cp = d.cache_chunk + row * d.n_chunks;
                uint32_t cnt = 0;
                for (uint32_t cb = c0; cb < c1; cb += 16) {
                    float4 w4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        w4[i] = cb + 4 * i < c1 ? *reinterpret_cast<const float4*>(cp + cb + 4 * i) : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xs[4] = {w4[i].x, w4[i].y, w4[i].z, w4[i].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            { const bool le = xs[q] <= tauf; cnt += le ? 1u : 0u; pbf = le ? xs[q] : pbf; }
                    }
                }
                found = found && cnt < c1 - c0;
                c_star = min(c0 + cnt, c1 - 1);
            }
            const double pb = static_cast<double>(pbf);
            const float rem = static_cast<float>(tau - pb);
            // ---- the 32 products of a chunk: eight users per pass, eight lanes per user, four products per lane.  For every
            // lane that wants it: mboxf[lane] = {index in the chunk of the first product whose fp32 prefix exceeds rem_f, the
            // prefix before it, the prefix with it}, or {-1, chunk total, chunk total} ----
            auto chunk_pass = [&](bool want, uint32_t chunk, float rem_f) {
                const int grp = lane >> 3, gl = lane & 7;
                // which lanes want it, in lane order, as a list in LDS: group g of pass p serves entry 8 p + g (picking the next
                // eight set bits of the ballot one by one was ~75 instructions per pass)
                const unsigned long long want_m = __ballot(want);
                const uint32_t n_want = static_cast<uint32_t>(__popcll(want_m));
                unsigned char* wlist = reinterpret_cast<unsigned char*>(mboxf) + 64 * 12;
                if (want) wlist[prefix_in_mask(want_m)] = static_cast<unsigned char>(lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t wbase_i = 0; wbase_i < n_want; wbase_i += 8) {
                    const bool has = wbase_i + static_cast<uint32_t>(grp) < n_want;
                    const int src = has ? static_cast<int>(wlist[wbase_i + grp]) : -1;
                    const int s2 = has ? src : 0;
                    const uint32_t cs = static_cast<uint32_t>(__shfl(static_cast<int>(chunk), s2));
                    const float Qs = __shfl(Q, s2);
                    const float rems = __shfl(rem_f, s2);                // what is left of u S~ at the chunk's start
                    const float4* gp = reinterpret_cast<const float4*>(d.gamma32t + (static_cast<size_t>(cs) * K2) * 32) + gl;
                    const float4 l = *(reinterpret_cast<const float4*>(d.mu32 + cs * 32) + gl);
                    float e0, e1, e2, e3;
                    if constexpr (kPreciseChunk) {
                        // float64 dot of the fp32 tables (products exact, sums to 1e-16), ONE fp32 rounding of the exp2 argument:
                        // the recomputed terms then carry the error xh_delta's `rec` grants them (rg_common.hpp)
                        double lx = static_cast<double>(l.x), ly = static_cast<double>(l.y), lz = static_cast<double>(l.z), lw = static_cast<double>(l.w);
#pragma unroll
                        for (int kh = 0; kh < K2; kh += KH) {
                            float4 gk[KH];
#pragma unroll
                            for (int k = 0; k < KH; ++k) gk[k] = gp[(kh + k) * 8];
#pragma unroll
                            for (int k = 0; k < KH; ++k) {
                                const double wk = static_cast<double>(__shfl(om[kh + k], s2));
                                lx = fma(static_cast<double>(gk[k].x), wk, lx); ly = fma(static_cast<double>(gk[k].y), wk, ly);
                                lz = fma(static_cast<double>(gk[k].z), wk, lz); lw = fma(static_cast<double>(gk[k].w), wk, lw);
                            }
                            asm volatile("" : "+v"(lx), "+v"(ly), "+v"(lz), "+v"(lw));
                        }
                        const double Qd = static_cast<double>(Qs);
                        e0 = __builtin_amdgcn_exp2f(static_cast<float>(fma(lx, kLog2e64, -Qd))); e1 = __builtin_amdgcn_exp2f(static_cast<float>(fma(ly, kLog2e64, -Qd)));
                        e2 = __builtin_amdgcn_exp2f(static_cast<float>(fma(lz, kLog2e64, -Qd))); e3 = __builtin_amdgcn_exp2f(static_cast<float>(fma(lw, kLog2e64, -Qd)));
                    } else {
                        float4 lf = l;
#pragma unroll
                        for (int kh = 0; kh < K2; kh += KH) {
                            float4 gk[KH];
#pragma unroll
                            for (int k = 0; k < KH; ++k) gk[k] = gp[(kh + k) * 8];
#pragma unroll
                            for (int k = 0; k < KH; ++k) {
                                const float wk = __shfl(om[kh + k], s2);
                                lf.x = fmaf(gk[k].x, wk, lf.x); lf.y = fmaf(gk[k].y, wk, lf.y);
                                lf.z = fmaf(gk[k].z, wk, lf.z); lf.w = fmaf(gk[k].w, wk, lf.w);
                            }
                            asm volatile("" : "+v"(lf.x), "+v"(lf.y), "+v"(lf.z), "+v"(lf.w));
                        }
                        e0 = __builtin_amdgcn_exp2f(fmaf(lf.x, kLog2e, -Qs)); e1 = __builtin_amdgcn_exp2f(fmaf(lf.y, kLog2e, -Qs));
                        e2 = __builtin_amdgcn_exp2f(fmaf(lf.z, kLog2e, -Qs)); e3 = __builtin_amdgcn_exp2f(fmaf(lf.w, kLog2e, -Qs));
                    }
                    const float q0 = e0, q1 = q0 + e1, q2 = q1 + e2, q3 = q2 + e3;
                    float inc = q3;
#pragma unroll
                    for (int o2 = 1; o2 < 8; o2 <<= 1) {
                        const float y = __shfl_up(inc, o2, 8);
                        if (gl >= o2) inc += y;
                    }
                    float ex = __shfl_up(inc, 1, 8);
                    if (gl == 0) ex = 0.0f;
                    // the product in fp32 (which product is only a proposal: the certificate is taken in float64 from the two
                    // prefixes around it and rejects a wrong one)
                    const float x0 = ex + q0, x1 = ex + q1, x2 = ex + q2, x3 = ex + q3;
                    const int j0 = x0 > rems ? 0 : x1 > rems ? 1 : x2 > rems ? 2 : x3 > rems ? 3 : -1;
                    const unsigned long long hits = __ballot(has && j0 >= 0);
                    const uint32_t gmask = static_cast<uint32_t>(hits >> (8 * grp)) & 0xFFu;
                    if (has) {
                        if (gmask) {
                            if (gl == __builtin_ctz(gmask)) {
                                mboxf[src * 3] = static_cast<float>(4 * gl + j0);
                                mboxf[src * 3 + 1] = j0 == 0 ? ex : j0 == 1 ? x0 : j0 == 2 ? x1 : x2;
                                mboxf[src * 3 + 2] = j0 == 0 ? x0 : j0 == 1 ? x1 : j0 == 2 ? x2 : x3;
                            }
                        } else if (gl == 7) { mboxf[src * 3] = -1.0f; mboxf[src * 3 + 1] = inc; mboxf[src * 3 + 2] = inc; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            };
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            chunk_pass(search, c_star, rem);
            bool ok = false;
            if (search) {
                const int idx = static_cast<int>(mboxf[lane * 3]);
                const CertLin ct = cert_correlated(S, pb, static_cast<double>(mboxf[lane * 3 + 1]), static_cast<double>(mboxf[lane * 3 + 2]), delta, rho_rel, delta_c);
                v = c_star * 32 + static_cast<uint32_t>(max(idx, 0));
                const bool lo_ok = v == 0 || u_org * ct.den_lo > ct.num_lo;
                const bool hi_ok = v == d.P - 1 || u_org * ct.den_hi < ct.num_hi;
                ok = found && idx >= 0 && v < d.P && ct.valid && lo_ok && hi_ok;
                if (ok && n_hot < static_cast<uint32_t>(kHotEntries)) {
                    // memoise the certified u-interval of v, rounded inwards (and a hair more for the float64 roundings of
                    // the inequality above): u in (lo, hi) implies both conditions, whatever u
                    float lo = -1.0f, hi = 2.0f;
                    if (v != 0) {
                        const double x = ct.num_lo / ct.den_lo * (1.0 + 1e-14);
                        lo = static_cast<float>(x);
                        if (static_cast<double>(lo) < x) lo = f32_up(lo);
                    }
                    if (v != d.P - 1) {
                        const double x = ct.num_hi / ct.den_hi * (1.0 - 1e-14);
                        hi = static_cast<float>(x);
                        if (static_cast<double>(hi) > x) hi = f32_down(hi);
                    }
                    float* hf = d.walk_hot + static_cast<size_t>(slot) * 32;
                    hf[4 + 3 * n_hot] = __builtin_bit_cast(float, v);
                    hf[5 + 3 * n_hot] = lo;
                    hf[6 + 3 * n_hot] = hi;
                    reinterpret_cast<uint32_t*>(hf)[3] = n_hot + 1u;
                }
            }
            // ---- uncertified: with the user's float64 sums, or park the user until they exist ----
            const bool need64 = is_s && !ok;
            const bool have64 = need64 && d.f64_valid[slot] != 0;
            parked = need64 && !have64;
            // ANCHORED certificate.  The float64 sums of a user (k_exact_prefix left them as prefixes at the end of every 64
            // products, on the same reference Q as the fp32 exps) pin the prefix at the start of the draw's 64-product chunk to
            // ~1e-13 S; only the part INSIDE the chunk is fp32, so the same test with delta applied to that part alone — and
            // 1e-12 S of slack for the anchors' own roundings — certifies all but ~2.5 % of the draws the plain certificate
            // rejected (those need a heavy product earlier in the same chunk).  Lane-parallel, like the search: what is left
            // for the wave-serial float64 pick below is ~0.1 % of the organic draws instead of 3 %.
            bool got64 = false;
            if (__ballot(have64)) {
                const double* pfx = d.exact_sums + static_cast<size_t>(have64 ? slot : 0u) * n_cc;
                const double S64 = pfx[n_cc - 1];
                const double target = u_org * S64;
                uint32_t cc = min(c_star >> 1, n_cc - 1u);               // the fp32 search's chunk is (nearly always) the float64 one
                double hi64 = pfx[cc], lo64 = cc ? pfx[cc - 1] : 0.0;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    if (!(lo64 <= target) && cc > 0u) { --cc; hi64 = lo64; lo64 = cc ? pfx[cc - 1] : 0.0; }
                    else if (!(target < hi64) && cc + 1u < n_cc) { ++cc; lo64 = hi64; hi64 = pfx[cc]; }
                }
                const bool anchored = have64 && lo64 <= target && target < hi64;
                const float rem1 = static_cast<float>(target - lo64);
                chunk_pass(anchored, 2u * cc, rem1);
                const int idx1 = anchored ? static_cast<int>(mboxf[lane * 3]) : 0;
                const float a1 = mboxf[lane * 3 + 1], b1 = mboxf[lane * 3 + 2];
                const bool in2 = anchored && idx1 < 0;
                __builtin_amdgcn_wave_barrier();
                chunk_pass(in2, 2u * cc + 1u, rem1 - a1);
                if (anchored) {
                    int ix = idx1;
                    float fa = a1, fb = b1;
                    uint32_t va = 64u * cc + static_cast<uint32_t>(max(idx1, 0));
                    if (in2) {
                        ix = static_cast<int>(mboxf[lane * 3]);
                        fa = a1 + mboxf[lane * 3 + 1]; fb = a1 + mboxf[lane * 3 + 2];
                        va = 64u * cc + 32u + static_cast<uint32_t>(max(ix, 0));
                    }
                    // (a computed in-chunk prefix s = e (1 + eps), |eps| <= delta: the true e is at most s / (1 - delta) <=
                    // s (1 + dp), dp = delta (1 + 2 delta) as in cert_correlated, and at least s / (1 + delta) >= s (1 - delta))
                    const double slack = 1.0e-12 * S64;
                    const double dp = delta_c * (1.0 + 2.0 * delta_c);       // (the in-chunk terms are recomputed ones)
                    const bool lo_ok = va == 0u || lo64 + static_cast<double>(fa) * (1.0 + dp) + slack < target;
                    const bool hi_ok = va == d.P - 1 || target + slack < lo64 + static_cast<double>(fb) * (1.0 - delta_c);
                    got64 = ix >= 0 && va < d.P && lo_ok && hi_ok;
                    if (got64) v = va;
                }
            }
            c_anch += static_cast<uint32_t>(__popcll(__ballot(got64)));
            c_pick += static_cast<uint32_t>(__popcll(__ballot(have64)));     // resolved with float64 sums: anchored or picked
            unsigned long long picks = __ballot(have64 && !got64);
            while (picks) {
                const int L = __builtin_ctzll(picks);
                picks &= picks - 1;
                const uint32_t s_slot = static_cast<uint32_t>(__shfl(static_cast<int>(slot), L));
                const double s_u = __shfl(u_org, L);
                const double M = static_cast<double>(d.exact_ref[s_slot]) * 0.69314718055994530942;
                const uint32_t pv = exact_pick_pfx(d, d.exact_sums + static_cast<size_t>(s_slot) * n_cc,
                                                   d.omega + static_cast<size_t>(s_slot) * d.OMS, M, s_u, lane);
                if (lane == L) v = pv;
                __builtin_amdgcn_wave_barrier();
            }
            if (parked) {
                d.park_t[slot] = t | (static_cast<uint32_t>(RG_STATE_ORGANIC) << 24) | (1u << 27);
                d.exact_ref[slot] = Q;
            }
            have_v = is_s && !parked;
            if (have_v) pend = false;
        }
        // ---- park list entries for the users parked in this step ----
        const unsigned long long pmask = __ballot(parked);
        if (pmask) {
            const uint32_t np = static_cast<uint32_t>(__popcll(pmask));
            if (park_next + np > park_end) {
                for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
                uint32_t base = 0;
                if (lane == 0) base = static_cast<uint32_t>(atomicAdd(d.q_park, 64ull));
                base = __builtin_amdgcn_readfirstlane(base);
                park_next = base; park_end = base + 64;
            }
            if (parked) { d.park_list[out_base + park_next + prefix_in_mask(pmask)] = slot; st = kEmpty; }      // (its line: written back at the refill)
            park_next += np;
        }
        // =========================== bandit event: the policy's act and the click ===========================
        const bool mine_clk = do_clk && st == kWClick && !helper;          // its uniform is known to be >= kNoClickBelow
        bool is_ban = mine_clk || (ban_all && (st == RG_STATE_BANDIT || helper));
        const bool is_ph = ban_all && st == kPhantom && !helper;
        double ps = 1.0;
        uint32_t a = 0;
        bool click = false, click_known = false;
        double ctr = 0.0;
        if (do_ban) {
            if ((is_ban || is_ph) && RG_WALK_ABL(28)) { a = user % d.P; ps = 1.0; }      // timing experiment: no policy act
            else
            if (is_ban || is_ph) {
                if (HIST == 2) {
                    // the same act on the COMPACT line (prefix form): the first product whose cumulative count exceeds u x views
                    // = the number of words below the key (Thi + 1) << 16 — the last word of each 8-word segment, then the segment
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, e_t, 0, RG_DRAW_POLICY);
                    const double u1 = rg_uniform(pw.w[2], pw.w[3]);
                    const hent_t* hr = hist_row(d, e_slot);
                    const uint32_t* hw32 = reinterpret_cast<const uint32_t*>(hle);     // word w: hw32[(w >> 1) * 128 + (w & 1)]
                    const uint32_t h0 = hw32[0];
                    const uint32_t p7 = hw32[3 * 128 + 1], p15 = hw32[7 * 128 + 1], p23 = hw32[11 * 128 + 1], p31 = hw32[15 * 128 + 1];
                    const uint32_t nd = h0 & 0x7FFFu;
                    const double sum = static_cast<double>(h0 >> 15);
                    const double T = u1 * sum;
                    const uint32_t Thi = static_cast<uint32_t>(fmin(floor(T * (1.0 + 0x1p-36)), 4294967295.0));
                    const uint32_t Tlo = static_cast<uint32_t>(fmin(ceil(T * (1.0 - 0x1p-36)), 4294967295.0));
                    const bool over = Thi >= 65535u;                                     // (u x views at the top of the range: no entry exceeds it)
                    const uint32_t khi = over ? 0u : (Thi + 1u) << 16;                   // prefix <= Thi  <=>  word < khi
                    const uint32_t seg = (p7 < khi ? 1u : 0u) + (p15 < khi ? 1u : 0u) + (p23 < khi ? 1u : 0u);
                    uint32_t x[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const hent_t y = hle[(seg * 4 + j) * 64];
                        x[2 * j] = static_cast<uint32_t>(y); x[2 * j + 1] = static_cast<uint32_t>(y >> 32);
                    }
                    if (seg == 0u) x[0] = 0u;                                             // (the header: counted, prefix 0)
                    uint32_t in_seg = 0u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) in_seg += x[j] < khi ? 1u : 0u;
                    const uint32_t idx = seg * 8u + in_seg;                              // first entry with prefix > Thi (32: none in the line)
                    // its word and the one before it (the entry before a segment's first: the segment end read above)
                    uint32_t w_at = 0xFFFFFFFFu, w_prev = seg == 0u ? 0u : (seg == 1u ? p7 : (seg == 2u ? p15 : p23));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        w_at = in_seg == static_cast<uint32_t>(j) ? x[j] : w_at;
                        w_prev = in_seg == static_cast<uint32_t>(j + 1) ? x[j] : w_prev;
                    }
                    bool found = !over && idx <= nd && idx < kHcLine;
                    // an entry inside the 2^-36 band of u x views: only the last one at or below Thi can be (prefixes ascend)
                    bool amb = idx >= 2u && (w_prev >> 16) >= Tlo;
                    uint32_t c_f = (w_at >> 16) - (idx >= 2u ? (w_prev >> 16) : 0u);
                    a = w_at & 0xFFFFu;
                    uint32_t C = p31 >> 16;                                              // (nd >= 31: the line's last prefix)
                    // (a helper leaves a history beyond the line to its owner: with 64 lanes evaluating, half the iterations had
                    // some lane in this loop)
                    for (uint32_t base = kHcLine; base <= nd && !found && !helper; base += kHistRegs) {      // longer histories: from the row
                        hent_t f[kHistRegs];
                        hist_load_line(hr + base, f);
#pragma unroll
                        for (int i = 0; i < kHistRegs; ++i)
                            if (base + i <= nd && !found) {
                                C += h_cnt(f[i]);
                                if (C > Thi) { found = true; a = h_prod(f[i]); c_f = h_cnt(f[i]); }
                                else if (C >= Tlo) amb = true;
                            }
                    }
                    if (found && !amb) ps = static_cast<double>(c_f) / sum;
                    else if (helper) is_ban = false;      // (dropped: not plain)
                    else {
                        // inside the band (~1e-10 of the acts): numpy's arithmetic over the viewed products, as below
                        auto ent = [&](uint32_t i, uint32_t* prev) -> hent_t {       // (product, count) of entry i, walked in order
                            if (i >= kHcLine) return hr[i];
                            const uint32_t w = hw32[(i >> 1) * 128 + (i & 1u)];
                            const uint32_t cnt = (w >> 16) - *prev;
                            *prev = w >> 16;
                            return (static_cast<hent_t>(w & 0xFFFFu) << 32) | cnt;
                        };
                        double last = 0.0;
                        uint32_t pv = 0u;
                        for (uint32_t i = 1; i <= nd; ++i) last += static_cast<double>(h_cnt(ent(i, &pv))) / sum;
                        double acc = 0.0, pa = 0.0;
                        a = d.P - 1;
                        bool fnd = false;
                        pv = 0u;
                        for (uint32_t i = 1; i <= nd && !fnd; ++i) {
                            const hent_t y = ent(i, &pv);
                            const double p = static_cast<double>(h_cnt(y)) / sum;
                            acc += p;
                            if (!(acc / last <= u1)) { a = h_prod(y); pa = p; fnd = true; }
                        }
                        ps = pa;
                    }
                } else
                if (HIST) {
                    // OrganicUserEventCounterModel.act (organic_user_count.py:45-96; exploit_explore, epsilon = 0,
                    // select_randomly: the host instantiates HIST = 1 for this form only) on the history line in LDS: decided
                    // by integer prefix counts outside a 2^-36 band (see policy_act), by the float64 cdf walk inside it
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, e_t, 0, RG_DRAW_POLICY);
                    const double u1 = rg_uniform(pw.w[2], pw.w[3]);
                    const hent_t h0 = hle[0];
                    const uint32_t nd = h_cnt(h0);
                    const double sum = static_cast<double>(h_prod(h0));
                    const hent_t* hr = hist_row(d, e_slot);
                    const double T = u1 * sum;
                    const uint32_t Thi = static_cast<uint32_t>(fmin(floor(T * (1.0 + 0x1p-36)), 4294967295.0));
                    const uint32_t Tlo = static_cast<uint32_t>(fmin(ceil(T * (1.0 - 0x1p-36)), 4294967295.0));
                    uint32_t C = 0, c_f = 0;
                    bool found = false, amb = false;
                    {
                        // the 15 entries of the line at once (one LDS round trip), then selects only: an entry-by-entry loop
                        // with its two exits compiled to 15 dependent round trips and 30 branches
                        hent_t e[16];
#pragma unroll
                        for (int i = 1; i < 16; ++i) e[i] = hle[i * 64];
#pragma unroll
                        for (int i = 1; i < 16; ++i) {
                            const bool in = static_cast<uint32_t>(i) <= nd;
                            const uint32_t cnt = in ? h_cnt(e[i]) : 0u;
                            C += cnt;
                            const bool take = in && !found && C > Thi;
                            amb = amb || (in && !found && !take && C >= Tlo);
                            a = take ? h_prod(e[i]) : a;
                            c_f = take ? cnt : c_f;
                            found = found || take;
                        }
                    }
                    for (uint32_t base = 16; base <= nd && !found && !helper; base += kHistRegs) {      // longer histories: from the row
                        hent_t f[kHistRegs];
                        hist_load_line(hr + base, f);
#pragma unroll
                        for (int i = 0; i < kHistRegs; ++i)
                            if (base + i <= nd && !found) {
                                C += h_cnt(f[i]);
                                if (C > Thi) { found = true; a = h_prod(f[i]); c_f = h_cnt(f[i]); }
                                else if (C >= Tlo) amb = true;
                            }
                    }
                    if (found && !amb) ps = static_cast<double>(c_f) / sum;
                    else if (helper) is_ban = false;      // (dropped: not plain)
                    else {
                        // inside the band (~1e-10 of the acts): numpy's arithmetic — p_i = count_i / sum, cdf = cumsum(p) / last,
                        // first index with cdf > u1 — over the viewed products (zero entries add exactly 0.0)
                        double last = 0.0;
                        for (uint32_t i = 1; i <= nd; ++i) last += static_cast<double>(h_cnt(i < 16 ? hle[i * 64] : hr[i])) / sum;
                        double acc = 0.0, pa = 0.0;
                        a = d.P - 1;
                        bool fnd = false;
                        for (uint32_t i = 1; i <= nd && !fnd; ++i) {
                            const hent_t x = i < 16 ? hle[i * 64] : hr[i];
                            const double p = static_cast<double>(h_cnt(x)) / sum;
                            acc += p;
                            if (!(acc / last <= u1)) { a = h_prod(x); pa = p; fnd = true; }
                        }
                        ps = pa;
                    }
                } else if (d.policy == RG_POLICY_LAST_VIEW_TABLE) {
                    const uint32_t p = d.lpv[e_slot];
                    ps = d.pol_ps ? static_cast<double>(d.pol_ps[p]) : 1.0;
                    a = static_cast<uint32_t>(d.pol_table[p]);
                } else {        // agent = None / RandomAgent: uniform over P from the env / the agent stream
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, e_t, 0, RG_DRAW_POLICY);
                    ps = 1.0 / static_cast<double>(d.P);
                    a = rg_bounded(pw.w[0], pw.w[1], d.P);
                }
            }
            if (is_ph) {       // final step_offline(done = True): the act above, reward 0 (abstract.py:223-233,311-316)
                rg_event e;
                e.u = user; e.t = t; e.code = RG_EV_BANDIT | RG_EV_PHANTOM | a;
                e.ps = static_cast<float>(ps);
                d.phantom[slot] = e;
                d.phantom_ps[slot] = ps;
                d.has_phantom[slot] = 1;
                st = kEmpty;
            }
            c_ph += static_cast<uint32_t>(__popcll(__ballot(is_ph)));
            if (is_ban && RG_WALK_ABL(24)) { click = false; click_known = true; }        // timing experiment: no beta row
            else
            if (is_ban && !d.aux_pclick && !mine_clk) {
                // no click below kNoClickBelow; the 3 % above it wait (kWClick) until walk_click_batch lanes of the wave do: the
                // beta row is a memory round trip the whole wave would otherwise sit out in every bandit iteration
                if (rg_uniform(w.w[0], w.w[1]) < kNoClickBelow) { click = false; click_known = true; }
                else if (d.walk_click_batch) { if (!helper) st = kWClick; is_ban = false; }      // (a helper drops the event: not plain)
            }
            if (is_ban && !d.aux_pclick && !click_known) {
                const int dec = click_decide32<KC>(d.beta32 + static_cast<size_t>(a) * d.KB4, [&](int k) { return om[k]; }, d.K, d.KB4,
                                                   static_cast<float>(d.mu_b[a]), rg_uniform(w.w[0], w.w[1]));
                if (dec >= 0) { click = dec != 0; click_known = true; }
            }
            if (is_ban && !click_known) {
                const double* b = d.beta + static_cast<size_t>(a) * d.K;
                const double* omd = d.omega + static_cast<size_t>(slot) * d.OMS;
                double x = 0.0;
                for (uint32_t k0 = 0; k0 < d.K; k0 += 8) {
                    double wv[8], bv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t k = min(k0 + i, d.K - 1);
                        wv[i] = omd[k];
                        bv[i] = b[k];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (k0 + i < d.K) x += bv[i] * wv[i];
                }
                ctr = ff64(x + d.mu_b[a]);
                const double p0 = 1.0 - ctr;
                click = (p0 / (p0 + ctr)) <= rg_uniform(w.w[0], w.w[1]);
            }
        }
        // =========================== the event's row, the view, the transition ===========================
        const bool ev = have_v || is_ban;          // (a helper's: a candidate)
        // the transition of every event (abstract.py:160-185), side effects below
        int ns = RG_STATE_STOP;
        bool limit = false, ended = false;
        if (ev) {
            const double u_trans = rg_uniform(w.w[2], w.w[3]);
            const double c0 = have_v ? d.cdf_o0 : d.cdf_b0, c1 = have_v ? d.cdf_o1 : d.cdf_b1;
            ns = (c0 <= u_trans) + (c1 <= u_trans);
            if (click) ns = RG_STATE_ORGANIC;                  // abstract.py:180-181 (sigma_omega == 0: no drift to apply)
            const bool organic_only = (d.first_user + e_slot) < d.organic_only_below;
            if (organic_only && ns != RG_STATE_ORGANIC) { ns = RG_STATE_STOP; ended = true; }
            else if (ns == RG_STATE_STOP) { ns = kPhantom; ended = true; }      // the phantom row's act: this lane's next bandit iteration
            else if (e_t + 2 >= kMaxSteps) { ns = RG_STATE_STOP; ended = true; limit = true; }
        }
        // which of the helpers' events count, and how far each owner moves
        bool counts = ev && !helper;
        uint32_t adv = 0u;
        if (n_help) {
            unsigned char* fl = reinterpret_cast<unsigned char*>(mboxf) + 512;          // [place] bit 0: stayed in the run, bit 1: plain
            const bool plain = is_ban && !click && !ended;                               // (next state organic or bandit)
            const bool stays = plain && ns == RG_STATE_BANDIT;
            if (owner || helper) fl[h_p] = static_cast<unsigned char>((stays ? 1u : 0u) | (plain ? 2u : 0u));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (helper) {
                bool ok = plain;
#pragma unroll
                for (uint32_t e2 = 1; e2 <= kWalkHelpersMax; ++e2)                       // the run's events before this one
                    if (e2 <= h_e) ok = ok && (fl[h_p - e2 * n_own] & 1u) != 0u;
                counts = ok;
            } else if (owner && stays) {
                bool alive = true;
#pragma unroll
                for (uint32_t e2 = 1; e2 <= kWalkHelpersMax; ++e2) {
                    const uint32_t pl = h_p + e2 * n_own;                                // (helper of idle rank pl - n_own)
                    const uint32_t f = pl < n_own + n_help ? fl[pl] : 0u;
                    const bool ok = alive && (f & 2u) != 0u;
                    if (ok) { adv = e2; ns = (f & 1u) ? RG_STATE_BANDIT : RG_STATE_ORGANIC; }
                    alive = ok && (f & 1u) != 0u;
                }
            }
        }
#ifdef RG_WALK_TIMING
        wk_hv += static_cast<unsigned long long>(__popcll(__ballot(helper && counts)));
        wk_hd += n_help;
#endif
        const unsigned long long rowm = __ballot(counts);
        if (rowm) {
            const uint32_t nrow = static_cast<uint32_t>(__popcll(rowm));
            if (row_next + nrow > row_end) {
                for (uint64_t r = row_next + lane; r < row_end; r += 64)
                    if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&d.counters[kCntTailRows], static_cast<unsigned long long>(chunk_rows));
                base = readfirstlane_u64(base);
                row_next = base; row_end = base + chunk_rows;
            }
            const uint64_t my_row = row_next + prefix_in_mask(rowm);
            row_next += nrow;
            if (counts && d.log && my_row < d.log_cap && !RG_WALK_ABL(25)) {
                rg_event e;
                e.u = user; e.t = e_t;
                e.code = have_v ? v : (RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a);
                e.ps = have_v ? __builtin_nanf("") : static_cast<float>(ps);
                d.log[my_row] = e;
                if (is_ban && d.aux_ps) d.aux_ps[my_row] = ps;
                if (is_ban && d.aux_pclick) d.aux_pclick[my_row] = ctr;
            }
            c_org += static_cast<uint32_t>(__popcll(__ballot(have_v)));
            c_ban += static_cast<uint32_t>(__popcll(__ballot(is_ban && counts)));
            c_clicks += static_cast<uint32_t>(__popcll(__ballot(is_ban && click && counts)));
            if (have_v) {
                if (d.lpv) d.lpv[slot] = v;
                if (HIST == 2 && !RG_WALK_ABL(27)) {
                    // ViewsFeaturesProvider.observe (agents/abstract.py:347-358) on the COMPACT line in LDS: position and hit of v,
                    // then the line with the prefixes from there on raised by the view (and shifted by the new product)
                    hent_t* hr = hist_row(d, slot);
                    uint32_t e[kHcLine];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const hent_t y = hl[i * 64];
                        e[2 * i] = static_cast<uint32_t>(y); e[2 * i + 1] = static_cast<uint32_t>(y >> 32);
                    }
                    const uint32_t h0 = e[0];
                    const uint32_t nd = h0 & 0x7FFFu;
                    // first entry with product >= v (min(nd, 31) + 1 if none: an unused word holds 0xFFFF, above every product);
                    // the products ascend, so v is in the line iff it is the product of that entry (one more look at the line)
                    uint32_t pos = 1;
#pragma unroll
                    for (int i = 1; i < static_cast<int>(kHcLine); ++i) pos += static_cast<uint16_t>(e[i]) < static_cast<uint16_t>(v) ? 1u : 0u;
                    uint32_t* hw32 = reinterpret_cast<uint32_t*>(hl);           // word w of this lane's line: hw32[(w >> 1) * 128 + (w & 1)]
                    const uint32_t wp = pos < kHcLine ? hw32[(pos >> 1) * 128u + (pos & 1u)] : 0xFFFFFFFFu;
                    const bool hit = (wp & 0xFFFFu) == v;
                    const bool room = nd < kHcLine - 1u;
                    if (hit || room) {
                        const bool full = !hit && nd + 1 >= d.hist_cap;
                        if (full) atomicAdd(&d.counters[RG_CNT_HIST_OVERFLOW], 1ull);
                        // new word i: below pos unchanged; from pos on raised by the view (a repeat view), or — a new product —
                        // the old word below it raised (entries pos + 1 .. nd + 1; its own word is written last).  Which words
                        // are raised / shifted: two bit masks per lane, two instructions per word each
                        const uint32_t last = min(hit ? nd : nd + 1u, kHcLine - 1u);          // last entry of the line in use after the view
                        const uint32_t upto = 0xFFFFFFFFu >> (31u - last), from = 0xFFFFFFFFu << pos;      // (pos <= 31 here)
                        const uint32_t shifted = (full || hit) ? 0u : (upto & (from << 1));
                        const uint32_t raised = full ? 0u : (hit ? (upto & from) : shifted);
                        auto word = [&](int i) -> uint32_t {
                            if (i == 0) return full ? h0 : h0 + (1u << 15) + (hit ? 0u : 1u);
                            const uint32_t x = ((shifted >> i) & 1u) ? e[i - 1] : e[i];
                            return x + (((raised >> i) & 1u) << 16);
                        };
#pragma unroll
                        for (int i = 15; i >= 0; --i) {       // (from the top: word i reads e[i - 1], nothing above it)
                            const uint32_t f0 = word(2 * i), f1 = word(2 * i + 1);
                            hl[i * 64] = static_cast<hent_t>(f0) | (static_cast<hent_t>(f1) << 32);
                        }
                        if (!hit && !full) {
                            // the new product's own word: the prefix of the entry before it (unchanged above; 0 before the first) + 1
                            const uint32_t wq = pos > 1u ? hw32[((pos - 1u) >> 1) * 128u + ((pos - 1u) & 1u)] : 0u;
                            hw32[(pos >> 1) * 128u + (pos & 1u)] = ((wq & 0xFFFF0000u) + 0x10000u) | v;
                        }
                        hdirty = true;
                    } else if (v > (e[kHcLine - 1] & 0xFFFFu)) {
                        // a longer history, v behind the line's 31 products: entries >= 32 of the row (always current), the
                        // header in LDS — one round trip, nothing to read back
                        if (nd + 1 >= d.hist_cap) {
                            flush_hist(hdirty);
                            hdirty = false;
                            history_add(d, slot, v);
                            const hent_t hh = hr[0];
                            hw32[0] = (h_prod(hh) << 15) | h_cnt(hh);
                        } else {
                            const uint32_t fresh = history_tail_add(hr, nd, v, kHcLine);
                            hw32[0] = h0 + (1u << 15) + fresh;
                            hdirty = true;
                        }
                    } else {
                        // a new product inside the line of a longer history (its last product moves to the row): the general
                        // insertion on the row, then the line again
                        flush_hist(hdirty);
                        hdirty = false;
                        history_add(d, slot, v);
                        load_compact(slot);
                    }
                } else
                if (HIST && !RG_WALK_ABL(27)) {
                    // ViewsFeaturesProvider.observe (agents/abstract.py:347-358) on the line in LDS, written through to the row
                    hent_t* hr = hist_row(d, slot);
                    const hent_t h0 = hl[0];
                    const uint32_t nd = h_cnt(h0);
                    const hent_t key = static_cast<hent_t>(v) << 32;
                    // the line in registers (one LDS round trip): position of v, whether it is there
                    hent_t e[17];
                    e[0] = h0; e[16] = 0ull;
#pragma unroll
                    for (int i = 1; i < 16; ++i) e[i] = hl[i * 64];
                    uint32_t pos = 1;                       // first entry with product >= v (min(nd, 15) + 1 if none)
                    bool hit = false;
#pragma unroll
                    for (int i = 1; i < 16; ++i) {
                        const bool in = static_cast<uint32_t>(i) <= nd;
                        pos += (in && e[i] < key) ? 1u : 0u;
                        hit = hit || (in && h_prod(e[i]) == v);
                    }
                    if (nd < 15u && RG_WALK_ABL(29)) {}                        // timing experiment: no insertion into the line
                    else if (nd >= 15u && RG_WALK_ABL(26)) {}                  // timing experiment: no insertion into a longer history
                    else
                    if (nd < 15u || hit) {
                        // the new line by selects, written back whole to LDS (the row gets it when the lane lets go of the
                        // user): no data-dependent branch, no dependent loads.  A longer history whose line holds v: the same
                        const bool full = !hit && nd + 1 >= d.hist_cap;
                        if (full) atomicAdd(&d.counters[RG_CNT_HIST_OVERFLOW], 1ull);
                        hent_t f[16];
                        f[0] = full ? h0 : h0 + (1ull << 32) + (hit ? 0ull : 1ull);
#pragma unroll
                        for (int i = 1; i < 16; ++i) {
                            const uint32_t ui = static_cast<uint32_t>(i);
                            const hent_t shifted = ui < pos ? e[i] : (ui == pos ? (key | 1ull) : e[i - 1]);
                            const hent_t bumped = ui == pos ? e[i] + 1ull : e[i];
                            f[i] = full ? e[i] : (hit ? bumped : shifted);
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) hl[i * 64] = f[i];
                        hdirty = true;                                         // (written back when the lane lets go of the user)
                    } else if (v > h_prod(e[15])) {
                        // a longer history, v behind the line's 15 products: entries >= 16 of the row (always current), the
                        // header in LDS — one round trip, nothing to read back
                        if (nd + 1 >= d.hist_cap) {
                            // (a new product would not fit: the general insertion decides and counts the overflow)
                            flush_hist(hdirty);
                            hdirty = false;
                            history_add(d, slot, v);
                            hl[0] = hr[0];
                        } else {
                            const uint32_t fresh = history_tail_add(hr, nd, v);
                            hl[0] = h0 + (1ull << 32) + fresh;
                            hdirty = true;
                        }
                    } else {
                        // a new product inside the line of a longer history (its last product moves to the row): the general
                        // insertion on the row, then the line again
                        flush_hist(hdirty);
                        hdirty = false;
                        history_add(d, slot, v);
                        const ulonglong2* hr2 = reinterpret_cast<const ulonglong2*>(hr);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const ulonglong2 x = hr2[i];
                            hl[(2 * i) * 64] = x.x; hl[(2 * i + 1) * 64] = x.y;
                        }
                    }
                }
            }
            if (ev && !helper) {
                if (ended) {
                    d.n_events[slot] = t + 1;
                    c_maxt = max(c_maxt, t + 1);       // (maximum over the wave taken once at the end: a per-lane maximum in one register)
                }
                if (limit) c_limit += 1;
                if (ns == RG_STATE_STOP) st = kEmpty;        // (its line is written back when the lane is refilled, or at the end)
                else { st = ns; t += 1u + adv; }       // (adv: the helpers' events of this run that count)
            }
        }
    }
#ifdef RG_WALK_TIMING
    if (lane == 0) {
        for (int k = 0; k < 4; ++k) { atomicAdd(&g_walk_stat[4 * k], wk_it[k]); atomicAdd(&g_walk_stat[4 * k + 1], wk_ln[k]); atomicAdd(&g_walk_stat[4 * k + 2], wk_cy[k]); }
        atomicAdd(&g_walk_stat[16], wk_hv); atomicAdd(&g_walk_stat[17], wk_hd);
    }
#endif
    // ---- leftovers of the reserved chunks, counters ----
    {
        const DevSim& d = *(const DevSim*)kargs;
        for (uint64_t r = row_next + lane; r < row_end; r += 64)
            if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
        for (uint32_t r = park_next + lane; r < park_end; r += 64) d.park_list[out_base + r] = 0xFFFFFFFFu;
        for (int o = 32; o > 0; o >>= 1) {
            c_maxt = max(c_maxt, static_cast<uint32_t>(__shfl_xor(static_cast<int>(c_maxt), o)));
            c_limit += static_cast<uint32_t>(__shfl_xor(static_cast<int>(c_limit), o));
        }
        if (lane == 0) {
            if (c_org) atomicAdd(&d.counters[kCntTailOrganic], static_cast<unsigned long long>(c_org));
            if (c_ban) atomicAdd(&d.counters[kCntTailBandit], static_cast<unsigned long long>(c_ban));
            if (c_clicks) atomicAdd(&d.counters[RG_CNT_CLICKS], static_cast<unsigned long long>(c_clicks));
            if (c_ph) atomicAdd(&d.counters[RG_CNT_PHANTOM], static_cast<unsigned long long>(c_ph));
            if (c_pick) atomicAdd(&d.counters[RG_CNT_EXACT_DRAWS], static_cast<unsigned long long>(c_pick));
            if (c_sweeps) atomicAdd(&d.counters[RG_CNT_EXACT_SWEEPS], static_cast<unsigned long long>(c_sweeps));
            if (c_hit) atomicAdd(&d.counters[kCntWalkHits], static_cast<unsigned long long>(c_hit));
            if (c_anch) atomicAdd(&d.counters[RG_CNT_ANCHORED], static_cast<unsigned long long>(c_anch));
            if (c_maxt) atomicMax(&d.counters[kCntTailMaxT], static_cast<unsigned long long>(c_maxt));
            if (c_limit) atomicAdd(&d.counters[kCntTailLimit], static_cast<unsigned long long>(c_limit));
        }
    }
}


template <int KH, int HIST>
__global__ void __launch_bounds__(kBlock) k_walk_solo(DevSim d_arg, uint32_t n_work, uint32_t chunk_rows, uint32_t in_base) {
    (void)d_arg;       // read where it lies, in the kernel-argument segment (the float64 pick is a call that takes its address)
    const DevSim& d = *(const DevSim*)(const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int K2 = 2 * KH;
    // (the draw's chunk recomputed as a float64 dot; -DRG_WALK_PRECISE_CHUNK=0: in fp32 with its own budget delta_c — a tie: rg_common.hpp)
    constexpr bool kPreciseChunk = RG_WALK_PRECISE_CHUNK && KH <= 10;
    constexpr double kLog2e64 = 1.4426950408889634074;
    constexpr int KC = ((K2 + 3) / 4) * 4;
    constexpr int kEmpty = 3, kPhantom = 4;
    __shared__ hent_t s_hist[kBlock / 64][HIST ? kSoloHist : 1];      // the user's history row: [0] header, then the entries
    __shared__ float s_mbox[kBlock / 64][64 * 3];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    hent_t* hs = s_hist[wave];
    float* mboxf = s_mbox[wave];
    const uint32_t n_cc = d.PT / 64;
    uint64_t row_next = 0, row_end = 0;
    uint32_t c_org = 0, c_ban = 0, c_clicks = 0, c_ph = 0, c_pick = 0, c_maxt = 0, c_limit = 0, c_hit = 0;
    if (d.q_count) n_work = static_cast<uint32_t>(*d.q_count);
    for (;;) {
        uint32_t idx = 0;
        if (lane == 0) idx = static_cast<uint32_t>(atomicAdd(d.q_ticket, 1ull));
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= n_work) break;
        const uint32_t slot = __builtin_amdgcn_readfirstlane(d.park_list[in_base + idx]);
        if (slot == 0xFFFFFFFFu) continue;
        const uint32_t pt = __builtin_amdgcn_readfirstlane(d.park_t[slot]);
        uint32_t t = pt & 0xFFFFFFu;
        int st = static_cast<int>((pt >> 24) & 7u);
        bool pend = (pt >> 27) & 1u;
        const uint32_t user = static_cast<uint32_t>(d.first_user + slot);
        const bool organic_only = (d.first_user + slot) < d.organic_only_below;
        float om[KC];
        {
            const float4* rp = reinterpret_cast<const float4*>(d.cache_row + static_cast<size_t>(slot) * d.cache_row_f);
#pragma unroll
            for (int k4 = 0; k4 < K2 / 4; ++k4) {
                const float4 x = rp[11 + k4];
                om[4 * k4] = x.x; om[4 * k4 + 1] = x.y; om[4 * k4 + 2] = x.z; om[4 * k4 + 3] = x.w;
            }
#pragma unroll
            for (int k = (K2 / 4) * 4; k < K2; ++k) om[k] = reinterpret_cast<const float*>(rp)[44 + k];
#pragma unroll
            for (int k = K2; k < KC; ++k) om[k] = 0.0f;
        }
        hent_t* hr = HIST ? hist_row(d, slot) : nullptr;
        if (HIST) {
            const uint32_t nd0 = h_cnt(hr[0]);
            for (uint32_t j = lane; j <= nd0; j += 64) hs[j] = hr[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t lastv = d.lpv ? d.lpv[slot] : 0u;
        // hot row of the user (no memo entries are added here: the lanes of a run would race for the row)
        const float4* hp = reinterpret_cast<const float4*>(d.walk_hot + static_cast<size_t>(slot) * 32);
        while (st != kEmpty) {
            const uint32_t te = t + static_cast<uint32_t>(lane);             // this lane's event
            const rg_u32x4 w = rg_draw(d.seed, user, te, 0, RG_DRAW_EVENT);
            const double u_trans = rg_uniform(w.w[2], w.w[3]);
            if (st == kPhantom) {
                // final step_offline(done = True): one more act, reward 0 (abstract.py:223-233,311-316) — lane 0's
                double ps = 1.0;
                uint32_t a = 0;
                if (HIST) {
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, t, 0, RG_DRAW_POLICY);
                    a = solo_ouc_act(d, hs, rg_uniform(pw.w[2], pw.w[3]), &ps);
                } else if (d.policy == RG_POLICY_LAST_VIEW_TABLE) {
                    ps = d.pol_ps ? static_cast<double>(d.pol_ps[lastv]) : 1.0;
                    a = static_cast<uint32_t>(d.pol_table[lastv]);
                } else {
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, t, 0, RG_DRAW_POLICY);
                    ps = 1.0 / static_cast<double>(d.P);
                    a = rg_bounded(pw.w[0], pw.w[1], d.P);
                }
                if (lane == 0) {
                    rg_event e;
                    e.u = user; e.t = t; e.code = RG_EV_BANDIT | RG_EV_PHANTOM | a;
                    e.ps = static_cast<float>(ps);
                    d.phantom[slot] = e;
                    d.phantom_ps[slot] = ps;
                    d.has_phantom[slot] = 1;
                }
                c_ph += 1;
                st = kEmpty;
                break;
            }
            const bool org = st == RG_STATE_ORGANIC;
            // ---- bandit run: act and click of every lane's event (they decide where the run ends) ----
            double ps = 1.0, ctr = 0.0;
            uint32_t a = 0;
            bool click = false;
            if (!org) {
                if (HIST) {
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, te, 0, RG_DRAW_POLICY);
                    a = solo_ouc_act(d, hs, rg_uniform(pw.w[2], pw.w[3]), &ps);
                } else if (d.policy == RG_POLICY_LAST_VIEW_TABLE) {
                    ps = d.pol_ps ? static_cast<double>(d.pol_ps[lastv]) : 1.0;
                    a = static_cast<uint32_t>(d.pol_table[lastv]);
                } else {
                    const rg_u32x4 pw = rg_draw(d.policy_seed, user, te, 0, RG_DRAW_POLICY);
                    ps = 1.0 / static_cast<double>(d.P);
                    a = rg_bounded(pw.w[0], pw.w[1], d.P);
                }
                int dec = -1;
                if (!d.aux_pclick) {
                    if (rg_uniform(w.w[0], w.w[1]) < kNoClickBelow) dec = 0;
                    else dec = click_decide32<KC>(d.beta32 + static_cast<size_t>(a) * d.KB4, [&](int k) { return om[k]; }, d.K, d.KB4,
                                                  static_cast<float>(d.mu_b[a]), rg_uniform(w.w[0], w.w[1]));
                }
                if (dec >= 0) click = dec != 0;
                else {
                    const double* b = d.beta + static_cast<size_t>(a) * d.K;
                    const double* omd = d.omega + static_cast<size_t>(slot) * d.OMS;
                    double x = 0.0;
                    for (uint32_t k = 0; k < d.K; ++k) x += b[k] * omd[k];
                    ctr = ff64(x + d.mu_b[a]);
                    const double p0 = 1.0 - ctr;
                    click = (p0 / (p0 + ctr)) <= rg_uniform(w.w[0], w.w[1]);
                }
            }
            // ---- the state after every lane's event, had the run reached it (abstract.py:123-197) ----
            const double c0 = org ? d.cdf_o0 : d.cdf_b0, c1 = org ? d.cdf_o1 : d.cdf_b1;
            int ns = (c0 <= u_trans) + (c1 <= u_trans);
            if (click) ns = RG_STATE_ORGANIC;
            bool limit = false;
            if (organic_only && ns != RG_STATE_ORGANIC) ns = RG_STATE_STOP;
            else if (ns == RG_STATE_STOP) ns = kPhantom;
            else if (te + 2 >= kMaxSteps) { ns = RG_STATE_STOP; limit = true; }
            const unsigned long long leave = __ballot(ns != st);
            const int last = leave ? __builtin_ctzll(leave) : 63;            // the run's events of this pass: lanes 0 .. last
            const bool mine = lane <= last;
            const int ns_last = __shfl(ns, last);
            // ---- organic run: the product of every event of the run ----
            uint32_t v = 0;
            if (org) {
                const float4 h0 = hp[0];
                const double S = static_cast<double>(h0.x), delta = static_cast<double>(h0.y);
                const float Q = h0.z;
                const uint32_t n_hot = __builtin_bit_cast(uint32_t, h0.w);
                double rho_rel, delta_c;
                hot_budgets(hp[7].w, delta, &rho_rel, &delta_c);
                if (kPreciseChunk) delta_c = delta;
                const double u_org = d.u_override ? d.u_override[slot] : rg_uniform(w.w[0], w.w[1]);   // (test hook)
                float uf = static_cast<float>(u_org), u_dn = uf, u_up = uf;
                if (static_cast<double>(uf) > u_org) u_dn = f32_down(uf);
                if (static_cast<double>(uf) < u_org) u_up = f32_up(uf);
                bool hit = false;
                {
                    float e[28];
#pragma unroll
                    for (int i = 1; i < 8; ++i) {
                        const float4 x = hp[i];
                        e[4 * i - 4] = x.x; e[4 * i - 3] = x.y; e[4 * i - 2] = x.z; e[4 * i - 1] = x.w;
                    }
#pragma unroll
                    for (int j = 0; j < kHotEntries; ++j) {
                        const bool in = static_cast<uint32_t>(j) < n_hot && e[3 * j + 1] < u_dn && u_up < e[3 * j + 2];
                        hit = hit || in;
                        v = in ? __builtin_bit_cast(uint32_t, e[3 * j]) : v;
                    }
                }
                const bool first_pend = pend && lane == 0;                   // the parked draw: float64, whatever the memo says
                hit = hit && !first_pend;
                c_hit += static_cast<uint32_t>(__popcll(__ballot(mine && hit)));
                const bool search = mine && !hit && !first_pend;
                bool ok = false;
                if (__ballot(search)) {
                    const double tau = u_org * S;
                    const float tauf = static_cast<float>(tau);
                    uint32_t sc_star = 0;
                    float pbf = 0.0f;
                    {
                        const float4* sp = reinterpret_cast<const float4*>(d.walk_scp + static_cast<size_t>(slot) * kMaxSC);
#pragma unroll
                        for (int i = 0; i < kMaxSC / 4; ++i) {
                            const float4 x = sp[i];
                            const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (xs[q] <= tauf) { sc_star += 1; pbf = fmaxf(pbf, xs[q]); }
                        }
                    }
                    bool found = sc_star < d.n_sc;
                    sc_star = min(sc_star, d.n_sc - 1);
                    uint32_t c_star;
                    {
                        const uint32_t cc0 = sc_star * d.sc_chunks, cc1 = min(cc0 + d.sc_chunks, d.n_chunks);
                        const float* cp = d.cache_chunk + static_cast<size_t>(slot) * d.n_chunks;
                        uint32_t cnt = 0;
                        for (uint32_t cb = cc0; cb < cc1; cb += 16) {
                            float4 w4[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                w4[i] = cb + 4 * i < cc1 ? *reinterpret_cast<const float4*>(cp + cb + 4 * i) : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float xs[4] = {w4[i].x, w4[i].y, w4[i].z, w4[i].w};
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    if (xs[q] <= tauf) { cnt += 1; pbf = fmaxf(pbf, xs[q]); }
                            }
                        }
                        found = found && cnt < cc1 - cc0;
                        c_star = min(cc0 + cnt, cc1 - 1);
                    }
                    const double pb = static_cast<double>(pbf);
                    const float rem = static_cast<float>(tau - pb);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    {
                        const int grp = lane >> 3, gl = lane & 7;
                        unsigned long long todo = __ballot(search);
                        while (todo) {
                            int src = -1;
#pragma unroll
                            for (int g = 0; g < 8; ++g) {
                                const int bit = todo ? __builtin_ctzll(todo) : -1;
                                if (g == grp) src = bit;
                                if (todo) todo &= todo - 1;
                            }
                            const bool has = src >= 0;
                            const int s2 = has ? src : 0;
                            const uint32_t cs = static_cast<uint32_t>(__shfl(static_cast<int>(c_star), s2));
                            const float rems = __shfl(rem, s2);
                            const float4* gp = reinterpret_cast<const float4*>(d.gamma32t + (static_cast<size_t>(cs) * K2) * 32) + gl;
                            const float4 l = *(reinterpret_cast<const float4*>(d.mu32 + cs * 32) + gl);
                            float e0, e1, e2, e3;
                            if constexpr (kPreciseChunk) {        // (as k_walk2's chunk_pass: float64 dot, one rounding of the exp2 argument)
                                double lx = static_cast<double>(l.x), ly = static_cast<double>(l.y), lz = static_cast<double>(l.z), lw = static_cast<double>(l.w);
#pragma unroll
                                for (int kh = 0; kh < K2; kh += KH) {
                                    float4 gk[KH];
#pragma unroll
                                    for (int k = 0; k < KH; ++k) gk[k] = gp[(kh + k) * 8];
#pragma unroll
                                    for (int k = 0; k < KH; ++k) {
                                        const double wk = static_cast<double>(om[kh + k]);   // (every lane holds THE user's omega32)
                                        lx = fma(static_cast<double>(gk[k].x), wk, lx); ly = fma(static_cast<double>(gk[k].y), wk, ly);
                                        lz = fma(static_cast<double>(gk[k].z), wk, lz); lw = fma(static_cast<double>(gk[k].w), wk, lw);
                                    }
                                    asm volatile("" : "+v"(lx), "+v"(ly), "+v"(lz), "+v"(lw));
                                }
                                const double Qd = static_cast<double>(Q);
                                e0 = __builtin_amdgcn_exp2f(static_cast<float>(fma(lx, kLog2e64, -Qd))); e1 = __builtin_amdgcn_exp2f(static_cast<float>(fma(ly, kLog2e64, -Qd)));
                                e2 = __builtin_amdgcn_exp2f(static_cast<float>(fma(lz, kLog2e64, -Qd))); e3 = __builtin_amdgcn_exp2f(static_cast<float>(fma(lw, kLog2e64, -Qd)));
                            } else {
                                float4 lf = l;
#pragma unroll
                                for (int kh = 0; kh < K2; kh += KH) {
                                    float4 gk[KH];
#pragma unroll
                                    for (int k = 0; k < KH; ++k) gk[k] = gp[(kh + k) * 8];
#pragma unroll
                                    for (int k = 0; k < KH; ++k) {
                                        const float wk = om[kh + k];
                                        lf.x = fmaf(gk[k].x, wk, lf.x); lf.y = fmaf(gk[k].y, wk, lf.y);
                                        lf.z = fmaf(gk[k].z, wk, lf.z); lf.w = fmaf(gk[k].w, wk, lf.w);
                                    }
                                    asm volatile("" : "+v"(lf.x), "+v"(lf.y), "+v"(lf.z), "+v"(lf.w));
                                }
                                e0 = __builtin_amdgcn_exp2f(fmaf(lf.x, kLog2e, -Q)); e1 = __builtin_amdgcn_exp2f(fmaf(lf.y, kLog2e, -Q));
                                e2 = __builtin_amdgcn_exp2f(fmaf(lf.z, kLog2e, -Q)); e3 = __builtin_amdgcn_exp2f(fmaf(lf.w, kLog2e, -Q));
                            }
                            const float q0 = e0, q1 = q0 + e1, q2 = q1 + e2, q3 = q2 + e3;
                            float inc = q3;
#pragma unroll
                            for (int o2 = 1; o2 < 8; o2 <<= 1) {
                                const float y = __shfl_up(inc, o2, 8);
                                if (gl >= o2) inc += y;
                            }
                            float ex = __shfl_up(inc, 1, 8);
                            if (gl == 0) ex = 0.0f;
                            const float x0 = ex + q0, x1 = ex + q1, x2 = ex + q2, x3 = ex + q3;
                            const int j0 = x0 > rems ? 0 : x1 > rems ? 1 : x2 > rems ? 2 : x3 > rems ? 3 : -1;
                            const unsigned long long hits = __ballot(has && j0 >= 0);
                            const uint32_t gmask = static_cast<uint32_t>(hits >> (8 * grp)) & 0xFFu;
                            if (has) {
                                if (gmask) {
                                    if (gl == __builtin_ctz(gmask)) {
                                        mboxf[src * 3] = static_cast<float>(4 * gl + j0);
                                        mboxf[src * 3 + 1] = j0 == 0 ? ex : j0 == 1 ? x0 : j0 == 2 ? x1 : x2;
                                        mboxf[src * 3 + 2] = j0 == 0 ? x0 : j0 == 1 ? x1 : j0 == 2 ? x2 : x3;
                                    }
                                } else if (gl == 0) { mboxf[src * 3] = -1.0f; mboxf[src * 3 + 1] = 0.0f; mboxf[src * 3 + 2] = 0.0f; }
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    if (search) {
                        const int ix = static_cast<int>(mboxf[lane * 3]);
                        const CertLin ct = cert_correlated(S, pb, static_cast<double>(mboxf[lane * 3 + 1]), static_cast<double>(mboxf[lane * 3 + 2]), delta, rho_rel, delta_c);
                        v = c_star * 32 + static_cast<uint32_t>(max(ix, 0));
                        const bool lo_ok = v == 0 || u_org * ct.den_lo > ct.num_lo;
                        const bool hi_ok = v == d.P - 1 || u_org * ct.den_hi < ct.num_hi;
                        ok = found && ix >= 0 && v < d.P && ct.valid && lo_ok && hi_ok;
                    }
                }
                // uncertified draws (and the parked one): float64 picks from the user's stored sums, one after the other
                unsigned long long picks = __ballot(mine && !hit && !ok);
                c_pick += static_cast<uint32_t>(__popcll(picks));
                while (picks) {
                    const int L = __builtin_ctzll(picks);
                    picks &= picks - 1;
                    const double s_u = __shfl(u_org, L);
                    const double M = static_cast<double>(d.exact_ref[slot]) * 0.69314718055994530942;
                    const uint32_t pv = exact_pick_pfx(d, d.exact_sums + static_cast<size_t>(slot) * n_cc,
                                                       d.omega + static_cast<size_t>(slot) * d.OMS, M, s_u, lane);
                    if (lane == L) v = pv;
                    __builtin_amdgcn_wave_barrier();
                }
                pend = false;
            }
            // ---- rows of the run's events ----
            const uint32_t n_ev = static_cast<uint32_t>(last) + 1u;
            if (row_next + n_ev > row_end) {
                for (uint64_t r = row_next + lane; r < row_end; r += 64)
                    if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&d.counters[kCntTailRows], static_cast<unsigned long long>(chunk_rows));
                base = readfirstlane_u64(base);
                row_next = base; row_end = base + chunk_rows;
            }
            const uint64_t my_row = row_next + static_cast<uint32_t>(lane);
            row_next += n_ev;
            if (mine && d.log && my_row < d.log_cap) {
                rg_event e;
                e.u = user; e.t = te;
                e.code = org ? v : (RG_EV_BANDIT | (click ? RG_EV_CLICK : 0u) | a);
                e.ps = org ? __builtin_nanf("") : static_cast<float>(ps);
                d.log[my_row] = e;
                if (!org && d.aux_ps) d.aux_ps[my_row] = ps;
                if (!org && d.aux_pclick) d.aux_pclick[my_row] = ctr;
            }
            if (org) {
                c_org += n_ev;
                lastv = static_cast<uint32_t>(__shfl(static_cast<int>(v), last));
                if (d.lpv && lane == 0) d.lpv[slot] = lastv;
                if (HIST)
                    for (int i = 0; i <= last; ++i) solo_hist_add(d, hs, hr, static_cast<uint32_t>(__shfl(static_cast<int>(v), i)), lane);
            } else {
                c_ban += n_ev;
                c_clicks += static_cast<uint32_t>(__popcll(__ballot(mine && click)));
            }
            // ---- the user after the run ----
            t += n_ev;
            if (ns_last == RG_STATE_STOP || ns_last == kPhantom) {
                if (lane == 0) d.n_events[slot] = t;
                c_maxt = max(c_maxt, t);
                c_limit += static_cast<uint32_t>(__shfl(static_cast<int>(limit), last));
            }
            st = ns_last == RG_STATE_STOP ? kEmpty : ns_last;
        }
    }
    for (uint64_t r = row_next + lane; r < row_end; r += 64)
        if (d.log && r < d.log_cap) { rg_event e; e.u = 0; e.t = 0; e.code = kHoleCode; e.ps = 0.0f; d.log[r] = e; }
    if (lane == 0) {
        if (c_org) atomicAdd(&d.counters[kCntTailOrganic], static_cast<unsigned long long>(c_org));
        if (c_ban) atomicAdd(&d.counters[kCntTailBandit], static_cast<unsigned long long>(c_ban));
        if (c_clicks) atomicAdd(&d.counters[RG_CNT_CLICKS], static_cast<unsigned long long>(c_clicks));
        if (c_ph) atomicAdd(&d.counters[RG_CNT_PHANTOM], static_cast<unsigned long long>(c_ph));
        if (c_pick) atomicAdd(&d.counters[RG_CNT_EXACT_DRAWS], static_cast<unsigned long long>(c_pick));
        if (c_hit) atomicAdd(&d.counters[kCntWalkHits], static_cast<unsigned long long>(c_hit));
        if (c_maxt) atomicMax(&d.counters[kCntTailMaxT], static_cast<unsigned long long>(c_maxt));
        if (c_limit) atomicAdd(&d.counters[kCntTailLimit], static_cast<unsigned long long>(c_limit));
    }
}
solo_kernel_t solo_kernel_for(const DevSim& d) {
    const bool ouc = d.policy == RG_POLICY_ORGANIC_USER_COUNT;
    if (walk2_kernel_for(d, 3) == nullptr || (ouc && d.hist_cap > kSoloHist)) return nullptr;
#ifdef RG_W2_ONLY
    return k_walk_solo<10, 1>;
#else
    switch (d.KH) {
        case 4: return ouc ? k_walk_solo<4, 1> : k_walk_solo<4, 0>;
        case 10: return ouc ? k_walk_solo<10, 1> : k_walk_solo<10, 0>;
        default: return ouc ? k_walk_solo<16, 1> : k_walk_solo<16, 0>;
    }
#endif
}
walk_kernel_t walk2_kernel_for(const DevSim& d, int occ) {
    // the forms k_walk2 is instantiated for: K <= 32, no group sums, the policies without a view history or the
    // OrganicUserEventCounter default (exploit_explore, epsilon = 0, select_randomly)
    const bool ouc = d.policy == RG_POLICY_ORGANIC_USER_COUNT;
    if (d.KH > 16 || !d.walk_hot) return nullptr;
    if (ouc && !(d.ouc_exploit_explore && d.ouc_epsilon == 0.0 && d.ouc_select_randomly)) return nullptr;
    if (d.policy != RG_POLICY_UNIFORM_ENV && d.policy != RG_POLICY_RANDOM_AGENT && d.policy != RG_POLICY_LAST_VIEW_TABLE && !ouc) return nullptr;
#ifdef RG_W2_ONLY     // kernel work: one instantiation, seconds to compile (never a shipped build)
    return k_walk2<10, 2>;
#else
    (void)occ;
    // the view-history line in LDS: compact (31 products per line) where a product fits 16 bits (RECOGYM_WALK_HIST=1: the
    // 64-bit line of 15 products, A/B)
    const bool compact = ouc && d.P <= 65535u && d.hist_cap >= 32u && d.hist_cap <= 32768u && !d.walk_line64;
    switch (d.KH) {
        case 4: return ouc ? (compact ? k_walk2<4, 2> : k_walk2<4, 1>) : k_walk2<4, 0>;
        case 10: return ouc ? (compact ? k_walk2<10, 2> : k_walk2<10, 1>) : k_walk2<10, 0>;
        default: return ouc ? (compact ? k_walk2<16, 2> : k_walk2<16, 1>) : k_walk2<16, 0>;
    }
#endif
}
void (*cache_prefix_kernel())(DevSim, int) { return k_cache_prefix; }
void (*exact_prefix_kernel())(DevSim, uint32_t) { return k_exact_prefix; }

// blocks per CU the kernel is compiled for (register budget 512 / OCC per lane): KH <= 16 at 2, 3 or 4, KH = 32 at 1
walk_kernel_t walk_kernel_for(const DevSim& d, int occ) {
#ifdef RG_W2_ONLY
    return nullptr;
#else
    // the O(P) forms of the OrganicUserEventCounter policy are compiled in only where the configuration can reach them
    const bool dense = d.policy == RG_POLICY_ORGANIC_USER_COUNT && !(d.ouc_exploit_explore && d.ouc_epsilon == 0.0);
#define RG_W(kh, o) (dense ? k_walk<kh, o, true> : k_walk<kh, o, false>)
    switch (d.KH) {
        // (three blocks per CU at K <= 32 — two and four were measured in round 2: 358 / 328 against 301 ms — one at K <= 64)
        case 4: return RG_W(4, 3);
        case 10: return RG_W(10, 3);
        case 16: return RG_W(16, 3);
        default: return RG_W(32, 1);
    }
#undef RG_W
#endif
}

}  // namespace rgk

#ifdef RG_WALK_TIMING
// (in this unit: g_walk_stat is a per-unit static)
extern "C" void rg_debug_walk_kinds(unsigned long long* out, int clear) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(rgk::g_walk_stat), sizeof(unsigned long long) * 20);
    if (clear) { unsigned long long z[20] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(rgk::g_walk_stat), z, sizeof(z)); }
}
#endif
