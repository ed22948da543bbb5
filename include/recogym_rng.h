/*
 * recogym_rng.h — the counter-RNG contract of the vectorised reco-gym-v1 simulator.
 *
 * One header, three consumers: the HIP kernels (recogym_amd/csrc), the C oracle
 * (oracle/recogym_oracle.c) and — restated in numpy — tests/ref_harness.py, which injects the
 * same draws into the UNMODIFIED reference (`env.rng = ...`).
 *
 * Why a counter RNG: the reference draws every user from one sequential MT19937 stream
 * (recogym/envs/abstract.py:59-62), so user n's trajectory depends on how many draws users
 * 0..n-1 consumed.  A simulator that advances millions of users concurrently must address a
 * draw by WHAT it is for, not by WHEN it happens.  Every draw the reference makes
 *  is addressed as
 *
 *     Philox4x32-10( key = seed64, counter = (user_id, t, slot, purpose) )
 *
 *   purpose RG_DRAW_EVENT  (slot 0): words 0,1 -> u_event  (organic product draw,
 *                                    reco_env_v1.py:124, or click draw, reco_env_v1.py:112);
 *                                    words 2,3 -> u_trans  (Markov transition, reco_env_v1.py:87)
 *   purpose RG_DRAW_POLICY (slot 0): words 0,1 -> first policy draw (uniform action /
 *                                    explore flag); words 2,3 -> second policy draw (action icdf)
 *                                    (abstract.py:214, random_agent.py:26,
 *                                    organic_user_count.py:48,66)
 *   purpose RG_DRAW_DRIFT  (slot j): Box-Muller pair -> z[2j], z[2j+1] of the omega drift that
 *                                    follows event t (reco_env_v1.py:96)
 *   purpose RG_DRAW_RESET  (slot j, t = 0): z[2j], z[2j+1] of omega_0 (reco_env_v1.py:80)
 *   purpose RG_DRAW_TIME   (slot 0): z0 -> the time increment after event t (NormalTimeGenerator only)
 *
 * seed64 is `random_seed + epoch` (abstract.py:62) for env draws and the agent's own
 * `random_seed` for RG_DRAW_POLICY draws of an agent (with agent=None the policy draw comes
 * from the env stream, abstract.py:214, so it uses the env seed).
 *
 * Draw -> sample maps (all exact integer / IEEE-double operations, identical everywhere):
 *   uniform double : ((a >> 5) * 2^26 + (b >> 6)) / 2^53      — the same 53-bit construction
 *                    numpy's RandomState.random_sample uses, so u in [0, 1)
 *   bounded integer: (m53 * n) >> 53, m53 the 53-bit integer above — floor(u * n) without a
 *                    floating-point multiply (unbiased to 2^-53 * n)
 *   standard normal: Box-Muller in double: r = sqrt(-2 log(1 - u1)), z0 = r cos(2 pi u2),
 *                    z1 = r sin(2 pi u2)
 */
#ifndef RECOGYM_RNG_H
#define RECOGYM_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RG_HD __host__ __device__ __forceinline__
#else
#define RG_HD static inline
#endif

#define RG_DRAW_EVENT 0u
#define RG_DRAW_POLICY 1u
#define RG_DRAW_DRIFT 2u
#define RG_DRAW_RESET 3u
#define RG_DRAW_TIME 4u    /* NormalTimeGenerator (normal_time_generator.py:23-26): slot 0, z0 of the Box-Muller pair -> the
                              increment |mu + sigma z| that follows event t */

#define RG_TWO_PI 6.283185307179586476925286766559

typedef struct rg_u32x4 {
    uint32_t w[4];
} rg_u32x4;

/* Philox4x32-10, Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3"
 * (SC'11).  Known-answer vectors are checked in tests/test_rng.py. */
RG_HD rg_u32x4 rg_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    rg_u32x4 out;
    out.w[0] = c0; out.w[1] = c1; out.w[2] = c2; out.w[3] = c3;
    return out;
}

/* One addressed draw block: 4 x 32 random bits for (seed, user, t, slot, purpose). */
RG_HD rg_u32x4 rg_draw(uint64_t seed, uint32_t user, uint32_t t, uint32_t slot,
                       uint32_t purpose) {
    return rg_philox4x32_10(user, t, slot, purpose, (uint32_t)seed, (uint32_t)(seed >> 32));
}

/* 53-bit integer from two words, numpy random_sample style. */
RG_HD uint64_t rg_m53(uint32_t a, uint32_t b) {
    return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
}

RG_HD double rg_uniform(uint32_t a, uint32_t b) {
    return (double)rg_m53(a, b) * (1.0 / 9007199254740992.0);
}

/* floor(u * n) computed exactly: (m53 * n) >> 53 == hi64((m53 << 11) * n). */
RG_HD uint32_t rg_bounded(uint32_t a, uint32_t b, uint32_t n) {
    const uint64_t x = rg_m53(a, b) << 11;
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__umul64hi(x, (uint64_t)n);
#else
    return (uint32_t)(((unsigned __int128)x * (unsigned __int128)n) >> 64);
#endif
}

#endif /* RECOGYM_RNG_H */
