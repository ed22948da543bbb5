// What k_sweep_xh (recogym_amd/csrc/rg_draw_exacthi.hip) relies on, probed on the device: v_mfma_f32_32x32x16_f16 with
// FIXED-POINT fp16 operands (every product a multiple of one quantum, every partial sum below 2^24 quanta) returns the EXACT
// sum, in chains of MFMAs, whatever the signs and magnitudes — and, for the record, how the unit treats fp16 subnormals and
// how it rounds sums that are NOT representable (neither property is relied on: the kernel keeps every piece normal and
// budgets one rounding per added term of the residual accumulator).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_exact.bin mfma_f16_exact.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// A [32][16 n] row-major halfs, B [16 n][32] as bt [32 columns][16 n] halfs, c0 [32][32] seeds: D = c0 + A B over n chained MFMAs
__global__ void k_chain(const _Float16* a, const _Float16* bt, const float* c0, float* out, int n) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c0[((r / 4) * 8 + 4 * h + (r % 4)) * 32 + j];
    for (int m = 0; m < n; ++m) {
        f16x8 av, bv;
        for (int e = 0; e < 8; ++e) { av[e] = a[j * 16 * n + 16 * m + 8 * h + e]; bv[e] = bt[j * 16 * n + 16 * m + 8 * h + e]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out[((r / 4) * 8 + 4 * h + (r % 4)) * 32 + j] = acc[r];
}

int main() {
    const int n = 2, KS = 16 * n;
    std::vector<_Float16> a(32 * KS), bt(32 * KS);
    std::vector<float> c0(32 * 32), out(32 * 32);
    _Float16 *da, *db; float *dc, *dout;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, bt.size() * 2); hipMalloc(&dc, c0.size() * 4); hipMalloc(&dout, out.size() * 4);
    auto run = [&]() {
        hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, bt.data(), bt.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dc, c0.data(), c0.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, da, db, dc, dout, n);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    };
    std::mt19937_64 rng(12345);
    // ---- 1. exactness on the fixed-point grid: A ints |.| <= 2047 x 2^-8, B ints x 2^-8 scaled so that sum |a b| < 2^24 quanta ----
    long long bad = 0, total = 0; double worst_fill = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        const int kk = 8 + static_cast<int>(rng() % 24);           // live slots
        const int amax = 1 + static_cast<int>(rng() % 2047);
        // B magnitudes: sum_k |a|max |b_k| must stay below 2^24 quanta
        const long long budget = (1ll << 24) - 1;
        const int bmax = static_cast<int>(std::min<long long>(2047, budget / (static_cast<long long>(amax) * kk)));
        std::vector<long long> ai(32 * KS, 0), bi(32 * KS, 0);
        for (int r = 0; r < 32; ++r)
            for (int s = 0; s < KS; ++s) {
                long long x = s < kk ? static_cast<long long>(rng() % (2 * amax + 1)) - amax : 0;
                long long y = s < kk && bmax > 0 ? static_cast<long long>(rng() % (2 * bmax + 1)) - bmax : 0;
                if (trial % 3 == 0 && s < kk) { x = (rng() & 1) ? amax : -amax; y = (rng() & 1) ? bmax : -bmax; }   // extremes: cancellation
                ai[r * KS + s] = x; bi[r * KS + s] = y;
                a[r * KS + s] = static_cast<_Float16>(static_cast<float>(x) * 0.00390625f);
                bt[r * KS + s] = static_cast<_Float16>(static_cast<float>(y) * 0.00390625f);
            }
        for (auto& c : c0) c = 0.0f;
        run();
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                long long sum = 0, asum = 0;
                for (int s = 0; s < KS; ++s) { sum += ai[i * KS + s] * bi[j * KS + s]; asum += std::llabs(ai[i * KS + s] * bi[j * KS + s]); }
                const double want = static_cast<double>(sum) / 65536.0;
                worst_fill = std::max(worst_fill, static_cast<double>(asum) / 16777216.0);
                ++total;
                if (static_cast<double>(out[i * 32 + j]) != want) ++bad;
            }
    }
    printf("{\"probe\": \"fixed_point_exact\", \"outputs\": %lld, \"mismatches\": %lld, \"largest_sum_abs_over_2p24\": %.4f}\n", total, bad, worst_fill);
    // ---- 2. fp16 subnormal operand: 2^-20 x 2^10 ----
    for (auto& x : a) x = static_cast<_Float16>(0.0f);
    for (auto& x : bt) x = static_cast<_Float16>(0.0f);
    for (auto& c : c0) c = 0.0f;
    a[0] = static_cast<_Float16>(9.5367431640625e-07f); bt[0] = static_cast<_Float16>(1024.0f);
    run();
    printf("{\"probe\": \"fp16_subnormal_operand\", \"got\": %.10g, \"want_if_honoured\": %.10g}\n", out[0], 9.5367431640625e-07 * 1024.0);
    // ---- 3. how a non-representable sum is rounded: C = 2^24, sixteen products of 0.5 (exact sum 2^24 + 8) ----
    for (int s = 0; s < 16; ++s) { a[s] = static_cast<_Float16>(0.5f); bt[s] = static_cast<_Float16>(1.0f); }
    c0[0] = 16777216.0f;
    run();
    printf("{\"probe\": \"sixteen_halves_onto_2p24\", \"got_minus_2p24\": %.1f, \"single_rounding_would_give\": 8.0, \"per_add_rounding_would_give\": 0.0}\n",
           static_cast<double>(out[0]) - 16777216.0);
    // ---- 4. ... and whether the products of ONE instruction are summed before they meet C: C = 2^24, products +1 and 15 x 0.5 ----
    a[0] = static_cast<_Float16>(1.0f);
    run();
    printf("{\"probe\": \"one_and_fifteen_halves_onto_2p24\", \"got_minus_2p24\": %.1f, \"exact\": 8.5}\n", static_cast<double>(out[0]) - 16777216.0);
    return bad != 0;
}
