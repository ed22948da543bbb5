// Checks (1) the operand layout of v_mfma_f32_32x32x16_bf16 and (2) the accuracy of the 3-way
// bf16 split product sum_{i+j<=4} G_i w_j against float64, for K = 20 and 64.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

__host__ __device__ inline unsigned short bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__host__ __device__ inline float bf16_to_f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__host__ __device__ inline void split3(float x, unsigned short* o) {
    o[0] = bf16_rne(x); float r = x - bf16_to_f(o[0]);
    o[1] = bf16_rne(r); r = r - bf16_to_f(o[1]);
    o[2] = bf16_rne(r);
}

// A: [32 rows][KE] bf16 row-major, B: [32 users][KE] bf16, C0: [32 rows] -> D[row][user]
__global__ void k_mfma(const unsigned short* A, const unsigned short* B, const float* c0, float* D, int KE) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c0[(r & 3) + 8 * (r >> 2) + 4 * h];
    for (int s = 0; s < KE / 16; ++s) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = (short)A[i * KE + 16 * s + 8 * h + e];
            b[e] = (short)B[i * KE + 16 * s + 8 * h + e];   // row i of B = user j = lane & 31
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

int main() {
    std::mt19937_64 gen(1);
    std::normal_distribution<double> nd(0.0, 1.0);
    for (int K : {20, 64}) {
        // effective K: segments [G1|G2|G3] x3 B variants, as three MFMA groups:
        //   group 1: A=[G1|G2|G3] B=[w1|w1|w1]; group 2: A=[G1|G2] B=[w2|w2]; group 3: A=[G1] B=[w3]
        const int seg = K, KE1 = ((3 * seg + 15) / 16) * 16;
        std::vector<float> G(32 * K), W(32 * K), mu(32);
        for (auto& x : G) x = (float)nd(gen);
        for (auto& x : W) x = (float)(nd(gen) * 1.5);
        for (auto& x : mu) x = (float)(3.0 * nd(gen));
        // concatenated operand with all 6 terms along K: total KE = KE1 + KE2 + KE3
        const int KE2 = ((2 * seg + 15) / 16) * 16, KE3 = ((seg + 15) / 16) * 16, KE = KE1 + KE2 + KE3;
        std::vector<unsigned short> A(32 * KE, 0), B(32 * KE, 0);
        for (int r = 0; r < 32; ++r)
            for (int k = 0; k < K; ++k) {
                unsigned short g[3], w[3];
                split3(G[r * K + k], g); split3(W[r * K + k], w);
                // group 1
                A[r * KE + k] = g[0]; A[r * KE + seg + k] = g[1]; A[r * KE + 2 * seg + k] = g[2];
                B[r * KE + k] = w[0]; B[r * KE + seg + k] = w[0]; B[r * KE + 2 * seg + k] = w[0];
                // group 2
                A[r * KE + KE1 + k] = g[0]; A[r * KE + KE1 + seg + k] = g[1];
                B[r * KE + KE1 + k] = w[1]; B[r * KE + KE1 + seg + k] = w[1];
                // group 3
                A[r * KE + KE1 + KE2 + k] = g[0];
                B[r * KE + KE1 + KE2 + k] = w[2];
            }
        unsigned short *dA, *dB; float *dC, *dD;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 128); hipMalloc(&dD, 4096);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, mu.data(), 128, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, KE);
        std::vector<float> D(1024);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        double max_rel = 0, max_rel_f32 = 0, sum_rel = 0;
        for (int p = 0; p < 32; ++p)
            for (int u = 0; u < 32; ++u) {
                double ref = mu[p], absum = std::fabs((double)mu[p]);
                float f = mu[p];
                for (int k = 0; k < K; ++k) {
                    ref += (double)G[p * K + k] * (double)W[u * K + k];
                    absum += std::fabs((double)G[p * K + k] * (double)W[u * K + k]);
                    f = std::fmaf(G[p * K + k], W[u * K + k], f);
                }
                const double e = std::fabs(D[p * 32 + u] - ref) / absum;
                const double e32 = std::fabs((double)f - ref) / absum;
                max_rel = std::fmax(max_rel, e); max_rel_f32 = std::fmax(max_rel_f32, e32); sum_rel += e;
            }
        printf("K=%d KE=%d: bf16x3 split max |err|/sum|terms| = %.3e (%.2f x 2^-24), mean %.3e; plain fp32 fma chain max %.3e (%.2f x 2^-24)\n",
               K, KE, max_rel, max_rel / 5.96e-8, sum_rel / 1024, max_rel_f32, max_rel_f32 / 5.96e-8);
    }
    return 0;
}
