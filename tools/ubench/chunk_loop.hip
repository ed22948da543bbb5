// Isolated inner loop of k_draw_bf16: per 32-product chunk 8 ds_read_b128, NM dependent bf16 MFMAs,
// 16 v_exp_f32 + packed tree sum.  Toggles show which part bounds the loop at 1..4 blocks per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

template <int NM, bool LDS, bool EXP, int NACC, bool ILV = false>
__global__ void __launch_bounds__(256) k(float* out, int chunks, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < lds_bytes / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
    __syncthreads();
    bf16x8 B[9];
    for (int s = 0; s < 9; ++s) for (int e = 0; e < 8; ++e) B[s][e] = (short)(0x3c00 + s + e + lane);
    float total = 0.f;
    f32x16 acc[NACC];
    bf16x8 Aall[NACC][4];
    for (int c = 0; c < chunks; c += NACC) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int cc = (c + a) & 3;
            const char* arow = smem + (cc * 32 + j) * 144 + 16 * h;
            const float* mu = reinterpret_cast<const float*>(smem + 20000) + cc * 32 + 4 * h;
            bf16x8 A[4];
            if (LDS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 m = *reinterpret_cast<const float4*>(mu + 8 * q);
                    acc[a][4 * q] = m.x; acc[a][4 * q + 1] = m.y; acc[a][4 * q + 2] = m.z; acc[a][4 * q + 3] = m.w;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) A[s] = *reinterpret_cast<const bf16x8*>(arow + 32 * s);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][r] = 0.01f * r + total * 1e-9f;
#pragma unroll
                for (int s = 0; s < 4; ++s) A[s] = B[s + 1];
            }
            if (!ILV) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m & 3], B[m % 9], acc[a], 0, 0, 0);
            }
            if (ILV) { Aall[a][0] = A[0]; Aall[a][1] = A[1]; Aall[a][2] = A[2]; Aall[a][3] = A[3]; }
        }
        if (ILV) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int a = 0; a < NACC; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aall[a][m & 3], B[m % 9], acc[a], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            f32x16 y = acc[a];
            if (EXP) {
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = __builtin_amdgcn_exp2f(y[r] * 1e-3f);
            }
            f32x2 p0 = {y[0], y[1]}, p1 = {y[2], y[3]}, p2 = {y[4], y[5]}, p3 = {y[6], y[7]};
            const f32x2 p4 = {y[8], y[9]}, p5 = {y[10], y[11]}, p6 = {y[12], y[13]}, p7 = {y[14], y[15]};
            p0 += p4; p1 += p5; p2 += p6; p3 += p7; p0 += p2; p1 += p3; p0 += p1;
            total += p0[0] + p0[1];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

template <int NM, bool LDS, bool EXP, int NACC, bool ILV = false>
void run(const char* name, float* d_out, int bpc) {
    const int chunks = 40000, grid = 256 * bpc, lds = 160 * 1024 / bpc - 2048;   // LDS size pins blocks/CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<NM, LDS, EXP, NACC, ILV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, LDS, EXP, NACC, ILV>), dim3(grid), dim3(256), lds, 0, d_out, 400, 24000);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, LDS, EXP, NACC, ILV>), dim3(grid), dim3(256), lds, 0, d_out, chunks, 24000);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // cycles per chunk per SIMD (each SIMD hosts bpc waves, each doing `chunks`)
    printf("%-44s blocks/CU=%d: %7.1f cycles per chunk per SIMD\n", name, bpc, ms * 1e-3 * 2.4e9 / (double(chunks) * bpc));
}

int main() {
    float* d_out; (void)hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8);
    for (int b : {1, 2, 3, 4}) {
        run<9, true, true, 1>("full: LDS + 9 MFMA + exp (1 acc)", d_out, b);
        run<9, true, true, 2>("full, 2 chunks per iteration (2 acc)", d_out, b);
        run<9, false, true, 1>("no LDS reads", d_out, b);
        run<9, true, false, 1>("no exp", d_out, b);
        run<0, true, true, 1>("no MFMA", d_out, b);
        run<9, false, false, 1>("MFMA + tree only", d_out, b);
        run<9, true, true, 2, true>("full, 2 acc interleaved", d_out, b);
        run<9, true, true, 4, true>("full, 4 acc interleaved", d_out, b);
        run<9, false, false, 2, true>("MFMA + tree only, 2 acc interleaved", d_out, b);
        run<9, false, false, 4, true>("MFMA + tree only, 4 acc interleaved", d_out, b);
    }
    return 0;
}
