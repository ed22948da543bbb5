// Software-pipelined form of the chunk loop: MFMAs of pair n+1 (two interleaved accumulator chains)
// issued between the exp-sum VALU work of pair n, interleave enforced with sched_group_barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

__device__ __forceinline__ float expsum(f32x16 y) {
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = __builtin_amdgcn_exp2f(y[r]);
    f32x2 p0 = {y[0], y[1]}, p1 = {y[2], y[3]}, p2 = {y[4], y[5]}, p3 = {y[6], y[7]};
    const f32x2 p4 = {y[8], y[9]}, p5 = {y[10], y[11]}, p6 = {y[12], y[13]}, p7 = {y[14], y[15]};
    p0 += p4; p1 += p5; p2 += p6; p3 += p7; p0 += p2; p1 += p3; p0 += p1;
    return p0[0] + p0[1];
}

template <int VPM, bool SCHED>
__global__ void __launch_bounds__(256) k(float* out, int pairs, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < lds_bytes / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 1e-6f * i;
    __syncthreads();
    bf16x8 B[9];
    for (int s = 0; s < 9; ++s) for (int e = 0; e < 8; ++e) B[s][e] = (short)(0x3000 + s + e + lane);
    float total = 0.f;
    f32x16 p0, p1;            // previous pair (being exp-summed)
    for (int r = 0; r < 16; ++r) { p0[r] = 0.f; p1[r] = 0.f; }
    for (int c = 0; c < pairs; ++c) {
        const int cc = (c & 1) * 2;
        const char* arow0 = smem + (cc * 32 + j) * 144 + 16 * h;
        const char* arow1 = arow0 + 32 * 144;
        const float* mu = reinterpret_cast<const float*>(smem + 20000) + cc * 32 + 4 * h;
        f32x16 a0, a1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 m0 = *reinterpret_cast<const float4*>(mu + 8 * q);
            const float4 m1 = *reinterpret_cast<const float4*>(mu + 32 + 8 * q);
            a0[4 * q] = m0.x; a0[4 * q + 1] = m0.y; a0[4 * q + 2] = m0.z; a0[4 * q + 3] = m0.w;
            a1[4 * q] = m1.x; a1[4 * q + 1] = m1.y; a1[4 * q + 2] = m1.z; a1[4 * q + 3] = m1.w;
        }
        bf16x8 A0[4], A1[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            A0[s] = *reinterpret_cast<const bf16x8*>(arow0 + 32 * s);
            A1[s] = *reinterpret_cast<const bf16x8*>(arow1 + 32 * s);
        }
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[m & 3], B[m], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[m & 3], B[m], a1, 0, 0, 0);
        }
        total += expsum(p0) + expsum(p1);
        if (SCHED) {
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            }
        }
        p0 = a0; p1 = a1;
    }
    total += expsum(p0) + expsum(p1);
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

template <int VPM, bool SCHED>
void run(const char* name, float* d_out, int bpc) {
    const int pairs = 20000, grid = 256 * bpc, lds = 160 * 1024 / bpc - 2048;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<VPM, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<VPM, SCHED>), dim3(grid), dim3(256), lds, 0, d_out, 200, 24000);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<VPM, SCHED>), dim3(grid), dim3(256), lds, 0, d_out, pairs, 24000);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s blocks/CU=%d: %7.1f cycles per chunk per SIMD\n", name, bpc, ms * 1e-3 * 2.4e9 / (2.0 * pairs * bpc));
}

int main() {
    float* d_out; (void)hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8);
    for (int b : {1, 2, 3}) {
        run<0, false>("pipelined pair, compiler schedule", d_out, b);
        run<3, true>("pipelined pair, 1 MFMA : 3 VALU", d_out, b);
        run<4, true>("pipelined pair, 1 MFMA : 4 VALU", d_out, b);
        run<5, true>("pipelined pair, 1 MFMA : 5 VALU", d_out, b);
    }
    return 0;
}
