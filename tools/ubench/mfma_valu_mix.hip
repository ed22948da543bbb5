// k_sweep_xh's instruction mix per chunk — 7 v_mfma_f32_32x32x16_f16 (two accumulators), 16 v_exp_f32, 24 other vector
// instructions — in isolation: does the matrix pipe run beside the vector ALU inside ONE wave, and across 2 / 3 waves per SIMD?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix.bin mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MF, int VA>   // MF: MFMAs on, VA: vector work on
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    f32x16 H, L;
    for (int r = 0; r < 16; ++r) { H[r] = seed + r; L[r] = seed - r; }
    f16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(0.001f * (threadIdx.x + i)); bv[i] = (_Float16)(0.002f * (i + 1)); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed * 0.01f * (i + 1) + threadIdx.x * 1e-4f;
    float s0 = 0, s1 = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            if (MF) { if (m & 1) H = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, H, 0, 0, 0); else L = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, L, 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
            if (VA) {
#pragma unroll
                for (int e = (m * 16) / 7; e < ((m + 1) * 16) / 7; ++e) {
                    x[e] = __builtin_amdgcn_exp2f(fmaf(x[e], 0.999f, -0.25f));     // independent of the MFMAs
                    s0 += x[e]; s1 = fmaf(s1, 0.5f, x[e]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = s0 + s1;
    for (int r = 0; r < 16; ++r) s += H[r] + L[r];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MF, int VA>
void run(const char* name, float* d_out, int blocks_per_cu) {
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MF, VA>), dim3(grid), dim3(256), 0, 0, d_out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MF, VA>), dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("{\"mix\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"simd_cycles_at_2p4GHz_per_chunk_of_a_wave\": %.1f}\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / iters / blocks_per_cu);
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    for (int b = 1; b <= 3; ++b) {
        run<1, 0>("7 mfma", d, b);
        run<0, 1>("16 exp + 48 fma/add", d, b);
        run<1, 1>("both", d, b);
    }
    return 0;
}
