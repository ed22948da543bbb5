// How fast do DEPENDENT v_mfma_f32_32x32x16_f16 issue on gfx950?  k_sweep_xh's residual accumulator is a chain of five.
// cycles (s_memtime-free: wall time x 2.4 GHz / MFMAs per wave) per MFMA for 1 / 2 / 4 independent accumulators, at 1 and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain.bin mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = seed + a + r;
    f16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(0.001f * (threadIdx.x + i)); bv[i] = (_Float16)(0.002f * (i + 1)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8 / NACC; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(float* d_out, int blocks_per_cu) {
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC>), dim3(grid), dim3(256), 0, 0, d_out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = ms * 1e-3 * 2.4e9 / (iters * 8.0);
    printf("{\"accumulators\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_at_2p4GHz_per_mfma_of_a_wave\": %.1f, \"simd_cycles_per_mfma\": %.1f}\n",
           NACC, blocks_per_cu, ms, per_wave, per_wave / blocks_per_cu);
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    for (int b = 1; b <= 2; ++b) { run<1>(d, b); run<2>(d, b); run<4>(d, b); }
    return 0;
}
