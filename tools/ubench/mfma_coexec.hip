// Does VALU work co-execute with MFMA on one SIMD?  fp32-input MFMA vs bf16 MFMA, gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_coexec mfma_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

template <int MODE, int NVALU>   // MODE 0: no mfma, 1: f32 mfma 32x32x2, 2: bf16 mfma 32x32x16
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = seed + a + r;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed * (i + 1) + threadIdx.x;
    float av = seed + threadIdx.x, bv = seed * 0.5f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(threadIdx.x + i); bb[i] = (short)(i * 3 + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (MODE == 1) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            if (MODE == 2) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[a], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NVALU; ++v) {
                const int i = (a * NVALU + v) & 7;
                x[i] = __builtin_amdgcn_exp2f(fmaf(x[i], 0.999f, -0.25f));
            }
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NVALU>
void run(const char* name, float* d_out, int blocks_per_cu) {
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NVALU>), dim3(grid), dim3(256), 0, 0, d_out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NVALU>), dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles per (mfma + NVALU pairs) per wave-slot at 2.4 GHz, per SIMD holding blocks_per_cu waves
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0) / blocks_per_cu;
    printf("%-34s blocks/CU=%d  %.3f ms  -> %.1f cycles per [1 mfma + %d (fma+exp)] per wave per SIMD-share\n",
           name, blocks_per_cu, ms, cyc, NVALU);
}

int main() {
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8);
    for (int b : {1, 2}) {
        run<1, 0>("f32 mfma only", d_out, b);
        run<0, 4>("valu only (4 fma+exp)", d_out, b);
        run<1, 4>("f32 mfma + 4 (fma+exp)", d_out, b);
        run<0, 8>("valu only (8 fma+exp)", d_out, b);
        run<1, 8>("f32 mfma + 8 (fma+exp)", d_out, b);
        run<2, 0>("bf16 mfma only", d_out, b);
        run<2, 4>("bf16 mfma + 4 (fma+exp)", d_out, b);
        run<2, 8>("bf16 mfma + 8 (fma+exp)", d_out, b);
    }
    return 0;
}
