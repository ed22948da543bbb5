// Does float64 VALU work co-execute with the float64 MFMA (v_mfma_f64_16x16x4_f64) on one SIMD of gfx950?
// k_exact_sums_h / _m put the dot products on the matrix pipe and the exps on the vector ALU and count on the two
// running side by side; the fp32-input MFMA did not (mfma_coexec.hip).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_coexec.bin mfma_f64_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int MFMA, int NVALU, int F32>   // per step: MFMA f64 MFMAs (4 independent accumulators) + NVALU v_fma_f64 (or f32) per accumulator
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed) {
    f64x4 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = seed + a + r;
    double x[8];
    float y[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed * (i + 1) + threadIdx.x; y[i] = static_cast<float>(x[i]); }
    const double av = seed + threadIdx.x, bv = seed * 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (MFMA) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[a], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NVALU; ++v) {
                const int i = (a * NVALU + v) & 7;
                if (F32) y[i] = fmaf(y[i], 0.999f, -0.25f);
                else x[i] = fma(x[i], 0.999, -0.25);
            }
        }
    }
    double s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += x[i] + y[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MFMA, int NVALU, int F32>
void run(const char* name, double* d_out, int blocks_per_cu) {
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MFMA, NVALU, F32>), dim3(grid), dim3(256), 0, 0, d_out, 100, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MFMA, NVALU, F32>), dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // SIMD cycles (at 2.4 GHz) per [1 mfma + NVALU fma] of one wave, with blocks_per_cu waves on the SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0) / blocks_per_cu;
    printf("%-40s waves/SIMD=%d  %.3f ms  -> %.1f SIMD cycles per [%d mfma_f64 + %d %s fma] of a wave\n",
           name, blocks_per_cu, ms, cyc, MFMA, NVALU, F32 ? "f32" : "f64");
}

int main() {
    double* d_out; hipMalloc(&d_out, sizeof(double) * 256 * 256 * 8);
    for (int b : {1, 2, 3}) {
        run<1, 0, 0>("f64 mfma only", d_out, b);
        run<0, 8, 0>("valu only (8 f64 fma)", d_out, b);
        run<1, 8, 0>("f64 mfma + 8 f64 fma", d_out, b);
        run<0, 16, 0>("valu only (16 f64 fma)", d_out, b);
        run<1, 16, 0>("f64 mfma + 16 f64 fma", d_out, b);
        run<0, 16, 1>("valu only (16 f32 fma)", d_out, b);
        run<1, 16, 1>("f64 mfma + 16 f32 fma", d_out, b);
    }
    return 0;
}
