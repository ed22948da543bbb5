// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: N independent accumulator chains per wave, W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int CH>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a0, double b0) {
    f64x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
void run(int blocks_per_cu) {
    int dev = 0; hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount, blocks = cus * blocks_per_cu, iters = 20000;
    double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, 2.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = double(blocks) * 4 * iters * CH;            // wave-level instructions
    const double tf = mfmas * 2048 / (ms * 1e-3) / 1e12;
    const double ns_per_mfma_simd = ms * 1e6 / (double(iters) * CH * blocks_per_cu);   // per SIMD (one wave of each block per SIMD)
    printf("chains %d  waves/SIMD %d  %.2f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD\n", CH, blocks_per_cu, ms, tf, ns_per_mfma_simd);
    hipFree(out);
}

int main() {
    run<1>(1); run<2>(1); run<4>(1); run<8>(1);
    run<1>(2); run<4>(2); run<4>(4);
    return 0;
}
