// Chunk loop with the A fragments read straight from the L2-resident split table (no LDS tile, no
// barrier, no LDS-DMA), prefetched PF chunks ahead in registers; mu also from global.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

template <int PF>
__global__ void __launch_bounds__(256) k(const char* __restrict__ table, const float* __restrict__ mu_all, float* out,
                                         int n_chunks, int reps) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    bf16x8 B[9];
    for (int s = 0; s < 9; ++s) for (int e = 0; e < 8; ++e) B[s][e] = (short)(0x3000 + s + e + lane);
    float total = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        bf16x8 A[PF + 1][4];
        float4 M[PF + 1][4];
        auto fetch = [&](int c, int slot) {
            const char* arow = table + (size_t)(c * 32 + j) * 144 + 16 * h;
            const float* mu = mu_all + c * 32 + 4 * h;
#pragma unroll
            for (int s = 0; s < 4; ++s) A[slot][s] = *reinterpret_cast<const bf16x8*>(arow + 32 * s);
#pragma unroll
            for (int q = 0; q < 4; ++q) M[slot][q] = *reinterpret_cast<const float4*>(mu + 8 * q);
        };
#pragma unroll
        for (int p = 0; p < PF; ++p) fetch(p, p);
        for (int c = 0; c < n_chunks; c += PF + 1) {
#pragma unroll
            for (int u = 0; u <= PF; ++u) {
                const int cc = c + u;
                const int nxt = min(cc + PF, n_chunks - 1);
                fetch(nxt, (u + PF) % (PF + 1));
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[4*q] = M[u][q].x; acc[4*q+1] = M[u][q].y; acc[4*q+2] = M[u][q].z; acc[4*q+3] = M[u][q].w; }
#pragma unroll
                for (int m = 0; m < 9; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[u][m & 3], B[m], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = __builtin_amdgcn_exp2f(acc[r] * 1e-3f);
                f32x2 p0 = {acc[0], acc[1]}, p1 = {acc[2], acc[3]}, p2 = {acc[4], acc[5]}, p3 = {acc[6], acc[7]};
                const f32x2 p4 = {acc[8], acc[9]}, p5 = {acc[10], acc[11]}, p6 = {acc[12], acc[13]}, p7 = {acc[14], acc[15]};
                p0 += p4; p1 += p5; p2 += p6; p3 += p7; p0 += p2; p1 += p3; p0 += p1;
                total += p0[0] + p0[1];
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

template <int PF>
void run(const char* name, const char* d_tab, const float* d_mu, float* d_out, int bpc) {
    const int n_chunks = 312, reps = 100, grid = 256 * bpc, lds = 160 * 1024 / bpc - 2048;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<PF>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PF>), dim3(grid), dim3(256), lds, 0, d_tab, d_mu, d_out, n_chunks, 2);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<PF>), dim3(grid), dim3(256), lds, 0, d_tab, d_mu, d_out, n_chunks, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-36s blocks/CU=%d: %7.1f cycles per chunk per SIMD\n", name, bpc, ms * 1e-3 * 2.4e9 / (double(n_chunks) * reps * bpc));
}

int main() {
    const size_t tab_bytes = (size_t)10240 * 144;
    std::vector<unsigned short> h(tab_bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3000 + (i * 7) % 1000);
    std::vector<float> hm(10240, 0.5f);
    char* d_tab; float *d_mu, *d_out;
    (void)hipMalloc(&d_tab, tab_bytes); (void)hipMalloc(&d_mu, 10240 * 4); (void)hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8);
    (void)hipMemcpy(d_tab, h.data(), tab_bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_mu, hm.data(), 10240 * 4, hipMemcpyHostToDevice);
    for (int b : {2, 3, 4}) {
        run<1>("A, mu from L2, prefetch 1 chunk", d_tab, d_mu, d_out, b);
        run<2>("A, mu from L2, prefetch 2 chunks", d_tab, d_mu, d_out, b);
        run<3>("A, mu from L2, prefetch 3 chunks", d_tab, d_mu, d_out, b);
    }
    return 0;
}
