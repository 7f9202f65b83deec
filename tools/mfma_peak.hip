// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate and in-kernel clock on this device.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                 unsigned long long* stamps) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
        a += 1e-9f;   // a trickle of VALU work, as in a real kernel
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k)
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 2;          // waves per SIMD
    int iters = 20000;
    int blocks = 256 * wps;                           // 4 waves per block, one per SIMD
    float *in, *out; unsigned long long* st;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
    std::vector<float> h(512);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, in, out, iters, st);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hs(blocks * 2);
        hipMemcpy(hs.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        double clk = 0; for (int i = 0; i < blocks; ++i) clk += (double)hs[2 * i] / hs[2 * i + 1] * 100e6; clk /= blocks;
        double flops = (double)blocks * 4 * iters * 4 * 4096.0;
        printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s  in-kernel clock %.3f GHz  (peak at that clock %.1f TF)\n", wps, ms,
               flops / ms / 1e9, clk / 1e9, clk * 256 * 4 * 64 / 1e12);
    }
    return 0;
}
