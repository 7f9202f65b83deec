// Probe for v_mfma_f32_32x32x16_f16 on gfx950: (1) operand / result lane layout against a host GEMM,
// (2) sustained rate of the 3-MFMA "split" group used by the f16x3 CIN kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_probe mfma_f16_probe.hip ; run: ./mfma_f16_probe [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void layout_kernel(const float* A, const float* B, float* C) {   // A[32][16], B[16][32], C[32][32]
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    h8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (_Float16)A[r * 16 + 8 * h + t]; b[t] = (_Float16)B[(8 * h + t) * 32 + r]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int q = 0; q < 16; ++q) C[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    h8 ah, al, bh, bl;
    for (int t = 0; t < 8; ++t) {
        ah[t] = (_Float16)in[threadIdx.x + t]; al[t] = (_Float16)in[threadIdx.x + 8 + t];
        bh[t] = (_Float16)in[threadIdx.x + 16 + t]; bl[t] = (_Float16)in[threadIdx.x + 24 + t];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[k], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 2;
    float *dA, *dB, *dC;
    hipMalloc(&dA, 512 * 4); hipMalloc(&dB, 512 * 4); hipMalloc(&dC, 1024 * 4);
    std::vector<float> A(512), B(512), C(1024);
    for (auto& v : A) v = (float)(rand() % 17 - 8);
    for (auto& v : B) v = (float)(rand() % 17 - 8);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j];
        bad += s != C[i * 32 + j];
    }
    printf("layout check: %d mismatches of 1024\n", bad);
    int iters = 20000, blocks = 256 * wps;
    float *in, *out; hipMalloc(&in, 1024 * 4); hipMalloc(&out, blocks * 256 * 4);
    std::vector<float> h(1024); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<4>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mf = (double)blocks * 4 * iters * 4 * 3;                 // MFMAs issued
        printf("waves/SIMD %d: %.3f ms  %.0f TFLOP/s f16 issued = %.0f TFLOP/s fp32-equivalent (3 MFMAs per product)\n", wps, ms,
               mf * 32768.0 / ms / 1e9, mf * 32768.0 / 3 / ms / 1e9);
    }
    return 0;
}
