// ALU rate of the Adam update when a row is replayed from registers (no memory traffic): element-steps per second.
#include <hip/hip_runtime.h>
#include <cstdio>
struct C { float w1, b2, w2, eps, g2; };
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float ss, float bc2, const C& c) {
#pragma clang fp contract(off)
    m = fmaf(c.w1, g - m, m);
    v = fmaf(c.w2 * g, g, c.b2 * v);
    const float denom = sqrtf(v) / bc2 + c.eps;
    p -= ss * m / denom;
}
__global__ __launch_bounds__(256) void k(float4* p4, const float2* __restrict__ stepc, int steps, float* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float4 p = p4[i], m = make_float4(0, 0, 0, 0), v = make_float4(1e-6f, 1e-6f, 1e-6f, 1e-6f);
    const C c = {0.1f, 0.999f, 0.001f, 1e-8f, 2e-5f};
    for (int s = 0; s < steps; ++s) {
        const float2 sc = stepc[s];          // (step size, sqrt(1 - b2^t)): wave-uniform scalar load
        adam_one(p.x, c.g2 * p.x, m.x, v.x, sc.x, sc.y, c); adam_one(p.y, c.g2 * p.y, m.y, v.y, sc.x, sc.y, c);
        adam_one(p.z, c.g2 * p.z, m.z, v.z, sc.x, sc.y, c); adam_one(p.w, c.g2 * p.w, m.w, v.w, sc.x, sc.y, c);
    }
    p4[i] = p;
    if (m.x + v.x == 12345.f) out[0] = m.y + v.z;
}
int main() {
    const int blocks = 256 * 8, steps = 2000;
    const long n = (long)blocks * 256;
    float4* p; float2* sc; float* out;
    hipMalloc(&p, n * 16); hipMalloc(&sc, steps * 8); hipMalloc(&out, 4);
    hipMemset(p, 0x3c, n * 16);
    float2* h = new float2[steps];
    for (int s = 0; s < steps; ++s) { h[s].x = 1e-3f / (1.f - powf(0.9f, s + 1.f)); h[s].y = sqrtf(1.f - powf(0.999f, s + 1.f)); }
    hipMemcpy(sc, h, steps * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, p, sc, steps, out);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, p, sc, steps, out);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double es = (double)n * 4 * steps;
    printf("%.3f ms for %.2f G element-steps: %.1f G element-steps/s -> 575 M parameters = %.2f ms per step\n", ms, es / 1e9, es / ms / 1e6, 575e6 / (es / ms / 1e3) );
    return 0;
}
