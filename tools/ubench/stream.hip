// HBM streaming probes for the Adam step's access pattern: three arrays read and written (p, m, v), one mark byte per
// 16-byte chunk.  Variants of the loop shape / cache hints, GB/s of the 24 bytes per element actually needed.
//   hipcc --offload-arch=gfx950 -O3 -o stream stream.hip && ./stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntl(const float4* a) { const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a)); return make_float4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void nts(float4 x, float4* a) { const v4f t = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(a)); }
__device__ __forceinline__ float4 upd(float4 a, float s) { a.x = a.x * s + 1.f; a.y = a.y * s + 1.f; a.z = a.z * s + 1.f; a.w = a.w * s + 1.f; return a; }

template <int MODE>
__global__ __launch_bounds__(256) void k(float4* __restrict__ p, float4* __restrict__ m, float4* __restrict__ v,
                                         const unsigned char* __restrict__ marks, long n4, float s) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    if (MODE == 0) {            // grid-stride, two chunks per array in flight (the current kernel)
        long i = tid;
        for (; i + stride < n4; i += 2 * stride) {
            const unsigned char ka = marks[i], kb = marks[i + stride];
            float4 pa = p[i], ma = m[i], va = v[i], pb = p[i + stride], mb = m[i + stride], vb = v[i + stride];
            const float t = (ka | kb) ? s : s * 1.0001f;
            p[i] = upd(pa, t); m[i] = upd(ma, t); v[i] = upd(va, t);
            p[i + stride] = upd(pb, t); m[i + stride] = upd(mb, t); v[i + stride] = upd(vb, t);
        }
    } else if (MODE == 1) {     // same, non-temporal loads and stores
        long i = tid;
        for (; i + stride < n4; i += 2 * stride) {
            const unsigned char ka = marks[i], kb = marks[i + stride];
            float4 pa = ntl(p + i), ma = ntl(m + i), va = ntl(v + i);
            float4 pb = ntl(p + i + stride), mb = ntl(m + i + stride), vb = ntl(v + i + stride);
            const float t = (ka | kb) ? s : s * 1.0001f;
            nts(upd(pa, t), p + i); nts(upd(ma, t), m + i); nts(upd(va, t), v + i);
            nts(upd(pb, t), p + i + stride); nts(upd(mb, t), m + i + stride); nts(upd(vb, t), v + i + stride);
        }
    } else if (MODE == 2) {     // four chunks per array in flight
        long i = tid;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            float4 a[4], b[4], c[4];
            unsigned char kk = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { kk |= marks[i + q * stride]; a[q] = p[i + q * stride]; b[q] = m[i + q * stride]; c[q] = v[i + q * stride]; }
            const float t = kk ? s : s * 1.0001f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { p[i + q * stride] = upd(a[q], t); m[i + q * stride] = upd(b[q], t); v[i + q * stride] = upd(c[q], t); }
        }
    } else if (MODE == 3) {     // block-contiguous: a block owns one contiguous span, walks it 4 KB at a time, 2 in flight
        const long per = (n4 + gridDim.x - 1) / gridDim.x;
        const long lo = (long)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
        for (long i = lo + threadIdx.x; i + 256 < hi; i += 512) {
            const unsigned char ka = marks[i], kb = marks[i + 256];
            float4 pa = p[i], ma = m[i], va = v[i], pb = p[i + 256], mb = m[i + 256], vb = v[i + 256];
            const float t = (ka | kb) ? s : s * 1.0001f;
            p[i] = upd(pa, t); m[i] = upd(ma, t); v[i] = upd(va, t);
            p[i + 256] = upd(pb, t); m[i + 256] = upd(mb, t); v[i + 256] = upd(vb, t);
        }
    } else if (MODE == 6 || MODE == 7) {     // non-temporal x2 (6) / x4 (7) with the Adam arithmetic (IEEE sqrt and two divisions per element)
        constexpr int Q = MODE == 6 ? 2 : 4;
        long i = tid;
        for (; i + (Q - 1) * stride < n4; i += Q * stride) {
            float4 a[Q], b[Q], c[Q];
            unsigned char kk = 0;
#pragma unroll
            for (int q = 0; q < Q; ++q) { kk |= marks[i + q * stride]; a[q] = ntl(p + i + q * stride); b[q] = ntl(m + i + q * stride); c[q] = ntl(v + i + q * stride); }
            const float g = kk ? s : 0.f;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                float* pp = &a[q].x; float* mm = &b[q].x; float* vv = &c[q].x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ge = fmaf(2e-5f, pp[e], g);
                    mm[e] = fmaf(0.1f, ge - mm[e], mm[e]);
                    vv[e] = fmaf(0.001f * ge, ge, 0.999f * vv[e]);
                    const float den = sqrtf(vv[e]) / 0.0316f + 1e-8f;
                    pp[e] -= 0.01f * mm[e] / den;
                }
                nts(a[q], p + i + q * stride); nts(b[q], m + i + q * stride); nts(c[q], v + i + q * stride);
            }
        }
    } else if (MODE == 4) {     // read-only (three arrays): the read side of the budget
        long i = tid;
        float acc = 0.f;
        for (; i + stride < n4; i += 2 * stride) {
            float4 pa = p[i], ma = m[i], va = v[i], pb = p[i + stride], mb = m[i + stride], vb = v[i + stride];
            acc += pa.x + ma.y + va.z + pb.x + mb.y + vb.z;
        }
        if (acc == 12345.f) p[0].x = acc;
    } else if (MODE == 5) {     // write-only
        long i = tid;
        const float4 z = make_float4(s, s, s, s);
        for (; i + stride < n4; i += 2 * stride) { p[i] = z; m[i] = z; v[i] = z; p[i + stride] = z; m[i + stride] = z; v[i + stride] = z; }
    }
}

template <int MODE>
static void run(const char* name, int blocks, float4* p, float4* m, float4* v, unsigned char* marks, long n4, double bytes_per) {
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, p, m, v, marks, n4, 0.999f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, p, m, v, marks, n4, 0.999f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s blocks %6d: %8.1f us  %7.1f GB/s\n", name, blocks, ms * 200.0, bytes_per * n4 * 16 / (ms / 5 * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const long n4 = argc > 1 ? atol(argv[1]) : (long)40e6;          // float4 chunks per array (40 M = 640 MB per array)
    float4 *p, *m, *v; unsigned char* marks;
    hipMalloc(&p, n4 * 16); hipMalloc(&m, n4 * 16); hipMalloc(&v, n4 * 16); hipMalloc(&marks, n4);
    hipMemset(p, 0, n4 * 16); hipMemset(m, 0, n4 * 16); hipMemset(v, 0, n4 * 16); hipMemset(marks, 0, n4);
    for (int blocks : {1024, 2048, 5120, 16384}) {
        run<0>("grid-stride x2 (current)", blocks, p, m, v, marks, n4, 6.0625);
        run<1>("grid-stride x2, non-temporal", blocks, p, m, v, marks, n4, 6.0625);
        run<2>("grid-stride x4", blocks, p, m, v, marks, n4, 6.0625);
        run<3>("block-contiguous x2", blocks, p, m, v, marks, n4, 6.0625);
        run<6>("non-temporal x2 + Adam arithmetic", blocks, p, m, v, marks, n4, 6.0625);
        run<7>("non-temporal x4 + Adam arithmetic", blocks, p, m, v, marks, n4, 6.0625);
        run<4>("read only", blocks, p, m, v, marks, n4, 3.0);
        run<5>("write only", blocks, p, m, v, marks, n4, 3.0);
    }
    return 0;
}
