// Machine model probes for gfx950: ticks (s_memtime) and 100 MHz ticks per instruction for the pieces of the CIN loop.
//   hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, long* t, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + blockIdx.x * 97u; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
        // two fp16 values in [-2, 2) with random mantissas
        ((unsigned*)smem)[i] = (x & 0x83ff83ffu) | 0x3c003c00u;
    }
    __syncthreads();
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    h8 a, b, b2, fr[8];
    h8 ar[4][2], nb = b, nb2 = b;
    const float* gin = out;
    for (int k = 0; k < 4; ++k) { ar[k][0] = *(const h8*)(smem + lane * 16 + k * 2048); ar[k][1] = *(const h8*)(smem + lane * 16 + k * 2048 + 1024); }
    f32x16 acc2[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc2[k][r] = 0.f;
    for (int r = 0; r < 8; ++r) b2[r] = (_Float16)(r * 0.25f);
    for (int k = 0; k < 8; ++k) fr[k] = b2;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(lane * 0.01f + r); b[r] = (_Float16)(r * 0.5f); }
    float v[8];
    for (int r = 0; r < 8; ++r) v[r] = lane + r;
    const char* st = smem + lane * 16;
    const long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // 12 independent-ish MFMAs
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 3], 0, 0, 0);
        } else if (MODE == 1) {     // 12 MFMAs + 8 ds_read_b128 feeding them
            h8 f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = *(const h8*)(st + k * 1024 + (it & 7) * 8192);
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k & 7], b, acc[k & 3], 0, 0, 0);
        } else if (MODE == 2) {     // 32 dependent-free VALU fmas
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
        } else if (MODE == 3) {     // 16 ds_read_b128 only
            h8 f[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) f[k] = *(const h8*)(st + k * 1024 + (it & 3) * 16384);
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k & 7] += f[k][k & 7];
        } else if (MODE == 4) {     // 12 MFMAs + 32 VALU interleaved
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 3], 0, 0, 0);
                v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
                v[(k + 3) & 7] = __builtin_fmaf(v[(k + 3) & 7], 1.0001f, 0.5f);
                v[(k + 5) & 7] = __builtin_fmaf(v[(k + 5) & 7], 1.0001f, 0.5f);
            }
        } else if (MODE == 6) {     // 12 MFMAs, random operands held in registers (8 A fragments, 2 B)
            if (it == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) fr[k] = *(const h8*)(st + k * 1024);
                b = *(const h8*)(st + 9 * 1024); b2 = *(const h8*)(st + 10 * 1024);
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[k & 7], (k & 4) ? b2 : b, acc[k & 3], 0, 0, 0);
        } else if (MODE == 7) {     // 24 MFMAs per 8 ds_read_b128 (every fragment meets two B operands), random data
            h8 f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = *(const h8*)(st + k * 1024 + (it & 7) * 8192);
            if (it == 0) { b = f[3]; b2 = f[5]; }
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k & 7], b, acc[k & 3], 0, 0, 0);
                acc2[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k & 7], b2, acc2[k & 3], 0, 0, 0);
            }
        } else if (MODE == 8) {     // 12 MFMAs per 8 ds_read_b128, random data
            h8 f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = *(const h8*)(st + k * 1024 + (it & 7) * 8192);
            if (it == 0) { b = f[3]; }
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k & 7], b, acc[k & 3], 0, 0, 0);
        } else if (MODE == 9) {     // 12 MFMAs chained through ONE accumulator
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[k & 7], b, acc[0], 0, 0, 0);
        } else if (MODE == 10) {    // 12 MFMAs alternating between two accumulators
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[k & 7], b, acc[k & 1], 0, 0, 0);
        } else if (MODE >= 11 && MODE <= 18) {
            // the forward kernel's step: REG regions of (look-ahead fragments of the next region, MFMAs of this one on
            // 12 / REG MFMAs); 11: 2 regions x 6 MFMAs (2 row tiles), 12: 1 region x 12 (4 row tiles), 13: as 11 without
            // the scheduling fences, 14: as 11 with the hi*hi, hi*lo, lo*hi MFMAs of a tile back to back
            constexpr int REG = MODE == 12 ? 1 : 2;
            constexpr int TG = 4 / REG;                  // row tiles per region
            const char* base = st + (it & 7) * 8192;
            if (it == 0) { b = *(const h8*)(st + 9 * 1024); b2 = *(const h8*)(st + 10 * 1024); }      // random B operands too
#pragma unroll
            for (int p = 0; p < REG; ++p) {
                h8 an[TG][2];
                const char* nx = p + 1 < REG ? base + (p + 1) * TG * 2048 : st + ((it + 1) & 7) * 8192;
#pragma unroll
                for (int k = 0; k < TG; ++k) { an[k][0] = *(const h8*)(nx + k * 2048); an[k][1] = *(const h8*)(nx + k * 2048 + 1024); }
                if (MODE == 14) {
#pragma unroll
                    for (int k = 0; k < TG; ++k) {
                        acc[TG * p + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[k][0], b, acc[TG * p + k], 0, 0, 0);
                        acc[TG * p + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[k][0], b2, acc[TG * p + k], 0, 0, 0);
                        acc[TG * p + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[k][1], b, acc[TG * p + k], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < 3; ++f)
#pragma unroll
                        for (int k = 0; k < TG; ++k)
                            acc[TG * p + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[k][f == 2 ? 1 : 0], f == 1 ? b2 : b, acc[TG * p + k], 0, 0, 0);
                }
                if (MODE == 15 || MODE == 18) {                  // + the B-operand arithmetic of the CIN step (2 pairs per region)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float x = v[(2 * p + q) & 7], y0 = v[(q + 3) & 7], y1 = v[(q + 5) & 7];
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const f2 z = (f2){x, x} * (f2){y0, y1};
                        const h2 hi = __builtin_convertvector(z, h2);
                        const unsigned hb = __builtin_bit_cast(unsigned, hi);
                        float r0, r1;
                        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x), "v"(y0), "v"(hb));
                        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x), "v"(y1), "v"(hb));
                        const f2 rr = {r0, r1};
                        const h2 lo = __builtin_convertvector(rr, h2);
                        nb[2 * (2 * p + q)] = hi.x; nb[2 * (2 * p + q) + 1] = hi.y;
                        nb2[2 * (2 * p + q)] = lo.x; nb2[2 * (2 * p + q) + 1] = lo.y;
                    }
                }
                if ((MODE == 16 || MODE == 18) && p == REG - 1 && (it & 1)) __builtin_amdgcn_s_barrier();
                if ((MODE == 17 || MODE == 18) && p == REG - 1 && (it & 1)) {
                    const unsigned lo_ = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + 40000 + (threadIdx.x >> 6) * 2048));
                    const float* gsrc = gin + ((it >> 1) & 15) * 4096 + (threadIdx.x >> 6) * 512 + lane * 4;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024" ::"v"(gsrc), "s"(lo_) : "memory", "m0");
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
#pragma unroll
                for (int k = 0; k < TG; ++k) { ar[k][0] = an[k][0]; ar[k][1] = an[k][1]; }
                if (MODE != 13) {
                    __builtin_amdgcn_sched_group_barrier(0x100, TG * 2, 0);
#pragma unroll
                    for (int i = 0; i < TG * 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 15 || MODE == 18) { b = nb; b2 = nb2; }
        } else if (MODE == 5) {     // 12 MFMAs + barrier
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 3], 0, 0, 0);
            __builtin_amdgcn_s_barrier();
        }
    }
    const long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r] + acc2[k][r];
    for (int k = 0; k < 8; ++k) s += (float)fr[k][k];
    for (int k = 0; k < 4; ++k) s += (float)ar[k][0][k] + (float)ar[k][1][k];
    for (int r = 0; r < 8; ++r) s += v[r] + (float)a[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
static void run(const char* name, int waves, int iters, int per_iter) {
    const int blocks = 256;
    float* out; long* t;
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipMalloc(&t, blocks * 2 * sizeof(long));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64 * waves), 65536, 0, out, t, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64 * waves), 65536, 0, out, t, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long> h(blocks * 2);
    hipMemcpy(h.data(), t, blocks * 2 * sizeof(long), hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int b = 0; b < blocks; ++b) { c += h[2 * b]; r += h[2 * b + 1]; }
    c /= blocks; r /= blocks;
    printf("%-34s %d waves/CU: %8.1f ticks/iter (%6.2f per op, %d ops/wave/iter)  %7.1f ns/iter  tick rate %.0f MHz | events %7.1f ns/iter\n", name, waves,
           c / iters, c / iters / per_iter, per_iter, r * 10.0 / iters, c / (r / 100.0), ms * 1e6 / iters);
    hipFree(out); hipFree(t);
}

int main(int argc, char** argv) {
    if (argc > 3) {                 // the forward kernel's region structure against the plain loop
        const int iters = atoi(argv[1]);
        for (int rep = 0; rep < 2; ++rep) {
            run<8>("12 MFMA + 8 ds_read random (plain)", 8, iters, 12);
            run<11>("2 regions x (4 reads ahead, 6 MFMA)", 8, iters, 12);
            run<12>("1 region x (8 reads ahead, 12 MFMA)", 8, iters, 12);
            run<13>("2 regions, no scheduling fences", 8, iters, 12);
            run<14>("2 regions, a tile's 3 MFMAs back to back", 8, iters, 12);
            run<15>("2 regions + B-operand arithmetic", 8, iters, 12);
            run<16>("2 regions + barrier every 2 steps", 8, iters, 12);
            run<17>("2 regions + LDS-DMA 2 KB per wave per 2 steps", 8, iters, 12);
            run<18>("2 regions + all three", 8, iters, 12);
        }
        return 0;
    }
    if (argc > 2) {                 // dependency chains
        const int iters = atoi(argv[1]);
        for (int waves : {4, 8}) {
            run<0>("12 MFMA, 4 accumulators", waves, iters, 12);
            run<10>("12 MFMA, 2 accumulators", waves, iters, 12);
            run<9>("12 MFMA, 1 accumulator", waves, iters, 12);
        }
        return 0;
    }
    if (argc > 1) {                 // sustained load: long runs of the MFMA modes
        const int iters = atoi(argv[1]);
        for (int rep = 0; rep < 3; ++rep) {
            run<0>("12 MFMA 32x32x16 f16 (long)", 8, iters, 12);
            run<1>("12 MFMA + 8 ds_read_b128 (long)", 8, iters, 12);
            run<4>("12 MFMA + 36 VALU (long)", 8, iters, 12);
            run<6>("12 MFMA random regs (long)", 8, iters, 12);
            run<8>("12 MFMA + 8 ds_read random (long)", 8, iters, 12);
            run<7>("24 MFMA + 8 ds_read random (long)", 4, iters, 24);
            run<7>("24 MFMA + 8 ds_read random (long)", 8, iters, 24);
        }
        return 0;
    }
    for (int waves : {4, 8}) {
        run<0>("12 MFMA 32x32x16 f16", waves, 2000, 12);
        run<1>("12 MFMA + 8 ds_read_b128", waves, 2000, 12);
        run<2>("32 VALU fma", waves, 2000, 32);
        run<3>("16 ds_read_b128", waves, 2000, 16);
        run<4>("12 MFMA + 36 VALU", waves, 2000, 12);
        run<5>("12 MFMA + s_barrier", waves, 2000, 12);
    }
    return 0;
}
