// f16x3 arithmetic for the CIN contraction (option "cin_math" = 1): fp32 operands are split on the fly
// into two fp16 halves (x = hi + lo, 22 mantissa bits) and every fp32 product is formed by THREE
// v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi) accumulated in fp32.  The dropped lo*lo term is 2^-22
// relative, i.e. below the rounding of the fp32 accumulation itself over K = Hp*m terms: against an fp64
// GEMM this path measures the same (slightly smaller) error as the plain fp32 MFMA path
// (tests/test_gpu_parity.py::test_cin_x3_*).  The matrix pipe runs 16 k per 32 cycles instead of 2 k per
// 64 cycles, so three MFMAs per product are still 5.3x the fp32 MFMA rate.
//
// fp16 has a 5-bit exponent, so operands are range-fitted with exact power-of-two scales that are
// removed from the fp32 accumulator in the epilogue: one scale for the weight matrix (|W| <  2^15) and
// one per column n for x0[:, n] and x_prev[:, n] (|Z| < 2^14).  Elements more than 2^18 below their
// column's maximum lose relative precision but keep an absolute error of 2^-40 of that maximum.
//
// This file: the forward kernel (K3) and its weight pack.  deepctr/layers/interaction.py:218-229.
#include <type_traits>
#include "xdfm_internal.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));      // 16 bytes of MFMA operand (fp16 halves, or bf16 bit patterns)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

// two fp32 values -> two bf16 (RNE, v_cvt_pk_bf16_f32) carried as the bit patterns of an h2
__device__ __forceinline__ h2 x3_bf16_pair(float a, float b) {
    const f2 z = {a, b};
    return __builtin_bit_cast(h2, __builtin_convertvector(z, bf2));
}
// one MFMA term: NT == 3 -> fp16 operands, NT == 1 -> the same 16 bytes read as bf16
template <int NT>
__device__ __forceinline__ f32x16 x3_mfma(const h8& a, const h8& b, const f32x16& c) {
    if constexpr (NT == 3) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// ---------------------------------------------------------------------------------------------
// geometry shared by pack and kernel.  The contraction index k = (i, j) is walked in blocks of 8 rows
// i of x_prev: lane half hh (= lane >> 5) owns rows blk*8 + hh*RH + il, il < RH (RH = 4; in the ragged
// last block RH = ceil(rows_left / 2)).  Inside a block, MFMA step s covers the flat positions
// q = 8s .. 8s+7 of the half's (il, j) list, q = il*m + j, so with an even m both halves run the same
// compile-time (il, j) pattern and only their x_prev rows differ.
X3Geom x3_fwd_geom(int H, int Hp, int m) {
    X3Geom g;
    const int t = ceil_div(H, 32);
    g.MT = t >= 8 ? 8 : (t > 2 ? 4 : 2);
    const int o = xdfm_opt(OPT_X3_FWD_MT);          // A/B knob: cap the row tiles per wave (more, smaller workgroups)
    if ((o == 2 || o == 4) && g.MT > o) g.MT = o;
    g.MB = ceil_div(H, 32 * g.MT);
    g.MP = m / 2;
    g.FB = Hp / 8;
    const int R = Hp - 8 * g.FB;
    g.RH = (R + 1) / 2;
    g.TS = ceil_div(g.RH * m, 8);
    g.NS = g.FB * g.MP + g.TS;
    return g;
}

bool x3_fwd_usable(int H, int Hp, int m) {
    (void)Hp;
    const int nt = x3_terms();
    // bf16: one 1-KB fragment per row tile and step, so a ring stage of < 4 row tiles cannot be dealt to 4 waves
    return nt != 0 && (m == 26 || m == 22) && H > (nt == 3 ? 32 : 64);
}

// power of two s with amax*s in [2^(target-1), 2^target); 1 for amax == 0 or denormal
__device__ __forceinline__ float x3_pow2_scale(float amax, int target) {
    const int E = (int)((__float_as_uint(amax) >> 23) & 0xff);
    int be = 253 + target - E;
    be = be < 1 ? 1 : (be > 253 ? 253 : be);
    return E == 0 ? 1.f : __uint_as_float((unsigned)be << 23);
}

__device__ __forceinline__ void x3_split2(float z0, float z1, h2& hi, h2& lo) {
    const f2 z = {z0, z1};
    hi = __builtin_convertvector(z, h2);                 // v_cvt_pk_f16_f32 (RNE)
    const f2 r = {z0 - (float)hi.x, z1 - (float)hi.y};   // exact in fp32
    lo = __builtin_convertvector(r, h2);
}

// hi / lo halves of the two products a0*b0, a1*b1: one v_pk_mul_f32 + v_cvt_pk_f16_f32 for hi, then
// lo = rne16(a*b - hi) with the exact product inside one v_fma_mix_f32 per element (its third operand is the
// fp16 half, read in place) -- 5 VALU instructions per pair.
__device__ __forceinline__ void x3_split_prod2(float a0, float b0, float a1, float b1, h2& hi, h2& lo) {
    const f2 z = (f2){a0, a1} * (f2){b0, b1};
    hi = __builtin_convertvector(z, h2);
    const unsigned hbits = __builtin_bit_cast(unsigned, hi);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "v"(b0), "v"(hbits));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "v"(b1), "v"(hbits));
    const f2 r = {r0, r1};
    lo = __builtin_convertvector(r, h2);
}

// max_i |col[i * N]| over rows i = first, first + 2, ... < rows: 16 unconditional loads in flight per round trip
// (rows past the end are clamped to a row of the set: harmless for a maximum; more in flight costs the MT = 8
// kernels registers they do not have).  A loop with a few loads per iteration pays an L2 round trip per iteration.
__device__ __forceinline__ float x3_col_absmax(const float* __restrict__ col, long N, int first, int rows) {
    float mx = 0.f;
    for (int i0 = first; i0 < rows; i0 += 32) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = i0 + 2 * k;
            v[k] = col[(long)(i < rows ? i : first) * N];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) mx = fmaxf(mx, fabsf(v[k]));
    }
    return first < rows ? mx : 0.f;
}

// ---------------------------------------------------------------------------------------------
// |W| maximum: every block stores its partial maximum in header slot X3_HDR_PART + blockIdx.x (plain stores,
// no zero-initialised cell and no atomics: nothing here depends on a memset node inside a captured graph);
// the pack kernels reduce the <= X3_ABSMAX_BLOCKS partials themselves.
#define X3_ABSMAX_BLOCKS 64
#define X3_HDR_PART 64
__global__ __launch_bounds__(1024) void x3_absmax_kernel(const float* __restrict__ W, long total, float* __restrict__ hdr) {
    __shared__ float red[16];
    float v = 0.f;
    const long nv = total >> 2;
    const bool vec = (((size_t)W) & 15) == 0;
    if (vec) {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
            const float4 a = reinterpret_cast<const float4*>(W)[i];
            v = fmaxf(fmaxf(v, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
        }
    }
    for (long i = (vec ? nv * 4 : 0) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        v = fmaxf(v, fabsf(W[i]));
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x < 16) {
        v = red[threadIdx.x];
        for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        if (threadIdx.x == 0) hdr[X3_HDR_PART + blockIdx.x] = v;
    }
}

static inline int x3_absmax_blocks(long total) {
    const int b = ceil_div(total, 1024 * 16);
    return b > X3_ABSMAX_BLOCKS ? X3_ABSMAX_BLOCKS : b;
}

static void x3_launch_absmax(const float* W, long total, float* pack, hipStream_t st) {
    hipLaunchKernelGGL(x3_absmax_kernel, dim3(x3_absmax_blocks(total)), dim3(1024), 0, st, W, total, pack);
}

__device__ __forceinline__ float x3_weight_scale(const float* __restrict__ hdr, int nparts) {
    float mx = 0.f;
    for (int k = 0; k < nparts; ++k) mx = fmaxf(mx, hdr[X3_HDR_PART + k]);
    return x3_pow2_scale(mx, 15);
}

// pack layout (after the X3_HDR-float header: [0] = sW, [1] = 1/sW, [64..127] = partial maxima of |W|):
//   [mb][g < NS+2][mt < MT][p: 0 = hi, 1 = lo][lane][8 halves]   -- 1 KB per (mt, p) fragment
// element t of lane (r = lane & 31, hh = lane >> 5) of step g = (blk, s):
//   row = (mb*MT + mt)*32 + r,  q = 8s + t,  il = q / m,  j = q % m,  i = blk*8 + hh*RH + il
// bf16 (nt == 1): no scale (sW = 1), one fragment per (g, mt): the bf16 bit patterns of W
__device__ __forceinline__ void x3_fwd_pack_one(const float* __restrict__ W, int H, int Hp, int m, const X3Geom& G,
                                                int nparts, float* __restrict__ pack, long idx, int nt) {
    const float sW = nt == 3 ? x3_weight_scale(pack, nparts) : 1.f;
    if (idx == 0) { pack[0] = sW; pack[1] = 1.f / sW; }
    const int lane = (int)(idx & 63);
    long rest = idx >> 6;
    const int mt = (int)(rest % G.MT); rest /= G.MT;
    const int g = (int)(rest % (G.NS + 2));
    const int mb = (int)(rest / (G.NS + 2));
    const int r = lane & 31, hh = lane >> 5;
    const int row = (mb * G.MT + mt) * 32 + r;
    int blk, s, RH;
    if (g < G.FB * G.MP) { blk = g / G.MP; s = g - blk * G.MP; RH = 4; }
    else { blk = G.FB; s = g - G.FB * G.MP; RH = G.RH; }
    h8 hi, lo;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int q = 8 * s + t, il = q / m, j = q - il * m;
        const int i = blk * 8 + hh * RH + il;
        float v = 0.f;
        if (g < G.NS && row < H && il < RH && i < Hp) v = W[(long)row * ((long)Hp * m) + (long)i * m + j] * sW;
        if (nt == 3) {
            const _Float16 a = (_Float16)v;
            hi[t] = a;
            lo[t] = (_Float16)(v - (float)a);
        } else {
            hi[t] = __builtin_bit_cast(_Float16, (__bf16)v);
        }
    }
    if (nt == 3) {
        h8* dst = reinterpret_cast<h8*>(pack + X3_HDR) + (((long)mb * (G.NS + 2) + g) * G.MT + mt) * 128 + lane;
        dst[0] = hi;
        dst[64] = lo;
    } else {
        reinterpret_cast<h8*>(pack + X3_HDR)[(((long)mb * (G.NS + 2) + g) * G.MT + mt) * 64 + lane] = hi;
    }
}

__global__ void x3_fwd_pack_kernel(const float* __restrict__ W, int H, int Hp, int m, X3Geom G, int nparts,
                                   float* __restrict__ pack, int nt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)G.MB * (G.NS + 2) * G.MT * 64) x3_fwd_pack_one(W, H, Hp, m, G, nparts, pack, idx, nt);
}

// ---------------------------------------------------------------------------------------------
// Forward kernel.  Workgroup = 4 waves = 128 columns x (32*MT rows of one row group mb); a wave owns
// 32 columns and all MT row tiles.  The packed weight fragments of one step (2*MT KB) are shared by the
// four waves through a 3-deep LDS ring filled by 16-byte LDS-DMA two steps ahead (one counted
// s_waitcnt vmcnt + raw s_barrier per step); each wave builds its own B operand (Z hi / lo) in registers.
// NT = MFMA terms per product: 3 (f16x3: hi / lo fp16 halves, range-fitted) or 1 (bf16 operands, no scales)
template <int MT, int M, int NW, int R = 3, int NT = 3>
__global__ __launch_bounds__(64 * NW, 8 / NW) void cin_fwd_x3_kernel(
    const float* __restrict__ xp, const float* __restrict__ x0, const float* __restrict__ pack,
    const float* __restrict__ bias, int H, int Hp, long N, X3Geom G, int act, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MP = M / 2;
    constexpr int FRT = NT == 3 ? 2 : 1;        // 1-KB fragments per row tile (hi, lo | bf16)
    constexpr int FR = FRT * MT;                // 1-KB fragments per stage
    constexpr int STAGE = FR * 1024;            // bytes
    constexpr int FPW = FR / NW;                // LDS-DMA instructions per wave and stage
    static_assert(FR % NW == 0, "every wave issues the same number of LDS-DMA pieces");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const long n = ((long)blockIdx.x * NW + wave) * 32 + c;
    const bool nok = n < N;                     // no early exit: every wave feeds the ring and the barriers
    const long nc = nok ? n : N - 1;
    const float nmask = nok ? 1.f : 0.f;
    const int mb = blockIdx.y;
    const int dbg = act >> 8;                   // timing experiments (xdfm option "dbg" bits 6..11): 1 no stores, 2 one block, 8 no operand work, 16 no weight DMA
    act &= 0xff;

    const char* wsrc = reinterpret_cast<const char*>(pack + X3_HDR) + (long)mb * (G.NS + 2) * STAGE + lane * 16;
    auto dma_stage = [&](const char* src, int slot_off) {
#pragma unroll
        for (int k = 0; k < FPW; ++k) {
            const int f = wave * FPW + k;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + f * 1024),
                                             (LDS_AS void*)(smem + slot_off + f * 1024), 16, 0, 0);
        }
    };
    // x_prev rows of a block (8 rows x the workgroup's 32*NW columns) also arrive by LDS-DMA, one block ahead, into
    // two buffers behind the ring: per-lane global loads of them made hipcc drain the whole VMEM queue (s_waitcnt
    // vmcnt(0)) at every block boundary -- the ring's look-ahead with it -- and cost 4 dependent round trips in the
    // prologue.  4-byte pieces: no alignment requirement on N or xp.
    constexpr int XCOLS = 32 * NW;              // columns of the workgroup
    constexpr int XPI = 8 * XCOLS / 64 / NW;    // x_prev DMA instructions per wave and block (256 B each)
    float* xbuf = reinterpret_cast<float*>(smem + R * STAGE);           // [2][8][XCOLS]
    float* bias_s = xbuf + 2 * 8 * XCOLS;                               // [32 * MT] bias of the workgroup's rows
    if ((int)threadIdx.x < 32 * MT) {
        const int row = blockIdx.y * MT * 32 + threadIdx.x;
        bias_s[threadIdx.x] = bias[row < H ? row : H - 1];
    }
    const long col0 = (long)blockIdx.x * XCOLS;
    auto dma_xp = [&](int blk, int buf) {
#pragma unroll
        for (int k = 0; k < XPI; ++k) {
            const int e = (wave * XPI + k) * 64;                        // first element of this piece in the [8][XCOLS] tile
            const int row = e / XCOLS;                                  // wave-uniform
            int i = blk * 8 + row;
            i = i < Hp ? i : Hp - 1;                                    // rows past the matrix: any valid row (their factor is 0)
            long col = col0 + (e - row * XCOLS) + lane;
            col = col < N ? col : N - 1;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(xp + (long)i * N + col),
                                             (LDS_AS void*)(xbuf + buf * 8 * XCOLS + e), 4, 0, 0);
        }
    };
    dma_xp(0, 0);
    // the ring runs R - 1 stages ahead of the step being read (R slots); the packed stream ends with 2 spare stages
    const char* wlast = wsrc + (long)(G.NS + 1) * STAGE;      // last stage that exists (reads past it are clamped to it)
#pragma unroll
    for (int k = 0; k < R - 1; ++k) dma_stage(wsrc + (long)(k < G.NS + 2 ? k : G.NS + 1) * STAGE, k * STAGE);

    // ---- x0 column (registers), column scales --------------------------------------------------
    float x0r[M];
    float a0 = 0.f;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        x0r[j] = x0[(long)j * N + nc] * nmask;
        a0 = fmaxf(a0, fabsf(x0r[j]));
    }
    float ap = 0.f;
    if (NT != 3) {
    } else if (xp == x0) {
        ap = a0;
    } else {
        ap = x3_col_absmax(xp + nc, N, hh, Hp);
        ap = fmaxf(ap, __shfl_xor(ap, 32)) * nmask;
    }
    // bf16 operands need no range fitting
    const float s0 = NT == 3 ? x3_pow2_scale(a0, 7) : 1.f, sp = NT == 3 ? x3_pow2_scale(ap, 7) : 1.f;
#pragma unroll
    for (int j = 0; j < M; ++j) x0r[j] *= s0;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // x_prev rows of this lane half in block blk: blk*8 + hh*RH + il, read from the block's LDS buffer and multiplied
    // by their factor (column scale, or 0 for rows / columns outside the matrix)
    auto read_xp = [&](int blk, int RH, float (&v)[4]) {
        const float* xb = xbuf + (blk & 1) * 8 * XCOLS + wave * 32 + c;
#pragma unroll
        for (int il = 0; il < 4; ++il) {
            const int i = blk * 8 + hh * RH + il;
            const bool ok = il < RH && i < Hp;
            v[il] = xb[(ok ? hh * RH + il : 0) * XCOLS] * (ok ? sp * nmask : 0.f);
        }
    };
    // B operand (hi, lo) of step s of a block from the block's 4 x_prev values
    auto build_b = [&](int s, const float (&xv)[4], h8& bh, h8& bl) {
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            const int q = 8 * s + 2 * t2, il = q / M, j = q - il * M;
            h2 hi = h2{0, 0}, lo = h2{0, 0};
            if (il < 4) {
                if constexpr (NT == 3) x3_split_prod2(xv[il], x0r[j], xv[il], x0r[j + 1], hi, lo);
                else hi = x3_bf16_pair(xv[il] * x0r[j], xv[il] * x0r[j + 1]);
            }
            bh[2 * t2] = hi.x; bh[2 * t2 + 1] = hi.y;
            bl[2 * t2] = lo.x; bl[2 * t2 + 1] = lo.y;
        }
    };

    int so[R];                                  // LDS offsets of the ring slots of steps s % R of this block
#pragma unroll
    for (int q = 0; q < R; ++q) so[q] = q * STAGE;
    const int nblk = G.FB + (G.TS > 0 ? 1 : 0);
    float xv[4], xn[4];
    // block 0's rows have landed (they were issued before the ring's first stages) for every wave
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 1) * FPW) : "memory");
    __builtin_amdgcn_s_barrier();
    read_xp(0, G.FB > 0 ? 4 : G.RH, xv);
    h8 bh, bl;
    build_b(0, xv, bh, bl);
    const char* wblk = wsrc;
    auto block_steps = [&](int blk, int nsteps_dyn, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const bool has_next = blk + 1 < nblk;
#pragma unroll
        for (int s = 0; s < MP; ++s) {
            if (FULL || s < nsteps_dyn) {       // wave-uniform
            h8 nh = bh, nl = bl;                         // operand of the step after this one, built in its shadow
            if (dbg & 8) {
            } else if (s + 1 < MP) {
                if (FULL || s + 1 < nsteps_dyn) build_b(s + 1, xv, nh, nl);
            } else if (has_next) {
                read_xp(blk + 1, blk + 1 < G.FB ? 4 : G.RH, xn);       // landed and published since step R - 1 of this block
                build_b(0, xn, nh, nl);
            }
            // stage (blk, s) has landed for this wave's pieces; after the barrier for everyone's.  Younger than its
            // DMA: the R - 2 stages after it, and in steps 1 .. R-2 of a block the next block's x_prev pieces (issued
            // in step 0 right after the barrier, before that step's stage)
            constexpr int AHEAD = R - 2;
            if (has_next && s >= 1 && s <= AHEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * FPW + XPI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * FPW) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s == 0 && has_next) dma_xp(blk + 1, (blk + 1) & 1);
            {
                const char* src = wblk + (long)(s + R - 1) * STAGE;
                if (!(dbg & 16)) dma_stage(src < wlast ? src : wlast, so[(s + R - 1) % R]);
            }
            const char* st = smem + so[s % R] + lane * 16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h8 ah = *reinterpret_cast<const h8*>(st + (FRT * mt) * 1024);
                acc[mt] = x3_mfma<NT>(ah, bh, acc[mt]);
                if constexpr (NT == 3) {
                    const h8 al = *reinterpret_cast<const h8*>(st + (2 * mt + 1) * 1024);
                    acc[mt] = x3_mfma<NT>(ah, bl, acc[mt]);
                    acc[mt] = x3_mfma<NT>(al, bh, acc[mt]);
                }
            }
            bh = nh; bl = nl;
            }
        }
        // next block: rotate the ring slots by the number of steps taken, first B operand
        const int adv = FULL ? MP : nsteps_dyn;
        wblk += (long)adv * STAGE;
        for (int k = 0; k < adv % R; ++k) {
#pragma unroll
            for (int q = 0; q + 1 < R; ++q) { const int t = so[q]; so[q] = so[q + 1]; so[q + 1] = t; }
        }
        if (has_next) {
#pragma unroll
            for (int il = 0; il < 4; ++il) xv[il] = xn[il];
        }
    };
    for (int blk = 0; blk < ((dbg & 2) ? 1 : G.FB); ++blk) block_steps(blk, MP, std::true_type{});
    if (G.TS > 0) block_steps(G.FB, G.TS, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the two look-ahead stages must land before LDS is released

    // ---- epilogue: remove the scales, bias + activation, FM-layout store --------------------------
    // The bias values of the workgroup's rows sit in LDS since the prologue: their reads count on lgkmcnt, so nothing
    // makes hipcc put an s_waitcnt vmcnt(0) -- which also waits for the previous STORE -- in front of every store
    // (with per-row global loads of the bias it did: 16 write round trips per row tile).
    const float sc = NT == 3 ? pack[1] * (1.f / sp) * (1.f / s0) : 1.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        __builtin_amdgcn_sched_barrier(0);          // one row tile at a time: 16 store addresses live, not 16 * MT
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (mb * MT + mt) * 32 + frag_row(r, hh);
            float v = acc[mt][r] * sc + bias_s[mt * 32 + frag_row(r, hh)];
            if (act == XDFM_ACT_RELU) v = fmaxf(v, 0.f);
            if (row < H && nok && (!(dbg & 1) || acc[mt][r] == 12345.f)) out[(long)row * N + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
size_t x3_fwd_pack_elems(int H, int Hp, int m) {
    const X3Geom g = x3_fwd_geom(H, Hp, m);
    // 2 KB (hi + lo) = 512 floats per (step, row tile); bf16: 1 KB
    return (size_t)X3_HDR + (size_t)g.MB * (g.NS + 2) * g.MT * (x3_terms() == 3 ? 512 : 256);
}

int x3_fwd_pack(const float* W, int H, int Hp, int m, float* pack, hipStream_t st) {
    const X3Geom g = x3_fwd_geom(H, Hp, m);
    const long total = (long)H * Hp * m;
    const int nt = x3_terms();
    if (nt == 3) x3_launch_absmax(W, total, pack, st);
    const long threads = (long)g.MB * (g.NS + 2) * g.MT * 64;
    hipLaunchKernelGGL(x3_fwd_pack_kernel, dim3(ceil_div(threads, 256)), dim3(256), 0, st, W, H, Hp, m, g,
                       x3_absmax_blocks(total), pack, nt);
    return xdfm_check_launch("cin_fwd_pack (f16x3 / bf16)");
}

// NW waves (= 32*NW columns) share one weight ring: 8 waves halve the L2 -> LDS traffic of the ring (every
// workgroup streams the whole packed matrix: 512 x 1.7 MB per launch at config 2 with 4 waves)
template <int MT, int M, int NT>
static int launch_x3(const float* xp, const float* x0, const float* pack, const float* bias, int H, int Hp, long N,
                     const X3Geom& g, int act, float* out, hipStream_t st) {
    constexpr int FR = (NT == 3 ? 2 : 1) * MT;
    constexpr int NWMAX = FR % 8 == 0 ? 8 : 4;
    static_assert(FR % 4 == 0, "a ring stage is dealt to 4 or 8 waves");
    // ring (3 weight stages) + bias of the workgroup's rows + two x_prev buffers
    if constexpr (NWMAX == 8) {
        if (xdfm_opt(OPT_X3_WAVES) != 4 && N >= 256 * 64) {
            const dim3 grid(ceil_div(N, 32 * NWMAX), g.MB), block(64 * NWMAX);
            const size_t ldsx = (size_t)3 * FR * 1024 + 32 * MT * sizeof(float) + (size_t)2 * 8 * 32 * NWMAX * sizeof(float);
            hipLaunchKernelGGL((cin_fwd_x3_kernel<MT, M, NWMAX, 3, NT>), grid, block, ldsx, st, xp, x0, pack, bias, H, Hp, N, g, act, out);
            return xdfm_check_launch("cin_level_fwd (f16x3 / bf16)");
        }
    }
    const size_t lds4 = (size_t)3 * FR * 1024 + 32 * MT * sizeof(float) + (size_t)2 * 8 * 128 * sizeof(float);
    hipLaunchKernelGGL((cin_fwd_x3_kernel<MT, M, 4, 3, NT>), dim3(ceil_div(N, 128), g.MB), dim3(256), lds4, st, xp, x0, pack,
                       bias, H, Hp, N, g, act, out);
    return xdfm_check_launch("cin_level_fwd (f16x3 / bf16)");
}

int x3_level_fwd(const float* xp, const float* x0, const float* pack, const float* bias, int H, int Hp, int m, long N,
                 int act, float* out, hipStream_t st) {
    const X3Geom g = x3_fwd_geom(H, Hp, m);
    const int nt = x3_terms();
    act |= ((xdfm_opt(OPT_DBG) >> 6) & 255) << 8;     // timing experiments (results become wrong): see the kernels' `dbg`
    if ((((size_t)pack) & 15) != 0) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_fwd: packed weights must be 16-byte aligned");
#define X3_CASE(MTV, MV) \
    if (g.MT == MTV && m == MV && nt == 3) return launch_x3<MTV, MV, 3>(xp, x0, pack, bias, H, Hp, N, g, act, out, st);
#define X1_CASE(MTV, MV) \
    if (g.MT == MTV && m == MV && nt == 1) return launch_x3<MTV, MV, 1>(xp, x0, pack, bias, H, Hp, N, g, act, out, st);
    X3_CASE(2, 26) X3_CASE(4, 26) X3_CASE(8, 26)
    X3_CASE(2, 22) X3_CASE(4, 22) X3_CASE(8, 22)
    X1_CASE(4, 26) X1_CASE(8, 26) X1_CASE(4, 22) X1_CASE(8, 22)
#undef X3_CASE
#undef X1_CASE
    return xdfm_fail(XDFM_ERR_INVALID, "cin_level_fwd (f16x3 / bf16): no kernel for MT=%d m=%d terms=%d", g.MT, m, nt);
}

// =============================================================================================
// dX (K4a) in f16x3 arithmetic.   dZ[(i,j)][n] = sum_h W[h][(i,j)] * dOut[h][n]   (never stored)
//   dxp[i][n] += sum_j dZ * x0[j][n],   dx0[j][n] += sum_i dZ * xp[i][n]
// GEMM tile: 32 rows i (one i-block) x 32 columns for a fixed j, contraction over h in blocks of 16.
// B operand = the wave's 32 columns of dOut for ALL h, split into hi / lo once and kept in registers;
// A operand = packed W^T fragments shared by the 4 waves of the workgroup through the LDS-DMA ring
// (stage = up to 8 h-blocks = 16 KB = 24 MFMAs per wave).
// =============================================================================================
X3BwxGeom x3_bwx_geom(int H, int Hp, int m) {
    X3BwxGeom g;
    const int hb = ceil_div(H, 16);
    g.HBT = hb > 8 ? 16 : (hb > 4 ? 8 : (hb > 2 ? 4 : 2));
    g.HBS = g.HBT > 8 ? 8 : g.HBT;
    g.IB = ceil_div(Hp, 32);
    g.NT = (long)g.IB * m;                              // tiles
    return g;
}

bool x3_bwx_usable(int H, int Hp, int m) {
    (void)Hp; (void)m;
    const int nt = x3_terms();
    // bf16: 1 KB per h-block, so a ring stage of < 4 h-blocks (H <= 32) cannot be dealt to 4 waves
    return nt != 0 && H > (nt == 3 ? 16 : 32) && H <= 256;
}

// pack: [tile = iblk*m + j][hb < HBT][p][lane][8 halves]; element t of lane (r, hh):
//   W[h = 16*hb + 8*hh + t][(iblk*32 + r)*m + j] * sW   (0 outside); two dummy stages appended.
__device__ __forceinline__ void x3_bwx_pack_one(const float* __restrict__ W, int H, int Hp, int m, const X3BwxGeom& G,
                                                int nparts, float* __restrict__ pack, long idx, int nt) {
    const float sW = nt == 3 ? x3_weight_scale(pack, nparts) : 1.f;
    if (idx == 0) { pack[0] = sW; pack[1] = 1.f / sW; }
    const int lane = (int)(idx & 63);
    long rest = idx >> 6;
    const int hb = (int)(rest % G.HBT);
    const long tile = rest / G.HBT;
    const int j = (int)(tile % m);
    const int iblk = (int)(tile / m);
    const int r = lane & 31, hh = lane >> 5;
    const int i = iblk * 32 + r;
    h8 hi, lo;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int h = 16 * hb + 8 * hh + t;
        float v = 0.f;
        if (tile < G.NT && h < H && i < Hp) v = W[(long)h * ((long)Hp * m) + (long)i * m + j] * sW;
        if (nt == 3) {
            const _Float16 a = (_Float16)v;
            hi[t] = a;
            lo[t] = (_Float16)(v - (float)a);
        } else {
            hi[t] = __builtin_bit_cast(_Float16, (__bf16)v);
        }
    }
    if (nt == 3) {
        h8* dst = reinterpret_cast<h8*>(pack + X3_HDR) + (tile * G.HBT + hb) * 128 + lane;
        dst[0] = hi;
        dst[64] = lo;
    } else {
        reinterpret_cast<h8*>(pack + X3_HDR)[(tile * G.HBT + hb) * 64 + lane] = hi;
    }
}

__global__ void x3_bwx_pack_kernel(const float* __restrict__ W, int H, int Hp, int m, X3BwxGeom G, long total,
                                   int nparts, float* __restrict__ pack, int nt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) x3_bwx_pack_one(W, H, Hp, m, G, nparts, pack, idx, nt);
}

// ---- all levels, both directions, in two launches (the packs depend on the weights only) ----------------------
#define X3_MAXJOBS 8
struct X3PackJobs {
    const float* W[X3_MAXJOBS];
    float* fwd[X3_MAXJOBS];
    float* bwd[X3_MAXJOBS];
    int H[X3_MAXJOBS], Hp[X3_MAXJOBS], m[X3_MAXJOBS], nparts[X3_MAXJOBS];
    long nW[X3_MAXJOBS], fthreads[X3_MAXJOBS], bthreads[X3_MAXJOBS];
    int nt;
    X3Geom fg[X3_MAXJOBS];
    X3BwxGeom bg[X3_MAXJOBS];
};

__global__ __launch_bounds__(1024) void x3_absmax_multi_kernel(const X3PackJobs J) {
    const int l = blockIdx.y;
    if ((int)blockIdx.x >= J.nparts[l]) return;
    __shared__ float red[16];
    const float* __restrict__ W = J.W[l];
    const long total = J.nW[l], nv = total >> 2;
    const long stride = (long)J.nparts[l] * blockDim.x;
    float v = 0.f;
    const bool vec = (((size_t)W) & 15) == 0;
    if (vec) {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
            const float4 a = reinterpret_cast<const float4*>(W)[i];
            v = fmaxf(fmaxf(v, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
        }
    }
    for (long i = (vec ? nv * 4 : 0) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) v = fmaxf(v, fabsf(W[i]));
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x < 16) {
        v = red[threadIdx.x];
        for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        if (threadIdx.x == 0) {
            if (J.fwd[l]) J.fwd[l][X3_HDR_PART + blockIdx.x] = v;
            if (J.bwd[l]) J.bwd[l][X3_HDR_PART + blockIdx.x] = v;
        }
    }
}

__global__ __launch_bounds__(256) void x3_pack_multi_kernel(const X3PackJobs J) {
    const int l = blockIdx.y >> 1, dir = blockIdx.y & 1;
    const long stride = (long)gridDim.x * blockDim.x;
    if (dir == 0) {
        if (!J.fwd[l]) return;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < J.fthreads[l]; idx += stride)
            x3_fwd_pack_one(J.W[l], J.H[l], J.Hp[l], J.m[l], J.fg[l], J.nparts[l], J.fwd[l], idx, J.nt);
    } else {
        if (!J.bwd[l]) return;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < J.bthreads[l]; idx += stride)
            x3_bwx_pack_one(J.W[l], J.H[l], J.Hp[l], J.m[l], J.bg[l], J.nparts[l], J.bwd[l], idx, J.nt);
    }
}

template <int HBT, int NW, int NT = 3>
__global__ __launch_bounds__(64 * NW, 8 / NW) void cin_bwd_x3_kernel(
    const float* __restrict__ dOut, const float* xp, const float* x0, const float* __restrict__ pack,
    int H, int Hp, int m, long N, int IB, float* dxp, float* dx0, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HBS = HBT > 8 ? 8 : HBT;      // h-blocks per stage
    constexpr int SPT = HBT / HBS;              // stages per tile
    constexpr int FRT = NT == 3 ? 2 : 1;        // 1-KB fragments per h-block (hi, lo | bf16)
    constexpr int STAGE = HBS * FRT * 1024;
    constexpr int FPW = HBS * FRT / NW;
    static_assert((HBS * FRT) % NW == 0, "every wave issues the same number of LDS-DMA pieces");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const long n0 = ((long)blockIdx.x * NW + wave) * 32;
    const long n = n0 + c;
    const bool nok = n < N;
    const long nc = nok ? n : N - 1;
    const float nmask = nok ? 1.f : 0.f;

    const char* wsrc = reinterpret_cast<const char*>(pack + X3_HDR) + lane * 16;
    auto dma_stage = [&](const char* src, int slot_off) {
#pragma unroll
        for (int k = 0; k < FPW; ++k) {
            const int f = wave * FPW + k;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + f * 1024),
                                             (LDS_AS void*)(smem + slot_off + f * 1024), 16, 0, 0);
        }
    };
    dma_stage(wsrc, 0);
    dma_stage(wsrc + STAGE, STAGE);

    float* x0s = reinterpret_cast<float*>(smem + 3 * STAGE) + wave * (m * 32);     // wave-private x0[j][n0..n0+31]
    float* dx0s = reinterpret_cast<float*>(smem + 3 * STAGE) + (NW + wave) * (m * 32);
    {   // 4 rows of x0 in flight per pass (lane half hh takes the odd rows of a pair)
        const long nn = n0 + (lane & 31);
        const long ncl = nn < N ? nn : N - 1;
        for (int j0 = 0; j0 < m; j0 += 8) {
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 2 * k + hh;
                t[k] = x0[(long)(j < m ? j : m - 1) * N + ncl];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 2 * k + hh;
                if (j < m) { x0s[j * 32 + (lane & 31)] = t[k]; dx0s[j * 32 + (lane & 31)] = 0.f; }
            }
        }
    }

    // B operand: dOut[h][n] for all h of this launch, column-scaled and split, in registers
    float dmax = 0.f;
    if constexpr (NT == 3) {
        dmax = x3_col_absmax(dOut + nc, N, hh, H);
        dmax = fmaxf(dmax, __shfl_xor(dmax, 32)) * nmask;
    }
    const float sD = NT == 3 ? x3_pow2_scale(dmax, 15) : 1.f;     // bf16 operands need no range fitting
    h8 bh[HBT], bl[NT == 3 ? HBT : 1];
#pragma unroll
    for (int hb = 0; hb < HBT; ++hb) {
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            const int h = 16 * hb + 8 * hh + 2 * t2;
            const float v0 = dOut[(long)(h < H ? h : H - 1) * N + nc] * ((h < H) ? sD * nmask : 0.f);
            const float v1 = dOut[(long)(h + 1 < H ? h + 1 : H - 1) * N + nc] * ((h + 1 < H) ? sD * nmask : 0.f);
            h2 hi, lo;
            if constexpr (NT == 3) {
                x3_split2(v0, v1, hi, lo);
                bl[hb][2 * t2] = lo.x; bl[hb][2 * t2 + 1] = lo.y;
            } else {
                hi = x3_bf16_pair(v0, v1);
            }
            bh[hb][2 * t2] = hi.x; bh[hb][2 * t2 + 1] = hi.y;
        }
    }
    const float inv = NT == 3 ? (1.f / sD) * pack[1] : 1.f;      // removes both scales from dZ

    int so0 = 0, so1 = STAGE, so2 = 2 * STAGE;   // ring slot of the current stage, +1, +2
    const char* wcur = wsrc;
    for (int iblk = 0; iblk < IB; ++iblk) {
        float xpr[16], dxa[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = iblk * 32 + frag_row(r, hh);
            xpr[r] = xp[(long)(i < Hp ? i : Hp - 1) * N + nc] * ((i < Hp) ? nmask : 0.f);
            dxa[r] = 0.f;
        }
        // One tile = HBT h-blocks of 3 MFMAs into `acc`.  The tile BEFORE it (accumulator `pacc`, column j = pj)
        // is consumed in the shadow of these MFMAs, 16 / HBT registers per h-block: the dZ tile leaves the
        // accumulator as dxp[i] += dZ x0[pj] and dx0[pj] += sum_i dZ xp[i] while the matrix pipe keeps running.
        // (A zero `pacc` makes the first tile of an i-block consume nothing.)
        constexpr int RPH = 16 / HBT;
        auto run_tile = [&](f32x16& acc, const f32x16& pacc, int pj) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float px0 = x0s[pj * 32 + c];
            float psj = 0.f;
#pragma unroll
            for (int st = 0; st < SPT; ++st) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FPW) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                dma_stage(wcur + 2 * STAGE, so2);
                const char* sp = smem + so0 + lane * 16;
#pragma unroll
                for (int hbl = 0; hbl < HBS; ++hbl) {
                    const int hb = st * HBS + hbl;
                    const h8 ah = *reinterpret_cast<const h8*>(sp + (FRT * hbl) * 1024);
                    acc = x3_mfma<NT>(ah, bh[hb], acc);
                    if constexpr (NT == 3) {
                        const h8 al = *reinterpret_cast<const h8*>(sp + (2 * hbl + 1) * 1024);
                        acc = x3_mfma<NT>(ah, bl[hb], acc);
                        acc = x3_mfma<NT>(al, bh[hb], acc);
                    }
#pragma unroll
                    for (int r = hb * RPH; r < (hb + 1) * RPH; ++r) {
                        dxa[r] = fmaf(pacc[r], px0, dxa[r]);
                        psj = fmaf(pacc[r], xpr[r], psj);
                    }
                }
                wcur += STAGE;
                const int t = so0; so0 = so1; so1 = so2; so2 = t;
            }
            psj += __shfl_xor(psj, 32);
            if (hh == 0) dx0s[pj * 32 + c] += psj * inv;
        };
        // acc[r] = dZ[(i = iblk*32 + frag_row(r, hh), j)][n] * sW * sD
        if constexpr (HBT >= 16) {
            // 128 VGPRs of dOut leave no room for a second accumulator: consume each tile right away
            f32x16 acc, zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            for (int j = 0; j < m; ++j) {
                run_tile(acc, zero, 0);
                const float px0 = x0s[j * 32 + c];
                float psj = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dxa[r] = fmaf(acc[r], px0, dxa[r]);
                    psj = fmaf(acc[r], xpr[r], psj);
                }
                psj += __shfl_xor(psj, 32);
                if (hh == 0) dx0s[j * 32 + c] += psj * inv;
            }
        } else {
        f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[r] = 0.f;
        int j = 0;
        for (; j + 2 <= m; j += 2) {
            run_tile(accA, accB, j > 0 ? j - 1 : 0);
            run_tile(accB, accA, j);
        }
        if (j < m) {                                   // odd m: one more tile, then its own consumption below
            run_tile(accA, accB, j - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[r] = accA[r];
        }
        {   // the last tile of the i-block (in accB, column m-1) has no successor to hide behind
            const float px0 = x0s[(m - 1) * 32 + c];
            float psj = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dxa[r] = fmaf(accB[r], px0, dxa[r]);
                psj = fmaf(accB[r], xpr[r], psj);
            }
            psj += __shfl_xor(psj, 32);
            if (hh == 0) dx0s[(m - 1) * 32 + c] += psj * inv;
        }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = iblk * 32 + frag_row(r, hh);
            if (i < Hp && nok) {
                float* d = dxp + (long)i * N + n;
                *d = (flags & XDFM_BWX_SET_DXP) ? dxa[r] * inv : *d + dxa[r] * inv;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const long nn = n0 + (lane & 31);
        const long ncl = nn < N ? nn : N - 1;
        const bool set = (flags & XDFM_BWX_SET_DX0) != 0;
        for (int j0 = 0; j0 < m; j0 += 8) {
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {                       // the read-modify-write's loads back to back
                const int j = j0 + 2 * k + hh;
                t[k] = set ? 0.f : dx0[(long)(j < m ? j : m - 1) * N + ncl];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 2 * k + hh;
                if (j < m && nn < N) dx0[(long)j * N + nn] = t[k] + dx0s[j * 32 + (lane & 31)];
            }
        }
    }
}

size_t x3_bwx_pack_elems(int H, int Hp, int m) {
    const X3BwxGeom g = x3_bwx_geom(H, Hp, m);
    return (size_t)X3_HDR + ((size_t)g.NT * g.HBT + 2 * g.HBS) * (x3_terms() == 3 ? 512 : 256);
}

int x3_bwx_pack(const float* W, int H, int Hp, int m, float* pack, hipStream_t st) {
    const X3BwxGeom g = x3_bwx_geom(H, Hp, m);
    const long nW = (long)H * Hp * m;
    const int nt = x3_terms();
    if (nt == 3) x3_launch_absmax(W, nW, pack, st);
    const long total = ((long)g.NT * g.HBT + 2 * g.HBS) * 64;
    hipLaunchKernelGGL(x3_bwx_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, W, H, Hp, m, g, total,
                       x3_absmax_blocks(nW), pack, nt);
    return xdfm_check_launch("cin_bwd_pack (f16x3)");
}

template <int HBT, int NT>
static int launch_bwx3(const float* dOut, const float* xp, const float* x0, const float* pack, int H, int Hp, int m,
                       long N, const X3BwxGeom& g, float* dxp, float* dx0, int flags, hipStream_t st) {
    constexpr int HBS = HBT > 8 ? 8 : HBT;
    constexpr int FR = HBS * (NT == 3 ? 2 : 1);
    constexpr int NWMAX = FR % 8 == 0 ? 8 : 4;
    static_assert(FR % 4 == 0, "a ring stage is dealt to 4 or 8 waves");
    const size_t lds8 = (size_t)3 * FR * 1024 + (size_t)2 * NWMAX * m * 32 * sizeof(float);
    const size_t lds4 = (size_t)3 * FR * 1024 + (size_t)8 * m * 32 * sizeof(float);
    if (lds4 > 160 * 1024) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_x: m=%d needs %zu B of LDS", m, lds4);
    if (NWMAX == 8 && xdfm_opt(OPT_X3_WAVES) != 4 && N >= 256 * 64 && lds8 <= 160 * 1024)
        hipLaunchKernelGGL((cin_bwd_x3_kernel<HBT, NWMAX, NT>), dim3(ceil_div(N, 32 * NWMAX)), dim3(64 * NWMAX), lds8, st, dOut, xp,
                           x0, pack, H, Hp, m, N, g.IB, dxp, dx0, flags);
    else
        hipLaunchKernelGGL((cin_bwd_x3_kernel<HBT, 4, NT>), dim3(ceil_div(N, 128)), dim3(256), lds4, st, dOut, xp, x0, pack, H, Hp,
                           m, N, g.IB, dxp, dx0, flags);
    return xdfm_check_launch("cin_level_bwd_x (f16x3 / bf16)");
}

int x3_level_bwd_x(const float* dOut, const float* xp, const float* x0, const float* pack, int H, int Hp, int m, long N,
                   float* dxp, float* dx0, int flags, hipStream_t st) {
    const X3BwxGeom g = x3_bwx_geom(H, Hp, m);
    if ((((size_t)pack) & 15) != 0) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_x: packed weights must be 16-byte aligned");
    if (x3_terms() == 1) {
        switch (g.HBT) {
            case 4: return launch_bwx3<4, 1>(dOut, xp, x0, pack, H, Hp, m, N, g, dxp, dx0, flags, st);
            case 8: return launch_bwx3<8, 1>(dOut, xp, x0, pack, H, Hp, m, N, g, dxp, dx0, flags, st);
            case 16: return launch_bwx3<16, 1>(dOut, xp, x0, pack, H, Hp, m, N, g, dxp, dx0, flags, st);
            default: return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_x (bf16): no kernel for H=%d", H);
        }
    }
    switch (g.HBT) {
        case 2: return launch_bwx3<2, 3>(dOut, xp, x0, pack, H, Hp, m, N, g, dxp, dx0, flags, st);
        case 4: return launch_bwx3<4, 3>(dOut, xp, x0, pack, H, Hp, m, N, g, dxp, dx0, flags, st);
        case 8: return launch_bwx3<8, 3>(dOut, xp, x0, pack, H, Hp, m, N, g, dxp, dx0, flags, st);
        default: return launch_bwx3<16, 3>(dOut, xp, x0, pack, H, Hp, m, N, g, dxp, dx0, flags, st);
    }
}

// every level through the f16x3 kernels in both directions?  (then xdfm_cin_pack_all can prepare a whole step)
bool x3_pack_all_usable(int H, int Hp, int m) { return x3_fwd_usable(H, Hp, m) && x3_bwx_usable(H, Hp, m); }

int x3_pack_all(const xdfm_cin_pack_job* jobs, int L, hipStream_t st) {
    X3PackJobs J;
    long maxthreads = 0;
    int maxparts = 0;
    for (int l = 0; l < X3_MAXJOBS; ++l) {
        const xdfm_cin_pack_job& j = jobs[l < L ? l : 0];
        J.W[l] = j.W; J.fwd[l] = l < L ? j.fwd_pack : nullptr; J.bwd[l] = l < L ? j.bwd_pack : nullptr;
        J.H[l] = j.H; J.Hp[l] = j.Hp; J.m[l] = j.m;
        J.nW[l] = (long)j.H * j.Hp * j.m;
        J.nparts[l] = x3_absmax_blocks(J.nW[l]);
        J.fg[l] = x3_fwd_geom(j.H, j.Hp, j.m);
        J.bg[l] = x3_bwx_geom(j.H, j.Hp, j.m);
        J.fthreads[l] = (long)J.fg[l].MB * (J.fg[l].NS + 2) * J.fg[l].MT * 64;
        J.bthreads[l] = ((long)J.bg[l].NT * J.bg[l].HBT + 2 * J.bg[l].HBS) * 64;
        if (l < L) {
            if (J.fthreads[l] > maxthreads) maxthreads = J.fthreads[l];
            if (J.bthreads[l] > maxthreads) maxthreads = J.bthreads[l];
            if (J.nparts[l] > maxparts) maxparts = J.nparts[l];
        }
    }
    J.nt = x3_terms();
    if (J.nt == 3) hipLaunchKernelGGL(x3_absmax_multi_kernel, dim3(maxparts, L), dim3(1024), 0, st, J);
    int gx = ceil_div(maxthreads, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(x3_pack_multi_kernel, dim3(gx, 2 * L), dim3(256), 0, st, J);
    return xdfm_check_launch("cin_pack_all");
}
