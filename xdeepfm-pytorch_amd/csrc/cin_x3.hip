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
#include "cin_x3_fwd.h"


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
    g.SPS = x3_fwd_sps(g.MT, x3_terms() == 1 ? 1 : 3, m);
    g.NSA = (ceil_div(g.NS, g.SPS) + 1) * g.SPS;
    return g;
}

// level 0 over the folded pair list: one block of x3_sym_steps(m) steps, no x_prev rows
X3Geom x3_fwd_geom_sym(int H, int m) {
    X3Geom g = x3_fwd_geom(H, m, m);
    g.MP = x3_sym_steps(m);
    g.FB = 1; g.RH = 0; g.TS = 0;
    g.NS = g.MP;
    g.SPS = x3_fwd_sps_sym(g.MT, x3_terms() == 1 ? 1 : 3);
    g.NSA = (ceil_div(g.NS, g.SPS) + 1) * g.SPS;
    return g;
}

bool x3_fwd_usable(int H, int Hp, int m) {
    (void)Hp;
    const int nt = x3_terms();
    // bf16: one 1-KB fragment per row tile and step, so a ring stage of < 4 row tiles cannot be dealt to 4 waves
    // any even field count 8..40: with an even m both lane halves walk the same compile-time (il, j) pattern; a block of
    // 8 x_prev rows must last m/2 >= ring-depth steps for the next block's rows to be landed AND published in time
    return nt != 0 && m >= 2 * X3_RING && m <= 40 && m % 2 == 0 && H > (nt == 3 ? 32 : 64);
}

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
    const int g = (int)(rest % G.NSA);
    const int mb = (int)(rest / G.NSA);
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
        h8* dst = reinterpret_cast<h8*>(pack + X3_HDR) + (((long)mb * G.NSA + g) * G.MT + mt) * 128 + lane;
        dst[0] = hi;
        dst[64] = lo;
    } else {
        reinterpret_cast<h8*>(pack + X3_HDR)[(((long)mb * G.NSA + g) * G.MT + mt) * 64 + lane] = hi;
    }
}

// folded pack of level 0 (x3_sym_*), same fragment layout with the SYM geometry; its own header: the folded weights
// reach twice the maximum, so their scale is half the one of the plain pack ([0] = sW/2, [1] = 2/sW).  `hdr` = the plain
// pack's header (partial maxima of |W|).
__device__ __forceinline__ void x3_fwd_pack_sym_one(const float* __restrict__ W, int H, int m, const X3Geom& G, int nparts,
                                                    const float* __restrict__ hdr, float* __restrict__ pack, long idx, int nt) {
    const float sW = nt == 3 ? 0.5f * x3_weight_scale(hdr, nparts) : 1.f;
    if (idx == 0) { pack[0] = sW; pack[1] = 1.f / sW; }
    const int lane = (int)(idx & 63);
    long rest = idx >> 6;
    const int mt = (int)(rest % G.MT); rest /= G.MT;
    const int g = (int)(rest % G.NSA);
    const int mb = (int)(rest / G.NSA);
    const int r = lane & 31, hh = lane >> 5;
    const int row = (mb * G.MT + mt) * 32 + r;
    const float* __restrict__ Wr = W + (long)(row < H ? row : 0) * ((long)m * m);
    h8 hi, lo;
    int i = x3_sym_i(m, 8 * g), j = x3_sym_j(m, 8 * g);        // slot 8g; the next ones follow along the list's rows
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int q = 8 * g + t;
        float v = 0.f;
        if (t > 0 && ++j > m - 1 - i) { ++i; j = i; }
        if (g < G.NS && row < H && q < x3_sym_pairs(m)) {
            const int a = hh ? m - 1 - i : i, b = hh ? m - 1 - j : j;
            if (a == b) v = Wr[a * m + a];
            else if (!(hh && i + j == m - 1)) v = Wr[a * m + b] + Wr[b * m + a];
            v *= sW;
        }
        if (nt == 3) {
            const _Float16 x = (_Float16)v;
            hi[t] = x;
            lo[t] = (_Float16)(v - (float)x);
        } else {
            hi[t] = __builtin_bit_cast(_Float16, (__bf16)v);
        }
    }
    if (nt == 3) {
        h8* dst = reinterpret_cast<h8*>(pack + X3_HDR) + (((long)mb * G.NSA + g) * G.MT + mt) * 128 + lane;
        dst[0] = hi;
        dst[64] = lo;
    } else {
        reinterpret_cast<h8*>(pack + X3_HDR)[(((long)mb * G.NSA + g) * G.MT + mt) * 64 + lane] = hi;
    }
}

__global__ void x3_fwd_pack_sym_kernel(const float* __restrict__ W, int H, int m, X3Geom G, int nparts,
                                       const float* __restrict__ hdr, float* __restrict__ pack, int nt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)G.MB * G.NSA * G.MT * 64) x3_fwd_pack_sym_one(W, H, m, G, nparts, hdr, pack, idx, nt);
}

__global__ void x3_fwd_pack_kernel(const float* __restrict__ W, int H, int Hp, int m, X3Geom G, int nparts,
                                   float* __restrict__ pack, int nt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)G.MB * G.NSA * G.MT * 64) x3_fwd_pack_one(W, H, Hp, m, G, nparts, pack, idx, nt);
}

// ---------------------------------------------------------------------------------------------
static size_t x3_fwd_pack_plain_elems(int H, int Hp, int m) {
    const X3Geom g = x3_fwd_geom(H, Hp, m);
    // 2 KB (hi + lo) = 512 floats per (step, row tile); bf16: 1 KB
    return (size_t)X3_HDR + (size_t)g.MB * g.NSA * g.MT * (x3_terms() == 3 ? 512 : 256);
}
// a level whose x_prev can be x0 (Hp == m) carries the folded pack behind the plain one (whether the level IS level 0
// is only known when it is launched: x_prev == x0)
static bool x3_fwd_has_sym(int Hp, int m) { return Hp == m && x3_sym_m(m); }
static size_t x3_fwd_sym_offset(int H, int Hp, int m) { return (size_t)round_up((long)x3_fwd_pack_plain_elems(H, Hp, m), 4); }
size_t x3_fwd_pack_elems(int H, int Hp, int m) {
    if (!x3_fwd_has_sym(Hp, m)) return x3_fwd_pack_plain_elems(H, Hp, m);
    const X3Geom g = x3_fwd_geom_sym(H, m);
    return x3_fwd_sym_offset(H, Hp, m) + (size_t)X3_HDR + (size_t)g.MB * g.NSA * g.MT * (x3_terms() == 3 ? 512 : 256);
}

int x3_fwd_pack(const float* W, int H, int Hp, int m, float* pack, hipStream_t st) {
    const X3Geom g = x3_fwd_geom(H, Hp, m);
    const long total = (long)H * Hp * m;
    const int nt = x3_terms();
    if (nt == 3) x3_launch_absmax(W, total, pack, st);
    const long threads = (long)g.MB * g.NSA * g.MT * 64;
    hipLaunchKernelGGL(x3_fwd_pack_kernel, dim3(ceil_div(threads, 256)), dim3(256), 0, st, W, H, Hp, m, g,
                       x3_absmax_blocks(total), pack, nt);
    if (x3_fwd_has_sym(Hp, m)) {
        const X3Geom gs = x3_fwd_geom_sym(H, m);
        const long ts = (long)gs.MB * gs.NSA * gs.MT * 64;
        hipLaunchKernelGGL(x3_fwd_pack_sym_kernel, dim3(ceil_div(ts, 256)), dim3(256), 0, st, W, H, m, gs,
                           x3_absmax_blocks(total), pack, pack + x3_fwd_sym_offset(H, Hp, m), nt);
    }
    return xdfm_check_launch("cin_fwd_pack (f16x3 / bf16)");
}

int x3_level_fwd(const float* xp, const float* x0, const float* pack, const float* bias, int H, int Hp, int m, long N,
                 int act, float* out, const X3FwdEpi& epi, hipStream_t st) {
    const X3Geom g = x3_fwd_geom(H, Hp, m);
    const int nt = x3_terms();
    act |= ((xdfm_opt(OPT_DBG) >> 6) & 255) << 8;     // timing experiments (results become wrong): see the kernels' `dbg`
    if ((((size_t)pack) & 15) != 0) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_fwd: packed weights must be 16-byte aligned");
    if (nt != 3 && g.MT < 4) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_fwd (bf16): no kernel for MT=%d", g.MT);
    const bool sym = xp == x0 && x3_fwd_has_sym(Hp, m) && xdfm_opt(OPT_X3_SYM) != 0;
    xdfm_opt_note(OPT_LAST_SYM, (xdfm_opt(OPT_LAST_SYM) & ~1) | (sym ? 1 : 0));
    if (sym)
        return x3_level_fwd_sym(x0, pack + x3_fwd_sym_offset(H, Hp, m), bias, H, m, N, x3_fwd_geom_sym(H, m), nt, act, out, epi, st);
    if (m == 26) return X3_FWD_DISPATCH_M(26);
    if (m == 22) return X3_FWD_DISPATCH_M(22);
    if (m < 22) return x3_level_fwd_ma(xp, x0, pack, bias, H, Hp, m, N, g, nt, act, out, epi, st);
    if (m <= 40) return x3_level_fwd_mb(xp, x0, pack, bias, H, Hp, m, N, g, nt, act, out, epi, st);
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

// folded pack of level 0 for the dX kernel of cin_x3_bwx_sym.hip: [tile t < m/2][hb < HBT][p][lane][8 halves]; row r of
// tile t is the pair (i = r, j = t) for r <= t, (i = 31-r, j = m-1-t) for r >= 32-m+t, nothing in between; element
// W'[h = 16*hb + 8*hh + e][(i, j)] * sW / 2 with W'(i, j) = W(i, j) + W(j, i), W(i, i) on the diagonal.  Own header as in
// the forward's folded pack; `hdr` = the plain pack's header (partial maxima of |W|).
__device__ __forceinline__ void x3_bwx_pack_sym_one(const float* __restrict__ W, int H, int m, const X3BwxGeom& G, int nparts,
                                                    const float* __restrict__ hdr, float* __restrict__ pack, long idx, int nt) {
    const float sW = nt == 3 ? 0.5f * x3_weight_scale(hdr, nparts) : 1.f;
    if (idx == 0) { pack[0] = sW; pack[1] = 1.f / sW; }
    const int lane = (int)(idx & 63);
    long rest = idx >> 6;
    const int hb = (int)(rest % G.HBT);
    const int t = (int)(rest / G.HBT);
    const int r = lane & 31, hh = lane >> 5;
    int i = -1, j = 0;
    if (t < m / 2) {
        if (r <= t) { i = r; j = t; }
        else if (31 - r <= m - 1 - t) { i = 31 - r; j = m - 1 - t; }
    }
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int h = 16 * hb + 8 * hh + e;
        float v = 0.f;
        if (i >= 0 && h < H) {
            const float* __restrict__ Wr = W + (long)h * ((long)m * m);
            v = (i == j ? Wr[i * m + i] : Wr[i * m + j] + Wr[j * m + i]) * sW;
        }
        if (nt == 3) {
            const _Float16 x = (_Float16)v;
            hi[e] = x;
            lo[e] = (_Float16)(v - (float)x);
        } else {
            hi[e] = __builtin_bit_cast(_Float16, (__bf16)v);
        }
    }
    if (nt == 3) {
        h8* dst = reinterpret_cast<h8*>(pack + X3_HDR) + ((long)t * G.HBT + hb) * 128 + lane;
        dst[0] = hi;
        dst[64] = lo;
    } else {
        reinterpret_cast<h8*>(pack + X3_HDR)[((long)t * G.HBT + hb) * 64 + lane] = hi;
    }
}

__global__ void x3_bwx_pack_sym_kernel(const float* __restrict__ W, int H, int m, X3BwxGeom G, long total, int nparts,
                                       const float* __restrict__ hdr, float* __restrict__ pack, int nt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) x3_bwx_pack_sym_one(W, H, m, G, nparts, hdr, pack, idx, nt);
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
    float* fsym[X3_MAXJOBS];                 // folded forward pack of a level with Hp == m (null otherwise)
    float* bsym[X3_MAXJOBS];                 // folded dX pack, likewise
    int H[X3_MAXJOBS], Hp[X3_MAXJOBS], m[X3_MAXJOBS], nparts[X3_MAXJOBS];
    long nW[X3_MAXJOBS], fthreads[X3_MAXJOBS], bthreads[X3_MAXJOBS], sthreads[X3_MAXJOBS], bsthreads[X3_MAXJOBS];
    int nt;
    X3Geom fg[X3_MAXJOBS], sg[X3_MAXJOBS];
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
    const int l = blockIdx.y >> 2, dir = blockIdx.y & 3;
    const long stride = (long)gridDim.x * blockDim.x;
    if (dir == 3) {
        if (!J.bsym[l]) return;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < J.bsthreads[l]; idx += stride)
            x3_bwx_pack_sym_one(J.W[l], J.H[l], J.m[l], J.bg[l], J.nparts[l], J.bwd[l], J.bsym[l], idx, J.nt);
    } else if (dir == 2) {
        if (!J.fsym[l]) return;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < J.sthreads[l]; idx += stride)
            x3_fwd_pack_sym_one(J.W[l], J.H[l], J.m[l], J.sg[l], J.nparts[l], J.fwd[l], J.fsym[l], idx, J.nt);
    } else if (dir == 0) {
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
    const X3DoutSrc S, const float* xp, const float* x0, const float* __restrict__ pack,
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

    constexpr int R = X3_BWX_RING;              // ring slots: stage k + R - 2 is in flight while k is read
    const char* wsrc = reinterpret_cast<const char*>(pack + X3_HDR) + lane * 16;
    const char* wlast = wsrc + ((long)IB * m * SPT + 1) * STAGE;   // last stage of the stream (two spare ones close it)
    const unsigned smem_lo = x3_lds_addr(smem);
    auto dma_stage = [&](const char* src, int slot_off) {
#pragma unroll
        for (int k = 0; k < FPW; ++k) {
            const int f = wave * FPW + k;
            x3_lds_dma16(src + f * 1024, smem_lo + slot_off + f * 1024);
        }
    };
#pragma unroll
    for (int k = 0; k < R - 1; ++k) {
        const char* src = wsrc + (long)k * STAGE;
        dma_stage(src < wlast ? src : wlast, k * STAGE);
    }

    float* x0s = reinterpret_cast<float*>(smem + R * STAGE) + wave * (m * 32);     // wave-private x0[j][n0..n0+31]
    float* dx0s = reinterpret_cast<float*>(smem + R * STAGE) + (NW + wave) * (m * 32);
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

    // B operand: dOut[h][n] for all h of this launch, column-scaled and split, in registers.  ONE read of dOut: the lane's
    // 8*HBT values (8*HBT independent loads in flight) are held raw while the column maximum is formed, then scaled
    // and split in place (a separate maximum pass read the wave's columns of dOut twice: 67 of 152 MB fetched per launch
    // at level 0 of config 2)
    h8 bh[HBT], bl[NT == 3 ? HBT : 1];
    float sD = 1.f;
    {
        float raw[8 * HBT];
        x3_load_dout<HBT>(raw, S, H, N, nc, n0, lane, c, hh);
        float dmax = 0.f;
#pragma unroll
        for (int hb = 0; hb < HBT; ++hb)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int h = 16 * hb + 8 * hh + t;
                raw[8 * hb + t] *= (h < H) ? nmask : 0.f;
                dmax = fmaxf(dmax, fabsf(raw[8 * hb + t]));
            }
        if constexpr (NT == 3) {                                      // bf16 operands need no range fitting
            dmax = fmaxf(dmax, __shfl_xor(dmax, 32));
            sD = x3_pow2_scale(dmax, 15);
        }
#pragma unroll
        for (int hb = 0; hb < HBT; ++hb) {
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                const float v0 = raw[8 * hb + 2 * t2] * sD, v1 = raw[8 * hb + 2 * t2 + 1] * sD;
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
    }
    const float inv = NT == 3 ? (1.f / sD) * pack[1] : 1.f;      // removes both scales from dZ

    // ring state (wave-uniform): slot of the stage being read, of the one after it, of the next DMA, and its source
    int rd_off = 0, nx_off = STAGE, dma_off = (R - 1) * STAGE;
    const char* dma_src = wsrc + (long)(R - 1) * STAGE;
    // stage 0 has landed (the dOut loads above are younger than every DMA of the prologue and were waited for);
    // the fragments of an h-block are read one h-block ahead of their MFMAs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    h8 a[FRT];
#pragma unroll
    for (int f = 0; f < FRT; ++f) a[f] = *reinterpret_cast<const h8*>(smem + lane * 16 + f * 1024);
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
#pragma unroll
                for (int hbl = 0; hbl < HBS; ++hbl) {
                    const int hb = st * HBS + hbl;
                    h8 an[FRT];
                    if (hbl == HBS - 1) {
                        // the next h-block opens a new stage: publish it.  One MFMA goes ahead of the barrier (the
                        // matrix pipe has work while the waves gather); behind the barrier the slot of the stage
                        // before this one (consumed: its MFMAs were issued) takes the DMA of stage + R - 1.
                        acc = x3_mfma<NT>(a[0], bh[hb], acc);
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * FPW) : "memory");
                        __builtin_amdgcn_s_barrier();
                        dma_stage(dma_src < wlast ? dma_src : wlast, dma_off);
                        dma_src += STAGE;
                        dma_off = dma_off + STAGE == R * STAGE ? 0 : dma_off + STAGE;
                        const char* sp = smem + nx_off + lane * 16;
#pragma unroll
                        for (int f = 0; f < FRT; ++f) an[f] = *reinterpret_cast<const h8*>(sp + f * 1024);
                        rd_off = nx_off;
                        nx_off = nx_off + STAGE == R * STAGE ? 0 : nx_off + STAGE;
                    } else {
                        const char* sp = smem + rd_off + lane * 16;
#pragma unroll
                        for (int f = 0; f < FRT; ++f) an[f] = *reinterpret_cast<const h8*>(sp + (FRT * (hbl + 1) + f) * 1024);
                        acc = x3_mfma<NT>(a[0], bh[hb], acc);
                    }
                    if constexpr (NT == 3) {
                        acc = x3_mfma<NT>(a[0], bl[hb], acc);
                        acc = x3_mfma<NT>(a[1], bh[hb], acc);
                    }
#pragma unroll
                    for (int r = hb * RPH; r < (hb + 1) * RPH; ++r) {
                        dxa[r] = fmaf(pacc[r], px0, dxa[r]);
                        psj = fmaf(pacc[r], xpr[r], psj);
                    }
#pragma unroll
                    for (int f = 0; f < FRT; ++f) a[f] = an[f];
                    // issue order: look-ahead reads first, the VALU consumption of the previous tile between the MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, FRT, 0);
#pragma unroll
                    for (int i = 0; i < (hbl == HBS - 1 ? (NT == 3 ? 2 : 0) : (NT == 3 ? 3 : 1)); ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
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

static size_t x3_bwx_pack_plain_elems(int H, int Hp, int m) {
    const X3BwxGeom g = x3_bwx_geom(H, Hp, m);
    return (size_t)X3_HDR + ((size_t)g.NT * g.HBT + 2 * g.HBS) * (x3_terms() == 3 ? 512 : 256);
}
// as in the forward: a level with Hp == m carries the folded pack (m/2 tiles + two spare stages) behind the plain one
static size_t x3_bwx_sym_offset(int H, int Hp, int m) { return (size_t)round_up((long)x3_bwx_pack_plain_elems(H, Hp, m), 4); }
static long x3_bwx_sym_threads(const X3BwxGeom& g, int m) { return ((long)(m / 2) * g.HBT + 2 * g.HBS) * 64; }
size_t x3_bwx_pack_elems(int H, int Hp, int m) {
    if (!x3_fwd_has_sym(Hp, m)) return x3_bwx_pack_plain_elems(H, Hp, m);
    const X3BwxGeom g = x3_bwx_geom(H, Hp, m);
    return x3_bwx_sym_offset(H, Hp, m) + (size_t)X3_HDR + (size_t)(x3_bwx_sym_threads(g, m) / 64) * (x3_terms() == 3 ? 512 : 256);
}

int x3_bwx_pack(const float* W, int H, int Hp, int m, float* pack, hipStream_t st) {
    const X3BwxGeom g = x3_bwx_geom(H, Hp, m);
    const long nW = (long)H * Hp * m;
    const int nt = x3_terms();
    if (nt == 3) x3_launch_absmax(W, nW, pack, st);
    const long total = ((long)g.NT * g.HBT + 2 * g.HBS) * 64;
    hipLaunchKernelGGL(x3_bwx_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, W, H, Hp, m, g, total,
                       x3_absmax_blocks(nW), pack, nt);
    if (x3_fwd_has_sym(Hp, m)) {
        const long ts = x3_bwx_sym_threads(g, m);
        hipLaunchKernelGGL(x3_bwx_pack_sym_kernel, dim3(ceil_div(ts, 256)), dim3(256), 0, st, W, H, m, g, ts,
                           x3_absmax_blocks(nW), pack, pack + x3_bwx_sym_offset(H, Hp, m), nt);
    }
    return xdfm_check_launch("cin_bwd_pack (f16x3)");
}

template <int HBT, int NT>
static int launch_bwx3(const X3DoutSrc& dOut, const float* xp, const float* x0, const float* pack, int H, int Hp, int m,
                       long N, const X3BwxGeom& g, float* dxp, float* dx0, int flags, hipStream_t st) {
    constexpr int HBS = HBT > 8 ? 8 : HBT;
    constexpr int FR = HBS * (NT == 3 ? 2 : 1);
    constexpr int NWMAX = FR % 8 == 0 ? 8 : 4;
    static_assert(FR % 4 == 0, "a ring stage is dealt to 4 or 8 waves");
    const size_t lds8 = (size_t)X3_BWX_RING * FR * 1024 + (size_t)2 * NWMAX * m * 32 * sizeof(float);
    const size_t lds4 = (size_t)X3_BWX_RING * FR * 1024 + (size_t)8 * m * 32 * sizeof(float);
    if (lds4 > 160 * 1024) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_x: m=%d needs %zu B of LDS", m, lds4);
    if (NWMAX == 8 && xdfm_opt(OPT_X3_WAVES) != 4 && N >= 256 * 64 && lds8 <= 160 * 1024)
        hipLaunchKernelGGL((cin_bwd_x3_kernel<HBT, NWMAX, NT>), dim3(ceil_div(N, 32 * NWMAX)), dim3(64 * NWMAX), lds8, st, dOut, xp,
                           x0, pack, H, Hp, m, N, g.IB, dxp, dx0, flags);
    else
        hipLaunchKernelGGL((cin_bwd_x3_kernel<HBT, 4, NT>), dim3(ceil_div(N, 128)), dim3(256), lds4, st, dOut, xp, x0, pack, H, Hp,
                           m, N, g.IB, dxp, dx0, flags);
    return xdfm_check_launch("cin_level_bwd_x (f16x3 / bf16)");
}

int x3_level_bwd_x(const X3DoutSrc& dOut, const float* xp, const float* x0, const float* pack, int H, int Hp, int m, long N,
                   float* dxp, float* dx0, int flags, hipStream_t st) {
    const X3BwxGeom g = x3_bwx_geom(H, Hp, m);
    if ((((size_t)pack) & 15) != 0) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_x: packed weights must be 16-byte aligned");
    const bool sym = xp == x0 && x3_fwd_has_sym(Hp, m) && xdfm_opt(OPT_X3_SYM) != 0 && dxp != dx0;
    xdfm_opt_note(OPT_LAST_SYM, (xdfm_opt(OPT_LAST_SYM) & ~2) | (sym ? 2 : 0));
    if (sym)
        return x3_level_bwd_x_sym(dOut, x0, pack + x3_bwx_sym_offset(H, Hp, m), H, m, N, g.HBT, x3_terms(), dxp, dx0, flags, st);
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
        J.fthreads[l] = (long)J.fg[l].MB * J.fg[l].NSA * J.fg[l].MT * 64;
        const bool sym = l < L && j.fwd_pack && x3_fwd_has_sym(j.Hp, j.m);
        J.sg[l] = x3_fwd_geom_sym(j.H, j.m);
        J.fsym[l] = sym ? j.fwd_pack + x3_fwd_sym_offset(j.H, j.Hp, j.m) : nullptr;
        J.sthreads[l] = (long)J.sg[l].MB * J.sg[l].NSA * J.sg[l].MT * 64;
        J.bthreads[l] = ((long)J.bg[l].NT * J.bg[l].HBT + 2 * J.bg[l].HBS) * 64;
        J.bsym[l] = l < L && j.bwd_pack && x3_fwd_has_sym(j.Hp, j.m) ? j.bwd_pack + x3_bwx_sym_offset(j.H, j.Hp, j.m) : nullptr;
        J.bsthreads[l] = x3_bwx_sym_threads(J.bg[l], j.m);
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
    hipLaunchKernelGGL(x3_pack_multi_kernel, dim3(gx, 4 * L), dim3(256), 0, st, J);
    return xdfm_check_launch("cin_pack_all");
}
