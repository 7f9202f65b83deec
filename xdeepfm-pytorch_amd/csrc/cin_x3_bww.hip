// dW (K4b) in f16x3 arithmetic (see cin_x3.hip for the arithmetic).
//   dW[h][(i,j)] = sum_n dOut[h][n] * xp[i][n] * x0[j][n]          autograd of deepctr/layers/interaction.py:218-229
// The contraction runs over n, so the range-fitting scales are per ROW: dOut rows h, x_prev rows i, x0
// rows j (row maxima by one reduction pass).  dOut is split once into fp16 hi / lo planes (it is the A
// operand of every one of the K/32 column tiles); Z = xp*x0 is formed, scaled and split in registers.
//
//   pass 1  x3_rowmax_kernel      partial row maxima of dOut, x_prev, x0
//   pass 2  x3_split_dout_kernel  partial maxima -> scales (header); dOut * sD[h] -> planes [Hpad][NP/32][hi 32 | lo 32] fp16 (zero padded)
//   pass 3  cin_bwd_w_x3_kernel   MFMA; per n-split slabs (same tiling as the fp32 kernel)
//   pass 4  x3_bww_unpack_kernel  ordered sum of the slabs, [h][i*m+j] layout
// Round 3: passes 1 and 2 are what the level's backward already does when it forms dOut -- x3_bwd_prep (below) is
// cin_dout with two more outputs: the fp16 hi / lo planes of dOut and the scales, both per n-SPLIT of the MFMA kernel
// (the split's workgroups contract over the split's columns only, so a scale need not hold beyond them; the MFMA kernel
// removes its split's scales from the accumulators before it stores its slab).  The stand-alone entry point
// (xdfm_cin_level_bwd_w on a dOut that somebody else produced) keeps passes 1 and 2 with one header for all splits.
#include "xdfm_internal.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));      // 16 bytes of MFMA operand (fp16 halves, or bf16 bit patterns)
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

template <int NT>
__device__ __forceinline__ f32x16 x3w_mfma(const h8& a, const h8& b, const f32x16& c) {
    if constexpr (NT == 3) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ float x3w_pow2_scale(float amax, int target) {
    const int E = (int)((__float_as_uint(amax) >> 23) & 0xff);
    int be = 253 + target - E;
    be = be < 1 ? 1 : (be > 253 ? 253 : be);
    return E == 0 ? 1.f : __uint_as_float((unsigned)be << 23);
}

bool x3_bww_usable(const float* dOut, const float* xp, const float* x0, int H, long N) {
    return x3_terms() != 0 && bww_mt(H) == 4 && N % 4 == 0 && N >= 32 &&
           ((((size_t)dOut) | ((size_t)xp) | ((size_t)x0)) & 15) == 0;
}

#define X3_RM_COLS 16384      // columns per block of the row-maximum pass (one block per row was 2.5x slower at N = 65536)
struct X3BwwWs { long hdr, parts, planes, NP, HS; int nbx; };     // element (float) counts of the workspace parts; HS = one header
static inline X3BwwWs x3_bww_ws(const BwwGeom& g, int H, int Hp, int m, long N, int nsplit_max) {
    X3BwwWs w;
    w.NP = round_up(N, 32);
    w.nbx = ceil_div(N, X3_RM_COLS);
    w.HS = round_up((long)g.Hpad + g.IPAD + m, 64);                // scales: dOut rows, x_prev rows, x0 rows
    w.hdr = w.HS * (nsplit_max > 1 ? nsplit_max : 1);              // one header per n-split (x3_bwd_prep); the stand-alone passes fill the first
    w.parts = round_up((long)(H + Hp + m) * w.nbx, 64);            // per-block partial row maxima (stand-alone passes)
    w.planes = (long)g.Hpad * w.NP;
    return w;
}

// ---------------------------------------------------------------------------------------------
// Row maxima without atomics or zero-initialised cells (nothing depends on a memset node inside a captured
// graph): block (bx, row) stores the maximum of its X3_RM_COLS columns in parts[row*nbx + bx]; rows [0, H) are
// dOut, [H, H+Hp) x_prev, [H+Hp, H+Hp+m) x0.  x3_split_dout_kernel reduces the partials to the power-of-two
// scales hdr[0..Hpad) (dOut, < 2^15), hdr[Hpad..Hpad+IPAD) (x_prev, < 2^7), hdr[Hpad+IPAD..) (x0, < 2^7);
// rows outside the matrices get scale 1.
// hdr != NULL (only with gridDim.x == 1: one block covers a whole row): the block writes the row's scale itself
__global__ __launch_bounds__(256) void x3_rowmax_kernel(const float* __restrict__ dOut, const float* __restrict__ xp,
                                                       const float* __restrict__ x0, int H, int Hp, long N,
                                                       float* __restrict__ parts, float* __restrict__ hdr, int Hpad,
                                                       int IPAD) {
    int row = blockIdx.y;
    if (hdr) {                                          // blockIdx.y walks the header slots, padding included
        const int t = blockIdx.y;
        if (t < Hpad) { if (t >= H) { if (threadIdx.x == 0) hdr[t] = 1.f; return; } row = t; }
        else if (t < Hpad + IPAD) { if (t - Hpad >= Hp) { if (threadIdx.x == 0) hdr[t] = 1.f; return; } row = H + (t - Hpad); }
        else row = H + Hp + (t - Hpad - IPAD);
    }
    const float* src = row < H ? dOut + (long)row * N : (row < H + Hp ? xp + (long)(row - H) * N : x0 + (long)(row - H - Hp) * N);
    float v = 0.f;
    const long base = (long)blockIdx.x * X3_RM_COLS;
#pragma unroll 8
    for (int k = 0; k < X3_RM_COLS / 1024; ++k) {
        const long n = base + ((long)k * 256 + threadIdx.x) * 4;
        if (n < N) {                                    // N % 4 == 0
            const float4 a = *reinterpret_cast<const float4*>(src + n);
            v = fmaxf(fmaxf(v, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
        }
    }
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (hdr) {
            if (row < H) hdr[row] = x3w_pow2_scale(mx, 15);
            else if (row < H + Hp) hdr[Hpad + (row - H)] = x3w_pow2_scale(mx, 7);
            else hdr[Hpad + IPAD + (row - H - Hp)] = x3w_pow2_scale(mx, 7);
        } else {
            parts[(long)row * gridDim.x + blockIdx.x] = mx;
        }
    }
}

// one thread = 8 columns of one row: 128-B blocks [hi 32 halves | lo 32 halves] per 32 columns
// parts != NULL: the scales are still partial maxima (x3_rowmax_kernel with several blocks per row).  Every block
// then reduces its own row's nbx partials (uniform loads), and the blockIdx.x == 0 column of blocks also writes the
// header -- its row's dOut scale, plus (thread t of block row r) entry r*256 + t of the x_prev / x0 scales -- for the
// MFMA kernel and the unpack pass (no launch of its own for that).
// nt == 1 (bf16): no scales -- the header is all ones (written here by the blockIdx.x == 0 column) and the hi half of
// a block holds the bf16 bit patterns of dOut; the lo half is not used (the planes keep the fp16 layout: same kernel,
// same LDS image; half of the staged bytes are idle in this mode).
__global__ __launch_bounds__(256) void x3_split_dout_kernel(const float* __restrict__ dOut, int H, long N, long NP,
                                                           float* __restrict__ hdr, char* __restrict__ planes,
                                                           const float* __restrict__ parts, int nbx, int Hp, int m,
                                                           int Hpad, int IPAD, int nt) {
    const int row = blockIdx.y;
    float s_row = 1.f;
    if (nt == 1) {
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) hdr[row] = 1.f;
            const int e = row * 256 + threadIdx.x;
            if (e < IPAD + m) hdr[Hpad + e] = 1.f;
        }
    } else if (parts) {
        if (row < H) {
            float mx = 0.f;
            for (int k = 0; k < nbx; ++k) mx = fmaxf(mx, parts[(long)row * nbx + k]);
            s_row = x3w_pow2_scale(mx, 15);
        }
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) hdr[row] = s_row;
            const int e = row * 256 + threadIdx.x;               // entry of the x_prev | x0 part of the header
            if (e < IPAD + m) {
                const int src = e < IPAD ? (e < Hp ? H + e : -1) : H + Hp + (e - IPAD);
                float mx = 0.f;
                if (src >= 0)
                    for (int k = 0; k < nbx; ++k) mx = fmaxf(mx, parts[(long)src * nbx + k]);
                hdr[Hpad + e] = x3w_pow2_scale(mx, 7);
            }
        }
    } else if (row < H) {
        s_row = hdr[row];
    }
    const long n = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (n >= NP) return;
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = 0.f;
    if (row < H) {
        // both float4 halves unconditionally from clamped addresses (N % 4 == 0, N >= 8 here), masked afterwards:
        // two loads in flight instead of one behind each condition
        const float s = s_row;
        const float* base = dOut + (long)row * N;
        const long n0c = n < N ? n : N - 4, n1c = n + 4 < N ? n + 4 : N - 4;
        const float4 a = *reinterpret_cast<const float4*>(base + n0c);
        const float4 b = *reinterpret_cast<const float4*>(base + n1c);
        const float m0 = n < N ? s : 0.f, m1 = n + 4 < N ? s : 0.f;
        v[0] = a.x * m0; v[1] = a.y * m0; v[2] = a.z * m0; v[3] = a.w * m0;
        v[4] = b.x * m1; v[5] = b.y * m1; v[6] = b.z * m1; v[7] = b.w * m1;
    }
    h8 hi, lo;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (nt == 3) {
            const _Float16 a = (_Float16)v[t];
            hi[t] = a;
            lo[t] = (_Float16)(v[t] - (float)a);
        } else {
            hi[t] = __builtin_bit_cast(_Float16, (__bf16)v[t]);
            lo[t] = (_Float16)0.f;
        }
    }
    char* blk = planes + ((long)row * NP + (n & ~31L)) * 4 + (n & 31) * 2;     // 4 bytes per column and row
    *reinterpret_cast<h8*>(blk) = hi;
    if (nt == 3) *reinterpret_cast<h8*>(blk + 64) = lo;
}

// ---------------------------------------------------------------------------------------------
// LDS image of every staged row: 128 bytes = 8 chunks of 16 B, chunk q of row r at position q ^ ((r >> 1) & 7)
// (r = row index inside its region).  A 16-byte fragment read by lanes r = 0..15 of the same logical
// chunk then covers all 64 banks; the LDS-DMA writes lane-linear, so the XOR goes on the source address.
__device__ __forceinline__ void x3w_dma16(const void* g, void* lds_wave_base) {
    x3_lds_dma16(g, x3_lds_addr(lds_wave_base));
}

// hi / lo halves of the products a0*b0, a1*b1 (see x3_split_prod2 in cin_x3.hip): 5 VALU instructions per pair
__device__ __forceinline__ void x3w_split_prod2(float a0, float b0, float a1, float b1, h2& hi, h2& lo) {
    const f2 z = (f2){a0, a1} * (f2){b0, b1};
    hi = __builtin_convertvector(z, h2);
    const unsigned hbits = __builtin_bit_cast(unsigned, hi);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "v"(b0), "v"(hbits));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "v"(b1), "v"(hbits));
    const f2 r = {r0, r1};
    lo = __builtin_convertvector(r, h2);
}

// wave tile as in cin_bwd_w_dma4_kernel: 32*MT rows of h x (32 i's of block iblk) x JT = 2 values of j
// NW waves per workgroup share the staged dOut chunk: with 8 waves the planes are streamed by half as many
// workgroups and half as many n-splits (slabs) are needed to fill the chip with one resident round
// SYM: level 0 (x_prev is x0) over the folded pair list.  dW[h][(i, j)] == dW[h][(j, i)], so only the pairs i <= j are
// contracted: a wave tile is 32*MT rows of h x JT = 2 COMBINED column tiles; lane r of combined tile t (< m/2) is the
// pair (i = r, j = t) for r <= t, (i = 31-r, j = m-1-t) for r >= 32-m+t (the dX kernel's tiling, cin_x3_bwx_sym.hip), both
// factors read from the staged x0 block.  Slabs are compact, [t][h][r]; x3_bww_unpack_sym_kernel mirrors them into dW.
template <int MT, int NW, int NT = 3, bool SYM = false>
__global__ __launch_bounds__(64 * NW, 8 / NW) void cin_bwd_w_x3_kernel(
    const char* __restrict__ planes, long PB, const float* __restrict__ xp, const float* __restrict__ x0,
    const float* __restrict__ hdr0, long hdr_stride, int Hp, int m, long N, int IB, int JP, int TPH, long n_per_split, int Hpad,
    int IPAD, float* __restrict__ dWt, long slab_stride, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int JT = 2;
    // The gridDim.x workgroups of one n-split stream the same dOut planes.  Workgroups are dealt to the 8 XCDs round
    // robin in launch order, so with the plain (x, y) indexing a split's workgroups sit on 7-8 different XCDs and every
    // L2 fetches the planes for itself (343 MB fetched per launch against 74 MB of operands at level 1 of config 2).
    // Remapped: XCD k runs a contiguous range of the (split, x) list, i.e. whole splits.
    int bx = blockIdx.x, by = blockIdx.y;
    if (xcd_remap) {
        const int total = gridDim.x * gridDim.y;
        const int L = blockIdx.y * gridDim.x + blockIdx.x, k = L & 7, slot = L >> 3;
        const int q = total >> 3, r = total & 7;
        const int Lr = (k < r ? k * (q + 1) : r * (q + 1) + (k - r) * q) + slot;
        by = Lr / (int)gridDim.x;
        bx = Lr - by * (int)gridDim.x;
    }
    const float* __restrict__ hdr = hdr0 + (long)by * hdr_stride;     // the scales of this split's columns (stride 0: one header)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wt0 = bx * NW;
    const int hg = wt0 / TPH;
    const int tin = wt0 + wave - hg * TPH;
    const bool active = tin < JP * IB;           // SYM: IB == 1, JP = pairs of combined tiles
    const int jp = active ? tin / IB : 0;
    const int iblk = active ? tin - jp * IB : 0;
    const long NP = PB >> 2;
    const long n_begin = (long)by * n_per_split;
    const long n_end = (n_begin + n_per_split < NP) ? n_begin + n_per_split : NP;   // multiples of 32
    const int nch = (int)((n_end - n_begin) / BWW_NC);

    constexpr int DROWS = 32 * MT;
    constexpr int WROWS = 32 + 8;                       // x_prev block + one 8-row group holding the x0 rows
    constexpr int BUF = (DROWS + NW * WROWS) * 128;     // bytes
    constexpr int ND_D = DROWS / (8 * NW);              // dOut-plane DMA instructions per wave (8 rows each)
    constexpr int NDMA = ND_D + 4 + (SYM ? 0 : 1);     // SYM: no x0 row group, both factors come from the x_prev (= x0) block
    static_assert(DROWS % (8 * NW) == 0, "every wave stages the same number of dOut rows");

    const int lrow = lane >> 3, pc = lane & 7;
    const int drow0 = wave * (DROWS / NW);
    auto swz = [](int row) { return (row >> 1) & 7; };

    auto dma_k = [&](int k, long nc0, int buf) {
        char* base = smem + buf * BUF;
        if (k < ND_D) {
            const int row = drow0 + 8 * k + lrow;
            const int q = pc ^ swz(row);
            x3w_dma16(planes + (long)(hg * DROWS + row) * PB + nc0 * 4 + q * 16, base + (drow0 + 8 * k) * 128);
        } else if (k < ND_D + 4) {
            const int kk = k - ND_D;
            char* wb = base + (DROWS + wave * WROWS) * 128;
            const int row = 8 * kk + lrow;
            int i = iblk * 32 + row;
            i = i < Hp ? i : Hp - 1;
            long col = nc0 + (pc ^ swz(row)) * 4;
            col = col < N - 4 ? col : N - 4;            // columns >= N meet zero rows of the dOut planes
            x3w_dma16(xp + (long)i * N + col, wb + (8 * kk) * 128);
        } else {
            char* wb = base + (DROWS + wave * WROWS + 32) * 128;
            int j = jp * JT + (lrow < JT ? lrow : JT - 1);     // rows >= JT of the group are never read
            j = j < m ? j : m - 1;
            long col = nc0 + pc * 4;                    // rows 0, 1 of the group: swizzle 0
            col = col < N - 4 ? col : N - 4;
            x3w_dma16(x0 + (long)j * N + col, wb);
        }
    };

    // per-lane scale of the B operand: sxp[i] * sx0[j]
    float fz[JT];
    int ri[JT], rj[JT];                          // SYM: the lane's pair in each combined tile (rows of the staged x0 block)
    if constexpr (SYM) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int t = jp * JT + jt;
            int i = -1, j = 0;
            if (active && t < m / 2) {
                if (r <= t) { i = r; j = t; }
                else if (31 - r <= m - 1 - t) { i = 31 - r; j = m - 1 - t; }
            }
            fz[jt] = i >= 0 ? hdr[Hpad + i] * hdr[Hpad + IPAD + j] : 0.f;
            ri[jt] = i >= 0 ? i : 0;
            rj[jt] = j;
        }
    } else {
        const int i = iblk * 32 + r;
        const float sx = hdr[Hpad + (i < IPAD ? i : 0)];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int j = jp * JT + jt;
            fz[jt] = (i < Hp && j < m) ? sx * hdr[Hpad + IPAD + (j < m ? j : 0)] : 0.f;
            ri[jt] = rj[jt] = 0;
        }
    }

    f32x16 acc[MT][JT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][jt][q] = 0.f;

    const int sr = swz(r);
    auto compute = [&](int buf, long next_nc0, int next_buf) {
        const char* dS = smem + buf * BUF;
        const char* xS = dS + (DROWS + wave * WROWS) * 128;
        const char* zS = xS + 32 * 128;
        const bool has_next = next_nc0 >= 0;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            // B operand: Z[(i_r, j)][n = 16 nb + 8 hh + t], t < 8
            const int q0 = nb * 4 + 2 * hh;
            float4 xa, xb;
            if constexpr (!SYM) {
                xa = *reinterpret_cast<const float4*>(xS + (r * 8 + (q0 ^ sr)) * 16);
                xb = *reinterpret_cast<const float4*>(xS + (r * 8 + ((q0 + 1) ^ sr)) * 16);
            }
            h8 bh[JT], bl[JT];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                float4 za, zb;
                if constexpr (SYM) {
                    const int si = swz(ri[jt]), sj = swz(rj[jt]);
                    xa = *reinterpret_cast<const float4*>(xS + (ri[jt] * 8 + (q0 ^ si)) * 16);
                    xb = *reinterpret_cast<const float4*>(xS + (ri[jt] * 8 + ((q0 + 1) ^ si)) * 16);
                    za = *reinterpret_cast<const float4*>(xS + (rj[jt] * 8 + (q0 ^ sj)) * 16);
                    zb = *reinterpret_cast<const float4*>(xS + (rj[jt] * 8 + ((q0 + 1) ^ sj)) * 16);
                } else {
                    za = *reinterpret_cast<const float4*>(zS + (jt * 8 + q0) * 16);
                    zb = *reinterpret_cast<const float4*>(zS + (jt * 8 + q0 + 1) * 16);
                }
                const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                const float zv[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    h2 hi, lo = h2{0, 0};
                    const f2 xs = (f2){xv[2 * t2], xv[2 * t2 + 1]} * (f2){fz[jt], fz[jt]};     // row scales: exact
                    if constexpr (NT == 3) {
                        x3w_split_prod2(xs.x, zv[2 * t2], xs.y, zv[2 * t2 + 1], hi, lo);
                    } else {
                        const f2 z = xs * (f2){zv[2 * t2], zv[2 * t2 + 1]};
                        hi = __builtin_bit_cast(h2, __builtin_convertvector(z, bf2));
                    }
                    bh[jt][2 * t2] = hi.x; bh[jt][2 * t2 + 1] = hi.y;
                    bl[jt][2 * t2] = lo.x; bl[jt][2 * t2 + 1] = lo.y;
                }
            }
            if (has_next) {
#pragma unroll
                for (int k = nb * ((NDMA + 1) / 2); k < (nb + 1) * ((NDMA + 1) / 2); ++k)
                    if (k < NDMA) dma_k(k, next_nc0, next_buf);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int qa = nb * 2 + hh;
                const h8 ah = *reinterpret_cast<const h8*>(dS + ((mt * 32 + r) * 8 + (qa ^ sr)) * 16);
                if constexpr (NT == 3) {
                    const h8 al = *reinterpret_cast<const h8*>(dS + ((mt * 32 + r) * 8 + ((qa + 4) ^ sr)) * 16);
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        acc[mt][jt] = x3w_mfma<3>(ah, bh[jt], acc[mt][jt]);
                        acc[mt][jt] = x3w_mfma<3>(ah, bl[jt], acc[mt][jt]);
                        acc[mt][jt] = x3w_mfma<3>(al, bh[jt], acc[mt][jt]);
                    }
                } else {
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) acc[mt][jt] = x3w_mfma<1>(ah, bh[jt], acc[mt][jt]);
                }
            }
        }
    };

    if (nch > 0) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) dma_k(k, n_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int ch = 0; ch < nch; ++ch) {
        const long nxt = (ch + 1 < nch) ? n_begin + (long)(ch + 1) * BWW_NC : -1;
        compute(ch & 1, nxt, (ch + 1) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // the split's scales leave the accumulators here (exact: powers of two, one factor at a time), so the slabs of
    // different splits -- each with scales of its own -- add up as they are.  The row group's 32 * MT inverse dOut scales
    // go through LDS (the staging buffers are free now): one round trip instead of one per accumulator register.
    float* inv_sd = reinterpret_cast<float*>(smem);
    for (int k = threadIdx.x; k < 32 * MT; k += 64 * NW)
        inv_sd[k] = __uint_as_float(0x7f000000u - __float_as_uint(hdr[hg * 32 * MT + k]));       // 1 / 2^e, exact
    __syncthreads();
    if (!active) return;
    const int i = iblk * 32 + r;
    float* __restrict__ dst = dWt + (long)by * slab_stride;
    float isx[JT], isz[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        int ii, jj;
        if constexpr (SYM) { ii = ri[jt]; jj = rj[jt]; }
        else { ii = i < IPAD ? i : 0; jj = jp * JT + jt; jj = jj < m ? jj : 0; }
        isx[jt] = __uint_as_float(0x7f000000u - __float_as_uint(hdr[Hpad + ii]));
        isz[jt] = __uint_as_float(0x7f000000u - __float_as_uint(hdr[Hpad + IPAD + jj]));
    }
    if constexpr (SYM) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int h = hg * 32 * MT + mt * 32 + frag_row(q, hh);
                const float isd = inv_sd[mt * 32 + frag_row(q, hh)];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const int t = jp * JT + jt;
                    if (t < m / 2) dst[((long)t * Hpad + h) * 32 + r] = acc[mt][jt][q] * isd * isx[jt] * isz[jt];
                }
            }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int h = hg * 32 * MT + mt * 32 + frag_row(q, hh);
            const float isd = inv_sd[mt * 32 + frag_row(q, hh)];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const int j = jp * JT + jt;
                if (i < Hp && j < m) dst[((long)j * Hpad + h) * IPAD + i] = acc[mt][jt][q] * isd * isx[jt] * isz[jt];
            }
        }
}

// dW[h][i*m+j] = sum over n-splits of dWt[split][j][h][i], in split order (the kernel removed each split's scales)
__global__ void x3_bww_unpack_kernel(const float* __restrict__ dWt, int H, int Hp, int m,
                                     int Hpad, int IPAD, int nslab, long slab_stride, float* __restrict__ dW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)m * Hpad * IPAD;
    if (idx >= total) return;
    const int i = (int)(idx % IPAD);
    const long jh = idx / IPAD;
    const int h = (int)(jh % Hpad), j = (int)(jh / Hpad);
    if (i >= Hp || h >= H) return;
    float acc = 0.f;
    int k = 0;
    for (; k + 8 <= nslab; k += 8) {                    // 8 slab loads in flight; the sum keeps its fixed order
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = dWt[(long)(k + q) * slab_stride + idx];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += t[q];
    }
    for (; k < nslab; ++k) acc += dWt[(long)k * slab_stride + idx];
    dW[(long)h * ((long)Hp * m) + (long)i * m + j] = acc;
}

// folded level 0: dW[h][i*m+j] = dW[h][j*m+i] = sum over n-splits of slab[split][t][h][r], (i, j) the pair of lane r in
// combined tile t.  One thread per slab element (coalesced reads of every split), two 4-byte stores.
__global__ void x3_bww_unpack_sym_kernel(const float* __restrict__ dWt, int H, int m,
                                         int Hpad, int nslab, long slab_stride, float* __restrict__ dW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)(m / 2) * Hpad * 32) return;
    const int r = (int)(idx & 31);
    const int h = (int)((idx >> 5) % Hpad), t = (int)((idx >> 5) / Hpad);
    int i, j;
    if (r <= t) { i = r; j = t; }
    else if (31 - r <= m - 1 - t) { i = 31 - r; j = m - 1 - t; }
    else return;
    if (h >= H) return;
    float acc = 0.f;
    int k = 0;
    for (; k + 8 <= nslab; k += 8) {                    // 8 slab loads in flight; the sum keeps its fixed order
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = dWt[(long)(k + q) * slab_stride + idx];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    for (; k < nslab; ++k) acc += dWt[(long)k * slab_stride + idx];
    const float v = acc;
    float* __restrict__ row = dW + (long)h * ((long)m * m);
    row[i * m + j] = v;
    if (i != j) row[j * m + i] = v;
}

// ---------------------------------------------------------------------------------------------
static inline int x3_bww_waves() { return xdfm_opt(OPT_X3_WAVES) == 4 ? 4 : 8; }

// tiling of bww_geometry with NW wave tiles per workgroup and one resident round of workgroups
// (2 x 4 waves or 1 x 8 waves per CU)
static BwwGeom x3_bww_geometry(int H, int Hp, int m, long N, int NW) {
    BwwGeom g = bww_geometry(H, Hp, m, N);
    g.TPH = (int)round_up((long)g.JP * g.IB, NW);
    g.gx = g.HG * g.TPH / NW;
    int nsplit = xdfm_opt(OPT_BWW_NSPLIT);
    const int max_split = ceil_div(N, BWW_NC);
    const int slots = NW == 8 ? 256 : 512;
    if (nsplit <= 0) nsplit = slots / g.gx > 0 ? slots / g.gx : 1;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit > 65535) nsplit = 65535;
    g.n_per_split = round_up(ceil_div(N, nsplit), BWW_NC);
    g.nsplit = ceil_div(N, g.n_per_split);
    return g;
}

// folded level 0: JP = pairs of combined tiles, compact slabs [m/2][Hpad][32]
static BwwGeom x3_bww_geometry_sym(int H, int m, long N, int NW) {
    BwwGeom g = bww_geometry(H, m, m, N);
    g.JP = ceil_div(m / 2, BWW_JT);
    g.TPH = (int)round_up((long)g.JP, NW);
    g.gx = g.HG * g.TPH / NW;
    int nsplit = xdfm_opt(OPT_BWW_NSPLIT);
    const int max_split = ceil_div(N, BWW_NC);
    const int slots = NW == 8 ? 256 : 512;
    if (nsplit <= 0) nsplit = slots / g.gx > 0 ? slots / g.gx : 1;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit > 65535) nsplit = 65535;
    g.n_per_split = round_up(ceil_div(N, nsplit), BWW_NC);
    g.nsplit = ceil_div(N, g.n_per_split);
    g.slab = (long)(m / 2) * g.Hpad * 32;
    return g;
}
static bool x3_bww_has_sym(int Hp, int m) { return Hp == m && x3_sym_m(m); }

// n-splits of the geometry a launch will use, and the larger of the two a level with Hp == m may use (whether the level
// IS level 0 -- x_prev == x0 -- is known at launch only): the header region of the workspace holds one header per split
static int x3_bww_nsplit_max(int H, int Hp, int m, long N, int NW) {
    int n = x3_bww_geometry(H, Hp, m, N, NW).nsplit;
    if (x3_bww_has_sym(Hp, m)) {
        const int ns = x3_bww_geometry_sym(H, m, N, NW).nsplit;
        if (ns > n) n = ns;
    }
    return n;
}

size_t x3_bww_ws_elems(int H, int Hp, int m, long N) {
    const int NW = x3_bww_waves();
    const BwwGeom g = x3_bww_geometry(H, Hp, m, N, NW);
    const X3BwwWs w = x3_bww_ws(g, H, Hp, m, N, x3_bww_nsplit_max(H, Hp, m, N, NW));
    size_t slabs = (size_t)g.slab * g.nsplit;
    if (x3_bww_has_sym(Hp, m)) {
        const BwwGeom gs = x3_bww_geometry_sym(H, m, N, NW);
        const size_t s2 = (size_t)gs.slab * gs.nsplit;
        if (s2 > slabs) slabs = s2;
    }
    return (size_t)w.hdr + (size_t)w.parts + (size_t)w.planes + slabs;
}

// ---------------------------------------------------------------------------------------------
// x3_bwd_prep: cin_dout (dOut = act'(A) * (dHid + dDirect), per-block sums for dbias; cin_bwd.hip) that ALSO leaves what
// the f16x3 / bf16 dW kernel needs -- the fp16 hi / lo planes of dOut and the row scales of dOut, x_prev and x0 -- in the
// dW workspace, instead of two more passes over dOut (x3_rowmax_kernel + x3_split_dout_kernel re-read 2 x 67 MB per
// level at config 2).  The scales are per n-SPLIT of the MFMA kernel: a block owns one row and KS whole splits
// (<= 4096 columns, values kept in registers between the maximum and the split), the maxima of its splits form in LDS
// with integer atomics (order-independent), and nothing crosses a block.
// grid.y walks the header slots: [0, Hpad) dOut rows (rows >= H are zero planes with scale 1), [Hpad, Hpad + IPAD)
// x_prev rows, then the m rows of x0 (maxima only).
// ---------------------------------------------------------------------------------------------
#define X3_PREP_COLS 4096
// exact n / d for 0 <= n < 2^31 with a multiplication: M = ceil(2^(31 + s) / d), s = ceil(log2 d) (64-bit integer
// division costs ~100 VALU instructions on this hardware, and the kernel below would do 12 of them per 32 columns)
struct X3Magic { unsigned M; int s; };
static X3Magic x3_magic(long d) {
    X3Magic k;
    k.s = 0;
    while ((1L << k.s) < d) ++k.s;
    k.M = (unsigned)((((unsigned long long)1 << (31 + k.s)) + (unsigned long long)d - 1) / (unsigned long long)d);
    if (d == 1) k.M = 0x80000000u;
    return k;
}
__device__ __forceinline__ int x3_div(unsigned n, X3Magic k) { return (int)(((unsigned long long)n * k.M) >> (31 + k.s)); }

template <int NT>
__global__ __launch_bounds__(256) void x3_bwd_prep_kernel(
    const float* __restrict__ A, const unsigned* __restrict__ mask, long mask_ld, int H, long N, X3Magic divD, int act,
    const float* __restrict__ dHid, int hid0, int hid_rows,
    const float* __restrict__ dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
    float* __restrict__ dOut, float* __restrict__ slots, float* __restrict__ dbias, unsigned* __restrict__ ticket,
    const float* __restrict__ xp, const float* __restrict__ x0, int Hp, int m, int Hpad, int IPAD,
    int n_per_split, X3Magic divS, int KS, int nsplit, float* __restrict__ hdr2, long HS, char* __restrict__ planes, long NP) {
    __shared__ unsigned seg_max[X3_PREP_COLS / 32 + 1];
    __shared__ float wsum[4];
    const int y = blockIdx.y;
    const int s0 = blockIdx.x * KS;                                   // first split of this block
    const int ns = s0 + KS <= nsplit ? KS : nsplit - s0;              // its splits
    const long c0 = (long)s0 * n_per_split;
    long c1l = c0 + (long)ns * n_per_split;                           // plane columns [c0, c1), multiples of 32
    c1l = c1l < NP ? c1l : NP;
    const int span = (int)(c1l - c0);                                 // columns of this block (offsets below are relative to c0)
    const int nreal = (int)((N < c1l ? N : c1l) - c0);                // ... of which these exist (N % 4 == 0); <= 0: none
    for (int k = threadIdx.x; k < ns; k += 256) seg_max[k] = 0u;
    __syncthreads();
    const bool reg = span <= X3_PREP_COLS;                            // values stay in registers between the two phases
    const bool is_d = y < Hpad, is_xp = !is_d && y < Hpad + IPAD;
    const int row = is_d ? y : (is_xp ? y - Hpad : y - Hpad - IPAD);
    const bool real = is_d ? row < H : (is_xp ? row < Hp : true);
    // the ReLU mask of a dOut row: the sign bits the forward left (X3FwdEpi.mask), or the saved output itself
    const bool use_mask = is_d && mask != nullptr;
    const unsigned* __restrict__ mrow = use_mask ? mask + (real ? row : 0) : nullptr;        // word of chunk q: mrow[q * mask_ld]
    const float* __restrict__ src = (is_d ? (use_mask ? x0 : A + (long)(real ? row : 0) * N)
                                          : (is_xp ? xp + (long)(real ? row : 0) * N : x0 + (long)row * N)) + c0;
    const bool has_hid = is_d && real && dHid && row >= hid0 && row < hid0 + hid_rows;
    const bool has_dir = is_d && real && dDir && row >= dir0 && row < dir0 + dir_rows;
    const float* __restrict__ hrow = has_hid ? dHid + (long)(row - hid0) * N + c0 : nullptr;
    const float* __restrict__ drow = (has_dir && dir_mode == 1) ? dDir + (long)(dir_off + row - dir0) * N + c0 : nullptr;
    const float* __restrict__ dres = (has_dir && dir_mode == 0) ? dDir + dir_off + (row - dir0) : nullptr;
    float* __restrict__ orow = dOut ? dOut + (long)(is_d && real ? row : 0) * N + c0 : nullptr;     // null: dOut is not materialised
    // the values of 4 columns at offset o (o % 4 == 0, o < nreal): dOut for a dOut row, the operand itself for x_prev / x0
    auto keep4 = [&](int o) -> float4 {                  // 1.0 where the level's output was > 0 (4 columns at offset o)
        const unsigned w = mrow[((c0 + o) >> 5) * mask_ld] >> ((c0 + o) & 31);
        return make_float4((w & 1u) ? 1.f : 0.f, (w & 2u) ? 1.f : 0.f, (w & 4u) ? 1.f : 0.f, (w & 8u) ? 1.f : 0.f);
    };
    auto value = [&](int o) -> float4 {
        const float4 av = use_mask ? keep4(o) : *reinterpret_cast<const float4*>(src + o);
        if (!is_d) return av;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hrow) g = *reinterpret_cast<const float4*>(hrow + o);
        if (drow) { const float4 dv = *reinterpret_cast<const float4*>(drow + o); g.x += dv.x; g.y += dv.y; g.z += dv.z; g.w += dv.w; }
        if (dres) { const float rr = dres[(long)x3_div((unsigned)(c0 + o), divD) * lddir]; g.x += rr; g.y += rr; g.z += rr; g.w += rr; }   // one example
        if (act == XDFM_ACT_RELU) {
            if (!(av.x > 0.f)) g.x = 0.f;
            if (!(av.y > 0.f)) g.y = 0.f;
            if (!(av.z > 0.f)) g.z = 0.f;
            if (!(av.w > 0.f)) g.w = 0.f;
        }
        return g;
    };
    // a thread owns 8 consecutive columns per iteration (two float4): 16-byte stores into the hi and the lo half of a plane block
    float4 gv[2][2];
    float part = 0.f;
    const int iters = (span + 2047) / 2048;
    auto relu_mask = [&](float4& g, const float4& av) {
        if (act == XDFM_ACT_RELU) {
            if (!(av.x > 0.f)) g.x = 0.f;
            if (!(av.y > 0.f)) g.y = 0.f;
            if (!(av.z > 0.f)) g.z = 0.f;
            if (!(av.w > 0.f)) g.w = 0.f;
        }
    };
    // dbias share and split maximum of 8 columns; the dOut store too when the values do not stay in registers (otherwise it
    // waits until the block has drawn its ticket: a block's ticket waits for its outstanding stores, see `finish`)
    auto account = [&](int o, const float4& g0, const float4& g1) {
        if (is_d) {
            if (!reg && orow) {
                *reinterpret_cast<float4*>(orow + o) = g0;
                if (o + 4 < nreal) *reinterpret_cast<float4*>(orow + o + 4) = g1;
            }
            part += ((g0.x + g0.y) + (g0.z + g0.w)) + ((g1.x + g1.y) + (g1.z + g1.w));
        }
        if (NT == 3) {
            const float mx = fmaxf(fmaxf(fmaxf(fabsf(g0.x), fabsf(g0.y)), fmaxf(fabsf(g0.z), fabsf(g0.w))),
                                   fmaxf(fmaxf(fabsf(g1.x), fabsf(g1.y)), fmaxf(fabsf(g1.z), fabsf(g1.w))));
            // 8 | 32 | n_per_split: the 8 columns lie in one split.  Non-negative floats order like their bits.
            atomicMax(&seg_max[x3_div((unsigned)o, divS)], __float_as_uint(mx));
        }
    };
    // `account` when the whole wave is inside the block's columns: the wave's 512 columns usually lie in ONE split, and
    // then one lane speaks for all 64 (64 LDS atomics on one address take 64 turns)
    auto account_wave = [&](int o, const float4& g0, const float4& g1) {          // (only in the register path)
        if (is_d) part += ((g0.x + g0.y) + (g0.z + g0.w)) + ((g1.x + g1.y) + (g1.z + g1.w));
        if (NT == 3) {
            float mx = fmaxf(fmaxf(fmaxf(fabsf(g0.x), fabsf(g0.y)), fmaxf(fabsf(g0.z), fabsf(g0.w))),
                             fmaxf(fmaxf(fabsf(g1.x), fabsf(g1.y)), fmaxf(fabsf(g1.z), fabsf(g1.w))));
            const int seg = x3_div((unsigned)o, divS);
            const int seg_first = __builtin_amdgcn_readfirstlane(seg);
            if (__builtin_amdgcn_ballot_w64(seg != seg_first) == 0) {
                for (int k = 32; k > 0; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k));
                if ((threadIdx.x & 63) == 0) atomicMax(&seg_max[seg], __float_as_uint(mx));
            } else {
                atomicMax(&seg_max[seg], __float_as_uint(mx));
            }
        }
    };
    if (real && (NT == 3 || is_d)) {
        if (reg) {
            // <= 2 iterations: EVERY load of the block's columns goes out before the first one is used -- unconditional
            // loads from clamped offsets, masked afterwards (behind per-load conditions hipcc waits for each load in turn:
            // one load in flight per thread, 2.7 TB/s); the operand choices (dHid / dDir rows) are block-uniform
            int o[2], oc[2][2];
            float msk[2][2];
            float4 av[2][2], g[2][2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                o[it] = (it * 256 + (int)threadIdx.x) * 8;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bool ok = it < iters && o[it] + 4 * q < nreal;
                    oc[it][q] = ok ? o[it] + 4 * q : 0;
                    msk[it][q] = ok ? 1.f : 0.f;
                    av[it][q] = use_mask ? keep4(oc[it][q]) : *reinterpret_cast<const float4*>(src + oc[it][q]);
                    g[it][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (hrow) {
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int q = 0; q < 2; ++q) g[it][q] = *reinterpret_cast<const float4*>(hrow + oc[it][q]);
            }
            if (drow) {
                float4 dv[2][2];
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int q = 0; q < 2; ++q) dv[it][q] = *reinterpret_cast<const float4*>(drow + oc[it][q]);
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { g[it][q].x += dv[it][q].x; g[it][q].y += dv[it][q].y; g[it][q].z += dv[it][q].z; g[it][q].w += dv[it][q].w; }
            }
            if (dres) {
                float rr[2][2];
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int q = 0; q < 2; ++q) rr[it][q] = dres[(long)x3_div((unsigned)(c0 + oc[it][q]), divD) * lddir];     // 4 columns: one example
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { g[it][q].x += rr[it][q]; g[it][q].y += rr[it][q]; g[it][q].z += rr[it][q]; g[it][q].w += rr[it][q]; }
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (is_d) relu_mask(g[it][q], av[it][q]);
                    else g[it][q] = av[it][q];
                    g[it][q].x *= msk[it][q]; g[it][q].y *= msk[it][q]; g[it][q].z *= msk[it][q]; g[it][q].w *= msk[it][q];
                    gv[it][q] = g[it][q];
                }
                const int wave_o = (it * 256 + ((int)threadIdx.x & ~63)) * 8;          // first column of this wave in this iteration
                if (it < iters && wave_o + 512 <= nreal) account_wave(o[it], g[it][0], g[it][1]);
                else if (it < iters && o[it] < nreal) account(o[it], g[it][0], g[it][1]);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                const int o = (it * 256 + (int)threadIdx.x) * 8;
                float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
                if (o < nreal) g0 = value(o);
                if (o + 4 < nreal) g1 = value(o + 4);
                if (o < nreal) account(o, g0, g1);
            }
        }
    }
    if (is_d && real) {                                               // dbias: this block's share, summed in block order later
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    }
    // LDS only crosses this barrier (maxima, wave sums): the dOut stores above stay in flight behind it (__syncthreads
    // would wait for every one of them to be acknowledged before the first plane store goes out)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (is_d && real && threadIdx.x == 0) xdfm_publish(&slots[(long)row * gridDim.x + blockIdx.x], (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    // the splits' scales: dOut rows < 2^15, x_prev / x0 rows < 2^7 (|Z| < 2^14); 1 for padding rows and for bf16
    const int target = is_d ? 15 : 7;
    for (int k = threadIdx.x; k < ns; k += 256) {
        const float sc = (NT == 3 && real) ? x3w_pow2_scale(__uint_as_float(seg_max[k]), target) : 1.f;
        hdr2[(long)(s0 + k) * HS + y] = sc;
    }
    // dbias: the block that draws the last ticket of a ROW adds the row's slots, in block order (no launch of its own for
    // that).  The ticket is drawn HERE, before the block's dOut / plane stores go out: drawing it waits for the block's
    // outstanding memory operations, and behind 48 KB of stores that wait kept every block on its CU for microseconds.
    if (ticket && is_d && real && xdfm_last_block_done(ticket + row, gridDim.x) && threadIdx.x == 0) {     // one ticket per dOut row
        float sacc = 0.f;
        for (int k = 0; k < (int)gridDim.x; ++k) sacc += xdfm_peer(slots + (long)row * gridDim.x + k);
        dbias[row] += sacc;
    }
    if (!is_d) return;
    // planes of this row and these columns: [hi 32 halves | lo 32 halves] per 32 columns, zero beyond N and for rows >= H
    char* __restrict__ prow = planes + ((long)y * NP + c0) * 4;
    const int iters2 = reg ? 2 : iters;              // register path: a fixed, unrolled pair of iterations (static register indices:
    //                                                  a run-time index into gv turns the array into 16 KB of LDS)
#pragma unroll 2
    for (int it = 0; it < iters2; ++it) {
        const int o = (it * 256 + (int)threadIdx.x) * 8;
        if (o >= span) continue;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
        if (real) {
            if (reg) { g0 = it == 0 ? gv[0][0] : gv[1][0]; g1 = it == 0 ? gv[0][1] : gv[1][1]; }
            else { if (o < nreal) g0 = value(o); if (o + 4 < nreal) g1 = value(o + 4); }
        }
        float sc = 1.f;
        if (NT == 3 && real) sc = x3w_pow2_scale(__uint_as_float(seg_max[x3_div((unsigned)o, divS)]), 15);
        const float v[8] = {g0.x * sc, g0.y * sc, g0.z * sc, g0.w * sc, g1.x * sc, g1.y * sc, g1.z * sc, g1.w * sc};
        h8 hi, lo;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (NT == 3) {
                const _Float16 a = (_Float16)v[t];
                hi[t] = a;
                lo[t] = (_Float16)(v[t] - (float)a);
            } else {
                hi[t] = __builtin_bit_cast(_Float16, (__bf16)v[t]);
                lo[t] = (_Float16)0.f;
            }
        }
        char* blk = prow + (long)(o & ~31) * 4 + (o & 31) * 2;
        *reinterpret_cast<h8*>(blk) = hi;
        if (NT == 3) *reinterpret_cast<h8*>(blk + 64) = lo;
        if (reg && real && o < nreal && orow) {                       // the fp32 dOut the dX kernel reads (unless it forms dOut itself)
            *reinterpret_cast<float4*>(orow + o) = g0;
            if (o + 4 < nreal) *reinterpret_cast<float4*>(orow + o + 4) = g1;
        }
    }
}

// geometry of the dW launch that will follow for these arguments (x3_level_bwd_w takes the same decisions)
static BwwGeom x3_bww_launch_geometry(const float* xp, const float* x0, int H, int Hp, int m, long N, int NW, bool* sym) {
    *sym = xp == x0 && x3_bww_has_sym(Hp, m) && xdfm_opt(OPT_X3_SYM) != 0;
    return *sym ? x3_bww_geometry_sym(H, m, N, NW) : x3_bww_geometry(H, Hp, m, N, NW);
}

// cin_dout + the dW kernel's operands (see x3_bwd_prep_kernel).  slots: H * x3_bwd_prep_slots() floats of dbias partials
// (the caller adds them up in block order).  ws: the dW workspace.
int x3_bwd_prep_blocks(bool xp_is_x0, int H, int Hp, int m, long N) {
    const bool sym = xp_is_x0 && x3_bww_has_sym(Hp, m) && xdfm_opt(OPT_X3_SYM) != 0;
    const BwwGeom g = sym ? x3_bww_geometry_sym(H, m, N, x3_bww_waves()) : x3_bww_geometry(H, Hp, m, N, x3_bww_waves());
    const int KS = g.n_per_split >= X3_PREP_COLS ? 1 : (int)(X3_PREP_COLS / g.n_per_split);
    return ceil_div(g.nsplit, KS);
}

int x3_bwd_prep(const float* A, const unsigned* mask, long mask_ld, int H, long N, int D, int act, const float* dHid, int hid0, int hid_rows, const float* dDir,
                int dir_mode, long lddir, int dir_off, int dir0, int dir_rows, float* dOut, float* slots, float* dbias,
                unsigned* ticket, const float* xp, const float* x0, int Hp, int m, float* ws, hipStream_t st) {
    const int NW = x3_bww_waves();
    bool sym;
    const BwwGeom g = x3_bww_launch_geometry(xp, x0, H, Hp, m, N, NW, &sym);
    const X3BwwWs w = x3_bww_ws(x3_bww_geometry(H, Hp, m, N, NW), H, Hp, m, N, x3_bww_nsplit_max(H, Hp, m, N, NW));
    if ((((size_t)ws) & 15) != 0) return xdfm_fail(XDFM_ERR_INVALID, "cin_bwd_prep: workspace must be 16-byte aligned");
    float* hdr2 = ws;
    char* planes = reinterpret_cast<char*>(ws + w.hdr + w.parts);
    const int KS = g.n_per_split >= X3_PREP_COLS ? 1 : (int)(X3_PREP_COLS / g.n_per_split);
    const dim3 grid(ceil_div(g.nsplit, KS), g.Hpad + g.IPAD + m);
    XDFM_REQUIRE(grid.y <= 65535, "cin_bwd_prep: %d header rows", (int)grid.y);
    XDFM_REQUIRE(N < (1L << 31) && g.n_per_split < (1L << 30), "cin_bwd_prep: N = %ld columns", N);
    const X3Magic divD = x3_magic(D), divS = x3_magic(g.n_per_split);
    if (x3_terms() == 3)
        hipLaunchKernelGGL(x3_bwd_prep_kernel<3>, grid, dim3(256), 0, st, A, mask, mask_ld, H, N, divD, act, dHid, hid0, hid_rows, dDir, dir_mode, lddir,
                           dir_off, dir0, dir_rows, dOut, slots, dbias, ticket, xp, x0, Hp, m, g.Hpad, g.IPAD, (int)g.n_per_split, divS, KS, g.nsplit, hdr2,
                           w.HS, planes, w.NP);
    else
        hipLaunchKernelGGL(x3_bwd_prep_kernel<1>, grid, dim3(256), 0, st, A, mask, mask_ld, H, N, divD, act, dHid, hid0, hid_rows, dDir, dir_mode, lddir,
                           dir_off, dir0, dir_rows, dOut, slots, dbias, ticket, xp, x0, Hp, m, g.Hpad, g.IPAD, (int)g.n_per_split, divS, KS, g.nsplit, hdr2,
                           w.HS, planes, w.NP);
    return xdfm_check_launch("cin_bwd_prep (f16x3 / bf16)");
}

// prepared: x3_bwd_prep has filled the planes and the per-split headers of `ws` for these very arguments
int x3_level_bwd_w(const float* dOut, const float* xp, const float* x0, int H, int Hp, int m, long N, float* ws,
                   float* dW, bool prepared, hipStream_t st) {
    const int NW = x3_bww_waves();
    bool sym;
    const BwwGeom g = x3_bww_launch_geometry(xp, x0, H, Hp, m, N, NW, &sym);
    // header(s), partial maxima, planes: the same in both tilings
    const X3BwwWs w = x3_bww_ws(x3_bww_geometry(H, Hp, m, N, NW), H, Hp, m, N, x3_bww_nsplit_max(H, Hp, m, N, NW));
    xdfm_opt_note(OPT_LAST_SYM, (xdfm_opt(OPT_LAST_SYM) & ~4) | (sym ? 4 : 0));
    if ((((size_t)ws) & 15) != 0) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_w: workspace must be 16-byte aligned");
    float* hdr = ws;
    float* parts = ws + w.hdr;
    char* planes = reinterpret_cast<char*>(ws + w.hdr + w.parts);
    float* slabs = ws + w.hdr + w.parts + w.planes;
    // "bww_phase" lets a profiler bracket the MFMA kernel alone: the three phases of one call, run in order with
    // the same arguments, are the whole call
    const int phase = xdfm_opt(OPT_BWW_PHASE);
    int rc = XDFM_OK;
    const int nt = x3_terms();
    if (!prepared && (phase == 0 || phase == 1)) {
    if (nt == 1) {                      // bf16: no row scales, no maxima pass
    } else if (w.nbx == 1) {            // one block covers a row: the maxima pass writes the scales itself
        hipLaunchKernelGGL(x3_rowmax_kernel, dim3(1, g.Hpad + g.IPAD + m), dim3(256), 0, st, dOut, xp, x0, H, Hp, N, parts,
                           hdr, g.Hpad, g.IPAD);
    } else {
        hipLaunchKernelGGL(x3_rowmax_kernel, dim3(w.nbx, H + Hp + m), dim3(256), 0, st, dOut, xp, x0, H, Hp, N, parts,
                           (float*)nullptr, g.Hpad, g.IPAD);
    }
    // several blocks per row: the split pass turns the partial maxima into the scales on its way (g.Hpad >= 128
    // block rows x 256 threads cover the IPAD + m further header entries with room to spare)
    XDFM_REQUIRE((long)g.Hpad * 256 >= g.IPAD + m, "cin_level_bwd_w: header larger than the split pass's grid");
    hipLaunchKernelGGL(x3_split_dout_kernel, dim3(ceil_div(w.NP, 2048), g.Hpad), dim3(256), 0, st, dOut, H, N, w.NP, hdr,
                       planes, w.nbx == 1 ? (const float*)nullptr : (const float*)parts, w.nbx, Hp, m, g.Hpad, g.IPAD, nt);
    rc = xdfm_check_launch("cin_level_bwd_w split");
    if (rc) return rc;
    }
    if (phase == 0 || phase == 2) {
    const size_t lds = (size_t)2 * (32 * 4 + NW * (32 + 8)) * 128;
    const int xcd = xdfm_opt(OPT_BWW_XCD) != 0 ? 1 : 0;
    const long hstride = prepared ? w.HS : 0;          // per-split headers, or the one header of the stand-alone passes
#define BWW_LAUNCH(NWV, NTV) \
    hipLaunchKernelGGL((cin_bwd_w_x3_kernel<4, NWV, NTV>), dim3(g.gx, g.nsplit), dim3(64 * NWV), lds, st, planes, w.NP * 4, xp, \
                       x0, hdr, hstride, Hp, m, N, g.IB, g.JP, g.TPH, g.n_per_split, g.Hpad, g.IPAD, slabs, g.slab, xcd)
#define BWW_LAUNCH_SYM(NWV, NTV) \
    hipLaunchKernelGGL((cin_bwd_w_x3_kernel<4, NWV, NTV, true>), dim3(g.gx, g.nsplit), dim3(64 * NWV), lds, st, planes, w.NP * 4, xp, \
                       x0, hdr, hstride, Hp, m, N, g.IB, g.JP, g.TPH, g.n_per_split, g.Hpad, g.IPAD, slabs, g.slab, xcd)
    if (sym) {
        if (NW == 8) { if (nt == 3) BWW_LAUNCH_SYM(8, 3); else BWW_LAUNCH_SYM(8, 1); }
        else { if (nt == 3) BWW_LAUNCH_SYM(4, 3); else BWW_LAUNCH_SYM(4, 1); }
    } else if (NW == 8) { if (nt == 3) BWW_LAUNCH(8, 3); else BWW_LAUNCH(8, 1); }
    else { if (nt == 3) BWW_LAUNCH(4, 3); else BWW_LAUNCH(4, 1); }
#undef BWW_LAUNCH_SYM
#undef BWW_LAUNCH
    rc = xdfm_check_launch("cin_level_bwd_w (f16x3)");
    if (rc) return rc;
    }
    if ((phase == 0 || phase == 3) && sym) {
    hipLaunchKernelGGL(x3_bww_unpack_sym_kernel, dim3(ceil_div(g.slab, 256)), dim3(256), 0, st, slabs, H, m,
                       g.Hpad, g.nsplit, g.slab, dW);
    rc = xdfm_check_launch("cin_level_bwd_w unpack (folded level 0)");
    } else if (phase == 0 || phase == 3) {
    hipLaunchKernelGGL(x3_bww_unpack_kernel, dim3(ceil_div(g.slab, 256)), dim3(256), 0, st, slabs, H, Hp, m, g.Hpad,
                       g.IPAD, g.nsplit, g.slab, dW);
    rc = xdfm_check_launch("cin_level_bwd_w unpack (f16x3)");
    }
    return rc;
}
