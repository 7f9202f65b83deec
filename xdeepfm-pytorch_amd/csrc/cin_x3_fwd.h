// Forward kernel of the f16x3 / bf16 CIN arithmetic (see cin_x3.hip) as a header: the instances for the
// BASELINE field counts (m = 22, 26) are compiled in cin_x3.hip, those for every other even m <= 40 in
// cin_x3_fwd_ma.hip / cin_x3_fwd_mb.hip (two more translation units, so the build stays parallel).
#pragma once
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
    const f2 z = (f2){a0, a1} * (f2){b0, b1};        // (two v_mul_f32 instead of the packed multiply: measured, no difference)
    hi = __builtin_convertvector(z, h2);
    const unsigned hbits = __builtin_bit_cast(unsigned, hi);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "v"(b0), "v"(hbits));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "v"(b1), "v"(hbits));
    const f2 r = {r0, r1};
    lo = __builtin_convertvector(r, h2);
}

// The lane's 8 * HBT values of dOut (rows h = 16 hb + 8 hh + t of this launch, column nc), zero for rows >= H: read from
// the materialised tensor, or formed from its sources (X3DoutSrc, xdfm_internal.h).  All loads are unconditional (clamped
// addresses) and issued before the first use.  lane, c = lane & 31, hh = lane >> 5; n0 = the wave's first column.
template <int HBT>
__device__ __forceinline__ void x3_load_dout(float (&raw)[8 * HBT], const X3DoutSrc& S, int H, long N, long nc, long n0, int lane,
                                             int c, int hh) {
    if (S.dOut) {
#pragma unroll
        for (int hb = 0; hb < HBT; ++hb)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int h = 16 * hb + 8 * hh + t;
                raw[8 * hb + t] = S.dOut[(long)(h < H ? h : H - 1) * N + nc];
            }
        return;
    }
    // sign bits of this wave's 32 columns: the chunk's words are contiguous over the rows -- lane L takes rows 4L .. 4L+3
    uint4 wv = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (S.mask) {
        const long chunk = n0 >> 5;
        int r4 = 4 * lane;
        r4 = r4 + 4 <= (int)S.mask_ld - S.h0 ? r4 : 0;                       // rows past the level: any words (their values are zeroed)
        wv = *reinterpret_cast<const uint4*>(S.mask + chunk * S.mask_ld + S.h0 + r4);
    }
    // Blocks of 16 rows lie -- for every level whose halves are multiples of 16 -- entirely in the hidden part (dHid) or
    // entirely in the direct-connect part (the pooled gradient): then the 16 loads of a block are a wave-uniform base
    // plus ONE 32-bit lane offset (128 per-lane 64-bit addresses do not fit beside the 128 values); a block that straddles
    // a boundary takes the element-wise path.
    const long bex = nc >> S.logD;
    const unsigned offh = (unsigned)(hh ? 8 * N : 0) + (unsigned)nc;            // element (row + 8 hh, column nc) from row r0 + t
    const unsigned offd = (unsigned)(bex * S.lddir) + (unsigned)(8 * hh);       // pooled gradient of this example, rows + 8 hh
#pragma unroll
    for (int hb = 0; hb < HBT; ++hb) {
        const int r0 = S.h0 + 16 * hb;                                           // wave-uniform
        const bool in_dir = S.dDir && r0 + 16 > S.dir0 && r0 < S.dir0 + S.dir_rows;      // the block touches direct-connect rows
        const bool in_hid = S.dHid && r0 < S.hid_rows;
        const bool all_h = in_hid && r0 + 16 <= S.hid_rows && !in_dir;
        const bool all_d = in_dir && r0 >= S.dir0 && r0 + 16 <= S.dir0 + S.dir_rows && !in_hid;
        if (all_h) {
#pragma unroll
            for (int t = 0; t < 8; ++t) raw[8 * hb + t] = (S.dHid + (long)(r0 + t) * N)[offh];
        } else if (all_d && S.dir_mode == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) raw[8 * hb + t] = (S.dDir + (S.dir_off + r0 + t - S.dir0))[offd];
        } else if (all_d) {
#pragma unroll
            for (int t = 0; t < 8; ++t) raw[8 * hb + t] = (S.dDir + (long)(S.dir_off + r0 + t - S.dir0) * N)[offh];
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int hl = r0 + 8 * hh + t;                                  // row of the level
                float v = 0.f;
                if (S.dHid && hl < S.hid_rows) v = S.dHid[(long)hl * N + nc];
                if (S.dDir && hl >= S.dir0 && hl < S.dir0 + S.dir_rows) {
                    const int r = S.dir_off + hl - S.dir0;
                    v += S.dir_mode == 0 ? S.dDir[bex * S.lddir + r] : S.dDir[(long)r * N + nc];
                }
                raw[8 * hb + t] = v;
            }
        }
    }
    const unsigned w4[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int hb = 0; hb < HBT; ++hb)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // rows of the two lane halves: 16 hb + t and 16 hb + 8 + t; their words sit in lanes (row >> 2), component row & 3
            const int ha = 16 * hb + t, hbr = 16 * hb + 8 + t;
            const unsigned wa = (unsigned)__builtin_amdgcn_readlane((int)w4[ha & 3], ha >> 2);
            const unsigned wb = (unsigned)__builtin_amdgcn_readlane((int)w4[hbr & 3], hbr >> 2);
            const unsigned w = hh ? wb : wa;
            const int h = 16 * hb + 8 * hh + t;
            const bool keep = h < H && (!S.mask || ((w >> c) & 1u));
            raw[8 * hb + t] = keep ? raw[8 * hb + t] : 0.f;
        }
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
// ---------------------------------------------------------------------------------------------
// Forward kernel.  Workgroup = 4 waves = 128 columns x (32*MT rows of one row group mb); a wave owns
// 32 columns and all MT row tiles.  The packed weight fragments of one step (2*MT KB) are shared by the
// four waves through a 3-deep LDS ring filled by 16-byte LDS-DMA two steps ahead (one counted
// s_waitcnt vmcnt + raw s_barrier per step); each wave builds its own B operand (Z hi / lo) in registers.
// NT = MFMA terms per product: 3 (f16x3: hi / lo fp16 halves, range-fitted) or 1 (bf16 operands, no scales)
// EXP: timing experiments compiled for one instance only (results wrong): 1 no B-operand build, 2 no barriers,
// 4 no weight DMA inside the loop (tools/fwd_phases.py)
// SYM: level 0 (x_prev is x0) over the folded pair list of xdfm_internal.h's x3_sym_*: no x_prev rows, one block of
// x3_sym_steps(M) steps, both factors of a product from the x0 registers
template <int M>
struct X3SymTab {
    signed char i[8 * x3_sym_steps(M) + 2], j[8 * x3_sym_steps(M) + 2];
    constexpr X3SymTab() : i(), j() {
        for (int q = 0; q < 8 * x3_sym_steps(M) + 2; ++q) {
            const bool pad = q >= x3_sym_pairs(M);          // padding slots: any field (their weights are 0)
            i[q] = (signed char)(pad ? 0 : x3_sym_i(M, q));
            j[q] = (signed char)(pad ? 0 : x3_sym_j(M, q));
        }
    }
};
template <int M>
__device__ constexpr X3SymTab<M> x3_symtab{};

template <int MT, int M, int NW, int R = X3_RING, int NT = 3, int EXP = 0, bool SYM = false>
__global__ __launch_bounds__(64 * NW, 8 / NW) void cin_fwd_x3_kernel(
    const float* __restrict__ xp, const float* __restrict__ x0, const float* __restrict__ pack,
    const float* __restrict__ bias, int H, int Hp, long N, X3Geom G, int act, float* __restrict__ out, X3FwdEpi E) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MP = M / 2;
    constexpr int FRT = NT == 3 ? 2 : 1;        // 1-KB fragments per row tile (hi, lo | bf16)
    constexpr int FR = FRT * MT;                // 1-KB fragments per step
    constexpr int STEPB = FR * 1024;            // bytes of packed weights per step
    constexpr int NSTEP = SYM ? x3_sym_steps(M) : MP;   // steps of a block (SYM: the whole contraction is one block)
    constexpr int SPS = SYM ? x3_fwd_sps_sym(MT, NT) : x3_fwd_sps(MT, NT, M);  // steps per ring stage: one barrier and one DMA batch per stage
    constexpr int STAGE = SPS * STEPB;          // bytes
    constexpr int FPW = SPS * FR / NW;          // LDS-DMA instructions per wave and stage
    static_assert((SPS * FR) % NW == 0, "every wave issues the same number of LDS-DMA pieces");
    static_assert(SYM || MP >= SPS * (R - 1) + 1, "the next block's x_prev rows must be published before the block's last step");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const long n = ((long)blockIdx.x * NW + wave) * 32 + c;
    const bool nok = n < N;                     // no early exit: every wave feeds the ring and the barriers
    const long nc = nok ? n : N - 1;
    const float nmask = nok ? 1.f : 0.f;
    const int mb = blockIdx.y;
    // timing experiments (xdfm option "dbg" bits 6..11; results become wrong): 1 no stores, 2 one block, 4 = the first
    // lane of every workgroup writes its phase times into out[blockIdx.x * 8 ..] (shader clocks: prologue, loop,
    // epilogue; then the same three in 100 MHz ticks) -- tools/fwd_phases.py
    const int dbg = act >> 8;
    const long tc0 = __builtin_readcyclecounter(), tr0 = __builtin_amdgcn_s_memrealtime();
    act &= 0xff;

    const char* wsrc = reinterpret_cast<const char*>(pack + X3_HDR) + (long)mb * G.NSA * STEPB + lane * 16;
    const unsigned smem_lo = x3_lds_addr(smem);
    auto dma_stage = [&](const char* src, int slot_off) {
#pragma unroll
        for (int k = 0; k < FPW; ++k) {
            const int f = wave * FPW + k;
            x3_lds_dma16(src + f * 1024, smem_lo + slot_off + f * 1024);
        }
    };
    // x_prev rows of a block (8 rows x the workgroup's 32*NW columns) also arrive by LDS-DMA, one block ahead, into
    // two buffers behind the ring: per-lane global loads of them made hipcc drain the whole VMEM queue (s_waitcnt
    // vmcnt(0)) at every block boundary -- the ring's look-ahead with it -- and cost 4 dependent round trips in the
    // prologue.  4-byte pieces: no alignment requirement on N or xp.
    constexpr int XCOLS = 32 * NW;              // columns of the workgroup
    constexpr int XPI = 8 * XCOLS / 64 / NW;    // x_prev DMA instructions per wave and block (256 B each)
    float* xbuf = reinterpret_cast<float*>(smem + R * STAGE);           // [2][8][XCOLS]
    const unsigned xbuf_lo = smem_lo + R * STAGE;
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
            x3_lds_dma4(xp + (long)i * N + col, xbuf_lo + (buf * 8 * XCOLS + e) * 4);
        }
    };
    if constexpr (!SYM) dma_xp(0, 0);
    // the ring runs R - 1 stages ahead of the stage being read (R slots); the packed stream ends with a spare stage
    const char* wlast = wsrc + (long)(G.NSA - SPS) * STEPB;   // last stage that exists (reads past it are clamped to it)
#pragma unroll
    for (int k = 0; k < R - 1; ++k) {
        const char* src = wsrc + (long)k * STAGE;
        dma_stage(src < wlast ? src : wlast, k * STAGE);
    }

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
    } else if (SYM || xp == x0) {
        ap = a0;
    } else {
        ap = x3_col_absmax(xp + nc, N, hh, Hp);
        ap = fmaxf(ap, __shfl_xor(ap, 32)) * nmask;
    }
    // bf16 operands need no range fitting
    const float s0 = NT == 3 ? x3_pow2_scale(a0, 7) : 1.f, sp = NT == 3 ? x3_pow2_scale(ap, 7) : 1.f;
#pragma unroll
    for (int j = 0; j < M; ++j) x0r[j] *= s0;
    if constexpr (SYM) {                        // lane half 1: the fields in reverse order
#pragma unroll
        for (int j = 0; j < M / 2; ++j) {
            const float u = x0r[j], v = x0r[M - 1 - j];
            x0r[j] = hh ? v : u;
            x0r[M - 1 - j] = hh ? u : v;
        }
    }

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
    // B operand (hi, lo) of step s of a block from the block's 4 x_prev values; pairs t2 in [T0, T1) of its 4
    auto build_b = [&](int s, const float (&xv)[4], h8& bh, h8& bl, int T0, int T1) {
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            if (t2 < T0 || t2 >= T1) continue;
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
    // SYM: slots q = 8s + 2*t2, + 1 of the pair list, both factors from the (per lane half ordered) x0 registers
    auto build_b_sym = [&](int s, h8& bh, h8& bl, int T0, int T1) {
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            if (t2 < T0 || t2 >= T1) continue;
            const int q = 8 * s + 2 * t2;
            const int i0 = x3_symtab<M>.i[q], j0 = x3_symtab<M>.j[q], i1 = x3_symtab<M>.i[q + 1], j1 = x3_symtab<M>.j[q + 1];
            h2 hi = h2{0, 0}, lo = h2{0, 0};
            if constexpr (NT == 3) x3_split_prod2(x0r[i0], x0r[j0], x0r[i1], x0r[j1], hi, lo);
            else hi = x3_bf16_pair(x0r[i0] * x0r[j0], x0r[i1] * x0r[j1]);
            bh[2 * t2] = hi.x; bh[2 * t2 + 1] = hi.y;
            bl[2 * t2] = lo.x; bl[2 * t2 + 1] = lo.y;
        }
    };
    // A fragments of row-tile pair `pair` of the stage in ring slot `slot_off`
    constexpr int TG = NT == 3 ? 2 : 4;         // row tiles per region (6 / 4 MFMAs: ~200 / 130 cycles of matrix pipe)
    constexpr int P = MT / TG;                  // regions per step
    static_assert(MT % TG == 0, "a step is a whole number of regions");
    auto load_pair = [&](int slot_off, int pair, h8 (&a)[TG][FRT]) {
        const char* st = smem + slot_off + lane * 16;
#pragma unroll
        for (int k = 0; k < TG; ++k)
#pragma unroll
            for (int f = 0; f < FRT; ++f) a[k][f] = *reinterpret_cast<const h8*>(st + (FRT * (TG * pair + k) + f) * 1024);
    };

    const int nblk = G.FB + (G.TS > 0 ? 1 : 0);
    float xv[4], xn[4];
    // block 0's x_prev rows (issued before the ring's first stages) and stage 0 have landed, for every wave
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * FPW) : "memory");
    __builtin_amdgcn_s_barrier();
    h8 bh, bl;
    if constexpr (SYM) {
        build_b_sym(0, bh, bl, 0, 4);
    } else {
        read_xp(0, G.FB > 0 ? 4 : G.RH, xv);
        build_b(0, xv, bh, bl, 0, 4);
    }
    h8 a[TG][FRT];                              // the row tiles about to be multiplied (loaded one region ahead)
    load_pair(0, 0, a);
    // ring state (wave-uniform): LDS offsets of the step being multiplied and of the one after it (steps sit back to
    // back, R * SPS of them), steps left in the current stage, where the next stage goes and where it comes from
    int rd_off = 0, la_off = R * SPS > 1 ? STEPB : 0, left = SPS - 1;
    int dma_off = (R - 1) * STAGE;
    const char* dma_src = wsrc + (long)(R - 1) * STAGE;
    bool xp_todo = false, xp_young = false;
    // One step = P regions; a region multiplies TG row tiles (MFMAs interleaved: no two consecutive ones share an
    // accumulator) while the NEXT region's fragments are on their way from LDS and a share of the next step's B operand
    // is built on the VALU.  When the look-ahead read of a step's last region is the first to touch a new stage k, the
    // barrier that publishes stage k sits in front of it (two MFMAs ahead of the barrier keep the matrix pipe fed while
    // the waves gather); right behind the barrier the slot of stage k - 2 (consumed: its MFMAs were issued before the
    // barrier; stage k - 1 may still have reads in flight) takes the DMA of stage k + R - 2.
    auto block_steps = [&](int blk, int nsteps_dyn, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const bool has_next = !SYM && blk + 1 < nblk;
        xp_todo = has_next;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (FULL || s < nsteps_dyn) {       // wave-uniform
            h8 nh = bh, nl = bl;                // operand of the step after this one
            const bool more = s + 1 < NSTEP && (FULL || s + 1 < nsteps_dyn);
            const bool wrap = !more && has_next;
            if (wrap) read_xp(blk + 1, blk + 1 < G.FB ? 4 : G.RH, xn);   // landed and published R - 2 stage boundaries after the first one of this block
#pragma unroll
            for (int p = 0; p < P; ++p) {
                h8 an[TG][FRT];
                constexpr int NM = TG * (NT == 3 ? 3 : 1);      // MFMAs of a region, in issue order: f16x3 terms hi*hi, hi*lo, lo*hi
                auto mfmas = [&](int i0, int i1) {
#pragma unroll
                    for (int i = i0; i < i1; ++i) {
                        const int f = i / TG, k = i - f * TG;
                        acc[TG * p + k] = x3_mfma<NT>(a[k][f == 2 ? FRT - 1 : 0], f == 1 ? bl : bh, acc[TG * p + k]);
                    }
                };
                constexpr int TPR = 4 / P > 0 ? 4 / P : 1;      // B pairs built per region
                if (p == P - 1) {
                    mfmas(0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (SPS == 1 || left == 0) {                // the next step opens a new stage
                        // younger than that stage's DMA: R - 3 stages, and (R > 3) the x_prev pieces issued at the
                        // previous boundary in front of that boundary's stage
                        constexpr int YOUNGER = (R - 3) * FPW;
                        if (R > 3 && xp_young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER + XPI) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
                        if constexpr (!(EXP & 2)) __builtin_amdgcn_s_barrier();
                        xp_young = xp_todo;
                        if (xp_todo) { dma_xp(blk + 1, (blk + 1) & 1); xp_todo = false; }
                        if constexpr (!(EXP & 4)) dma_stage(dma_src < wlast ? dma_src : wlast, dma_off);
                        dma_src += STAGE;
                        dma_off = dma_off + STAGE == R * STAGE ? 0 : dma_off + STAGE;
                        left = SPS;
                    }
                    left -= 1;
                    load_pair(la_off, 0, an);
                    mfmas(2, NM);
                } else {
                    load_pair(rd_off, p + 1, an);
                    mfmas(0, NM);
                }
                if constexpr (EXP & 1) {
                } else if (SYM) {
                    if (more) build_b_sym(s + 1, nh, nl, p * TPR, p * TPR + TPR);
                } else if (more) build_b(s + 1, xv, nh, nl, p * TPR, p * TPR + TPR);
                else if (wrap) build_b(0, xn, nh, nl, p * TPR, p * TPR + TPR);
#pragma unroll
                for (int k = 0; k < TG; ++k)
#pragma unroll
                    for (int f = 0; f < FRT; ++f) a[k][f] = an[k][f];
                // issue order inside the region: the look-ahead LDS reads, then MFMAs with the VALU work spread between
                // them (left alone, hipcc queues the VALU behind the last MFMA, where the matrix pipe runs dry)
                __builtin_amdgcn_sched_group_barrier(0x100, TG * FRT, 0);
#pragma unroll
                for (int i = 0; i < (p == P - 1 ? NM - 2 : NM); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
                __builtin_amdgcn_sched_barrier(0);              // regions stay regions: nothing moves across
            }
            bh = nh; bl = nl;
            rd_off = la_off;
            la_off = la_off + STEPB == R * STAGE ? 0 : la_off + STEPB;
            }
        }
        if (has_next) {
#pragma unroll
            for (int il = 0; il < 4; ++il) xv[il] = xn[il];
        }
    };
    const long tc1 = __builtin_readcyclecounter(), tr1 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SYM) {
        block_steps(0, NSTEP, std::true_type{});
    } else {
        for (int blk = 0; blk < ((dbg & 2) ? 1 : G.FB); ++blk) block_steps(blk, MP, std::true_type{});
        if (G.TS > 0) block_steps(G.FB, G.TS, std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead stages must land before LDS is released
    const long tc2 = __builtin_readcyclecounter(), tr2 = __builtin_amdgcn_s_memrealtime();

    // ---- epilogue: remove the scales, bias + activation, FM-layout store --------------------------
    // The bias values of the workgroup's rows sit in LDS since the prologue: their reads count on lgkmcnt, so nothing
    // makes hipcc put an s_waitcnt vmcnt(0) -- which also waits for the previous STORE -- in front of every store
    // (with per-row global loads of the bias it did: 16 write round trips per row tile).
    // X3FwdEpi (xdfm_internal.h): rows >= keep_rows are not stored; rows >= dir0 are summed over the embedding axis into
    // `res`; the ReLU sign bits of every row go to `mask`.
    const float sc = NT == 3 ? pack[1] * (1.f / sp) * (1.f / s0) : 1.f;
    const long wcol = ((long)blockIdx.x * NW + wave) * 32;       // first column of this wave
    const int Dm1 = (1 << E.logD) - 1;
    const bool writer = nok && (c & Dm1) == Dm1;                 // this lane holds an example's last column
    // addresses as a wave-uniform base + ONE 32-bit lane offset that serves every row (16 x MT 64-bit lane addresses do not
    // fit beside the accumulators: hipcc then spills accumulator tiles to scratch)
    const unsigned out_off = (unsigned)(hh ? 4 * N : 0) + (unsigned)nc;                       // element offset from row frag_row(r, 0)
    const unsigned res_lane = (unsigned)((nc >> E.logD) * E.ldres) + (unsigned)(4 * hh);     // example's row of res + the half's 4 rows
    const bool res_vec = E.res && ((E.ldres | (long)E.res_off | (long)E.dir0) & 3) == 0 && (((size_t)E.res) & 15) == 0;
    const int keep_rows = E.keep_rows < H ? E.keep_rows : H;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        __builtin_amdgcn_sched_barrier(0);          // one row tile at a time: 16 store addresses live, not 16 * MT
        const int base = (mb * MT + mt) * 32;       // wave-uniform, like every condition on rows below
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = acc[mt][r] * sc + bias_s[mt * 32 + frag_row(r, hh)];
            if (act == XDFM_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
        }
        // ---- the rows that are kept (all of them, or the hidden half): one lane mask for the whole tile when it is inside
        if (base + 32 <= keep_rows) {
            if (nok && !(dbg & 1)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) (out + (long)(base + frag_row(r, 0)) * N)[out_off] = v[r];
            }
        } else if (base < keep_rows) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (base + frag_row(r, hh) < keep_rows && nok && !(dbg & 1)) (out + (long)(base + frag_row(r, 0)) * N)[out_off] = v[r];
        }
        // ---- ReLU sign bits: a ballot per register = the 32-column words of rows frag_row(r, 0) and + 4; they are dealt
        // to the lanes (lane = row of the tile) and leave with one store
        if (E.mask && !(dbg & 64)) {
            unsigned mw = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned long long bits = __ballot(nok && v[r] > 0.f);
                mw = lane == frag_row(r, 0) ? (unsigned)bits : mw;
                mw = lane == frag_row(r, 0) + 4 ? (unsigned)(bits >> 32) : mw;
            }
            // transposed layout: the words of one 32-column chunk are contiguous over the rows (this store is 128 B per
            // tile, and the dX kernel picks up a chunk's words of 256 rows with one 16-byte load per lane)
            if (lane < 32 && base + lane < H && wcol < N) E.mask[(wcol >> 5) * E.mask_ld + base + lane] = mw;
        }
        // ---- direct-connect sums: over the D lanes of an example, log2(D) DPP steps (row_shr 1, 2, 4, 8 inside a row of 16
        // lanes; fixed order); the lane with an example's last column holds its sum.  Registers 4g .. 4g+3 are four
        // consecutive rows: one 16-byte store per example and group.
        if (E.res && base + 32 > E.dir0 && !(dbg & 128)) {
            const bool inside = base >= E.dir0 && base + 32 <= H && res_vec;
            float* __restrict__ res_t = E.res + (E.res_off + base - E.dir0);       // wave-uniform
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float s4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = v[4 * g4 + q];
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x111, 0xf, 0xf, true));
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x112, 0xf, 0xf, true));
                    if (E.logD > 2) t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x114, 0xf, 0xf, true));
                    if (E.logD > 3) t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x118, 0xf, 0xf, true));
                    // D = 32: lane 15's sum of the first 16 columns joins lane 31's (row_bcast:15 into rows 1 and 3)
                    if (E.logD > 4) t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x142, 0xa, 0xf, true));
                    s4[q] = t;
                }
                float* dst = res_t + 8 * g4 + res_lane;                           // rows base + 8 g4 + 4 hh .. + 3
                if (inside) {
                    if (writer) *reinterpret_cast<float4*>(dst) = make_float4(s4[0], s4[1], s4[2], s4[3]);
                } else if (writer) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = base + 8 * g4 + 4 * hh + q;
                        if (row >= E.dir0 && row < H) dst[q] = s4[q];
                    }
                }
            }
        }
    }
    if ((dbg & 4) && threadIdx.x == 0 && blockIdx.y == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long tc3 = __builtin_readcyclecounter(), tr3 = __builtin_amdgcn_s_memrealtime();
        float* o = out + (long)blockIdx.x * 8;
        o[0] = (float)(tc1 - tc0); o[1] = (float)(tc2 - tc1); o[2] = (float)(tc3 - tc2);
        o[3] = (float)(tr1 - tr0); o[4] = (float)(tr2 - tr1); o[5] = (float)(tr3 - tr2);
    }
}

// NW waves (= 32*NW columns) share one weight ring: 8 waves halve the L2 -> LDS traffic of the ring (every
// workgroup streams the whole packed matrix: 512 x 1.7 MB per launch at config 2 with 4 waves)
template <int MT, int M, int NT, bool SYM = false>
static int launch_x3(const float* xp, const float* x0, const float* pack, const float* bias, int H, int Hp, long N,
                     const X3Geom& g, int act, float* out, const X3FwdEpi& epi, hipStream_t st) {
    constexpr int FR = (NT == 3 ? 2 : 1) * MT;
    constexpr int NWMAX = FR % 8 == 0 ? 8 : 4;
    static_assert(FR % 4 == 0, "a ring stage is dealt to 4 or 8 waves");
    // ring (X3_RING stages of SPS steps) + bias of the workgroup's rows + two x_prev buffers
    constexpr size_t RING = (size_t)X3_RING * (SYM ? x3_fwd_sps_sym(MT, NT) : x3_fwd_sps(MT, NT, M)) * FR * 1024;
    if constexpr (NWMAX == 8) {
        if (xdfm_opt(OPT_X3_WAVES) != 4 && N >= 256 * 64) {
            const dim3 grid(ceil_div(N, 32 * NWMAX), g.MB), block(64 * NWMAX);
            const size_t ldsx = RING + 32 * MT * sizeof(float) + (size_t)2 * 8 * 32 * NWMAX * sizeof(float);
            if constexpr (MT == 4 && M == 26 && NT == 3 && !SYM) {
                const int e = (act >> 8) >> 3;          // dbg bits 8 / 16 / 32 -> EXP 1 / 2 / 4 (7 = all three)
#define X3_EXP_CASE(E) if (e == E) { hipLaunchKernelGGL((cin_fwd_x3_kernel<MT, M, NWMAX, X3_RING, NT, E>), grid, block, ldsx, st, xp, x0, pack, bias, H, Hp, N, g, act, out, epi); return xdfm_check_launch("cin_level_fwd (experiment)"); }
                X3_EXP_CASE(1) X3_EXP_CASE(2) X3_EXP_CASE(3) X3_EXP_CASE(4) X3_EXP_CASE(5) X3_EXP_CASE(6) X3_EXP_CASE(7)
#undef X3_EXP_CASE
            }
            hipLaunchKernelGGL((cin_fwd_x3_kernel<MT, M, NWMAX, X3_RING, NT, 0, SYM>), grid, block, ldsx, st, xp, x0, pack, bias, H, Hp, N, g, act, out, epi);
            return xdfm_check_launch("cin_level_fwd (f16x3 / bf16)");
        }
    }
    const size_t lds4 = RING + 32 * MT * sizeof(float) + (size_t)2 * 8 * 128 * sizeof(float);
    hipLaunchKernelGGL((cin_fwd_x3_kernel<MT, M, 4, X3_RING, NT, 0, SYM>), dim3(ceil_div(N, 128), g.MB), dim3(256), lds4, st, xp, x0, pack,
                       bias, H, Hp, N, g, act, out, epi);
    return xdfm_check_launch("cin_level_fwd (f16x3 / bf16)");
}


// the instances of one field count: MT row tiles per wave by the geometry, nt MFMA terms by the arithmetic
#define X3_FWD_DISPATCH_M(MV)                                                                                          \
    (nt == 3 ? (g.MT == 2   ? launch_x3<2, MV, 3>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st)                        \
                : g.MT == 4 ? launch_x3<4, MV, 3>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st)                        \
                            : launch_x3<8, MV, 3>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st))                       \
             : (g.MT == 4 ? launch_x3<4, MV, 1>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st)                          \
                          : launch_x3<8, MV, 1>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st)))

// level 0 with folded weights (x_prev is x0): same choice of instance, SYM kernels
#define X3_FWD_DISPATCH_SYM(MV)                                                                                        \
    (nt == 3 ? (g.MT == 2   ? launch_x3<2, MV, 3, true>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st)                  \
                : g.MT == 4 ? launch_x3<4, MV, 3, true>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st)                  \
                            : launch_x3<8, MV, 3, true>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st))                 \
             : (g.MT == 4 ? launch_x3<4, MV, 1, true>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st)                    \
                          : launch_x3<8, MV, 1, true>(xp, x0, pack, bias, H, Hp, N, g, act, out, epi, st)))
