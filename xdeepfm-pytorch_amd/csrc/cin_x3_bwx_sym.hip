// dX of CIN level 0 (x_prev is x0) over the folded pair list, f16x3 / bf16 arithmetic.
//   out = sum over pairs i <= j of W'[h][(i, j)] x0[i] x0[j],   W'(i, j) = W(i, j) + W(j, i)  (W(i, i) on the diagonal)
//   dZ'[(i, j)][n] = sum_h W'[h][(i, j)] dOut[h][n]      (never stored)
//   g[i] += dZ' x0[j],  g[j] += dZ' x0[i]                (the diagonal adds twice: d/dx of W' x^2)
// Half the tiles of the full (i, j) grid: tile t (of m/2) carries column j = t in its rows r <= t (i = r) and column
// j = m-1-t in its rows r >= 32-m+t (i = 31-r), 27 of 32 rows at m = 26.  Everything else -- the weight ring, the
// dOut operand kept in registers -- is the plain kernel's (cin_x3.hip).  The whole gradient goes to dx0; dxp
// (== gradient wrt x_prev, the same tensor here) is zero-filled when the caller asks for it to be set.
// deepctr/layers/interaction.py:218-229 at i == 0, backward.
#include "cin_x3_fwd.h"

template <int HBT, int NW, int NT, int M>
__global__ __launch_bounds__(64 * NW, 8 / NW) void cin_bwd_x3_sym_kernel(
    const X3DoutSrc S, const float* __restrict__ x0, const float* __restrict__ pack, int H, long N,
    float* __restrict__ dxp, float* __restrict__ dx0, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(M % 2 == 0 && M < 32, "two columns share the 32 rows of a tile");
    constexpr int NTILE = M / 2;
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

    constexpr int R = X3_BWX_RING;
    const char* wsrc = reinterpret_cast<const char*>(pack + X3_HDR) + lane * 16;
    const char* wlast = wsrc + ((long)NTILE * SPT + 1) * STAGE;   // last stage of the stream (two spare ones close it)
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

    float* x0s = reinterpret_cast<float*>(smem + R * STAGE) + wave * (M * 32);     // wave-private x0[j][n0..n0+31]
    float* gs = reinterpret_cast<float*>(smem + R * STAGE) + (NW + wave) * (M * 32);   // gradient of the same tile
    // (rolled on purpose, like the loop that stores dx0 at the end: unrolled, hipcc keeps all 26 row offsets j*N of
    // this loop alive for that one -- 52 VGPRs through the whole kernel, and spills)
#pragma unroll 1
    for (int j0 = 0; j0 < M; j0 += 8) {          // 4 rows of x0 in flight per pass (lane half hh takes the odd rows)
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + 2 * k + hh;
            t[k] = x0[(long)(j < M ? j : M - 1) * N + nc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + 2 * k + hh;
            if (j < M) { x0s[j * 32 + c] = t[k]; gs[j * 32 + c] = 0.f; }
        }
    }
    if (flags & XDFM_BWX_SET_DXP) {
        for (int i = hh; i < M; i += 2)
            if (nok) dxp[(long)i * N + n] = 0.f;
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
    const float inv = NT == 3 ? (1.f / sD) * pack[1] : 1.f;      // removes both scales from dZ'

    int rd_off = 0, nx_off = STAGE, dma_off = (R - 1) * STAGE;
    const char* dma_src = wsrc + (long)(R - 1) * STAGE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // stage 0 landed; x0s of this wave written (wave-private, in order)
    h8 a[FRT];
#pragma unroll
    for (int f = 0; f < FRT; ++f) a[f] = *reinterpret_cast<const h8*>(smem + lane * 16 + f * 1024);

    // the other factor of this lane's 16 rows in the current tile: x0[row] where the row is on the j = t side
    // (row <= t), x0[31 - row] on the j = m-1-t side; one row changes sides per tile
    float xsel[16], dxa[16], dxb[16];           // dxa: g[row] from the rows <= t, dxb: g[31 - row] from the others
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = frag_row(r, hh);
        const int f = row == 0 ? 0 : 31 - row;
        xsel[r] = x0s[(f < M ? f : 0) * 32 + c] * (f < M ? 1.f : 0.f);
        dxa[r] = 0.f; dxb[r] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int st = 0; st < SPT; ++st) {
#pragma unroll
            for (int hbl = 0; hbl < HBS; ++hbl) {
                const int hb = st * HBS + hbl;
                h8 an[FRT];
                if (hbl == HBS - 1) {
                    // the next h-block opens a new stage: publish it (one MFMA ahead of the barrier); behind the
                    // barrier the slot of the stage before this one takes the DMA of stage + R - 1
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
                for (int f = 0; f < FRT; ++f) a[f] = an[f];
                __builtin_amdgcn_sched_group_barrier(0x100, FRT, 0);      // look-ahead reads first
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // consume the tile: acc[r] = dZ'[(i, j)][n] * sW' * sD of row frag_row(r, hh)
        const float xa = x0s[t * 32 + c], xb = x0s[(M - 1 - t) * 32 + c];
        float psa = 0.f, psb = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int base = (r & 3) + 8 * (r >> 2);        // row of lane half 0; lane half 1 holds base + 4
            if (base + 4 <= t) {                            // both halves on the j = t side
                dxa[r] = fmaf(acc[r], xa, dxa[r]);
                psa = fmaf(acc[r], xsel[r], psa);
            } else if (base > t) {                          // both on the j = m-1-t side
                dxb[r] = fmaf(acc[r], xb, dxb[r]);
                psb = fmaf(acc[r], xsel[r], psb);
            } else {
                dxa[r] = fmaf(acc[r], hh ? 0.f : xa, dxa[r]);
                dxb[r] = fmaf(acc[r], hh ? xb : 0.f, dxb[r]);
                psa = fmaf(acc[r], hh ? 0.f : xsel[r], psa);
                psb = fmaf(acc[r], hh ? xsel[r] : 0.f, psb);
            }
        }
        // pin the row sums here: the branch below splits the block, and hipcc would sink these side-effect-free chains
        // to their use at the end of the kernel -- keeping the accumulators of all m/2 tiles alive (spills)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            asm volatile("" : "+v"(dxa[r]));
            asm volatile("" : "+v"(dxb[r]));
        }
        psa += __shfl_xor(psa, 32);
        psb += __shfl_xor(psb, 32);
        if (hh == 0) {
            gs[t * 32 + c] += psa * inv;
            gs[(M - 1 - t) * 32 + c] += psb * inv;
        }
        if (t + 1 < NTILE) {                                // row t + 1 moves to the j side of the next tile
            const float xn = x0s[(t + 1) * 32 + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) {                  // (indexed by the loop variable only: a computed index keeps the array in scratch)
                const int base = (r & 3) + 8 * (r >> 2);
                if (base == t + 1) xsel[r] = hh == 0 ? xn : xsel[r];
                else if (base + 4 == t + 1) xsel[r] = hh == 1 ? xn : xsel[r];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead stages must land before LDS is released
    // per-row parts: two passes (a row of the first pass is another lane's row of the second)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = frag_row(r, hh);
        if (row < M) gs[row * 32 + c] += dxa[r] * inv;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = 31 - frag_row(r, hh);
        if (f < M) gs[f * 32 + c] += dxb[r] * inv;
    }
    {
        const bool set = (flags & XDFM_BWX_SET_DX0) != 0;
#pragma unroll 1
        for (int j0 = 0; j0 < M; j0 += 8) {
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {                       // the read-modify-write's loads back to back
                const int j = j0 + 2 * k + hh;
                t[k] = set ? 0.f : dx0[(long)(j < M ? j : M - 1) * N + nc];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 2 * k + hh;
                if (j < M && nok) dx0[(long)j * N + n] = t[k] + gs[j * 32 + c];
            }
        }
    }
}

template <int HBT, int NT, int M>
static int launch_bwx3_sym(const X3DoutSrc& dOut, const float* x0, const float* pack, int H, long N, float* dxp, float* dx0,
                           int flags, hipStream_t st) {
    constexpr int HBS = HBT > 8 ? 8 : HBT;
    constexpr int FR = HBS * (NT == 3 ? 2 : 1);
    constexpr int NWMAX = FR % 8 == 0 ? 8 : 4;
    static_assert(FR % 4 == 0, "a ring stage is dealt to 4 or 8 waves");
    const size_t lds8 = (size_t)X3_BWX_RING * FR * 1024 + (size_t)2 * NWMAX * M * 32 * sizeof(float);
    const size_t lds4 = (size_t)X3_BWX_RING * FR * 1024 + (size_t)8 * M * 32 * sizeof(float);
    if (NWMAX == 8 && xdfm_opt(OPT_X3_WAVES) != 4 && N >= 256 * 64)
        hipLaunchKernelGGL((cin_bwd_x3_sym_kernel<HBT, NWMAX, NT, M>), dim3(ceil_div(N, 32 * NWMAX)), dim3(64 * NWMAX), lds8, st,
                           dOut, x0, pack, H, N, dxp, dx0, flags);
    else
        hipLaunchKernelGGL((cin_bwd_x3_sym_kernel<HBT, 4, NT, M>), dim3(ceil_div(N, 128)), dim3(256), lds4, st, dOut, x0, pack,
                           H, N, dxp, dx0, flags);
    return xdfm_check_launch("cin_level_bwd_x (folded level 0)");
}

template <int M>
static int dispatch_bwx3_sym(const X3DoutSrc& dOut, const float* x0, const float* pack, int H, long N, int HBT, int nt,
                             float* dxp, float* dx0, int flags, hipStream_t st) {
    if (nt == 1) {
        switch (HBT) {
            case 4: return launch_bwx3_sym<4, 1, M>(dOut, x0, pack, H, N, dxp, dx0, flags, st);
            case 8: return launch_bwx3_sym<8, 1, M>(dOut, x0, pack, H, N, dxp, dx0, flags, st);
            case 16: return launch_bwx3_sym<16, 1, M>(dOut, x0, pack, H, N, dxp, dx0, flags, st);
            default: return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_x (bf16, folded level 0): no kernel for H=%d", H);
        }
    }
    switch (HBT) {
        case 2: return launch_bwx3_sym<2, 3, M>(dOut, x0, pack, H, N, dxp, dx0, flags, st);
        case 4: return launch_bwx3_sym<4, 3, M>(dOut, x0, pack, H, N, dxp, dx0, flags, st);
        case 8: return launch_bwx3_sym<8, 3, M>(dOut, x0, pack, H, N, dxp, dx0, flags, st);
        default: return launch_bwx3_sym<16, 3, M>(dOut, x0, pack, H, N, dxp, dx0, flags, st);
    }
}

int x3_level_bwd_x_sym(const X3DoutSrc& dOut, const float* x0, const float* pack, int H, int m, long N, int HBT, int nt,
                       float* dxp, float* dx0, int flags, hipStream_t st) {
    if (m == 26) return dispatch_bwx3_sym<26>(dOut, x0, pack, H, N, HBT, nt, dxp, dx0, flags, st);
    if (m == 22) return dispatch_bwx3_sym<22>(dOut, x0, pack, H, N, HBT, nt, dxp, dx0, flags, st);
    return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_x (folded level 0): no kernel for m=%d", m);
}
