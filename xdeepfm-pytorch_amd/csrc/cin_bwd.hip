// K4: backward of one CIN level (autograd of deepctr/layers/interaction.py:218-243).
//
//   dOut[h][n] = act'(A[h][n]) * (dHid[h][n] + dDirect)            cin_dout_kernel (+ dbias)
//   dZ[(i,j)][n] = sum_h W[h][(i,j)] * dOut[h][n]                  cin_bwd_x_kernel  (MFMA, never stored)
//      dxp[i][n] += sum_j dZ[(i,j)][n] * x0[j][n]
//      dx0[j][n] += sum_i dZ[(i,j)][n] * xp[i][n]
//   dW[h][(i,j)] = sum_n dOut[h][n] * xp[i][n] * x0[j][n]          cin_bwd_w_kernel  (MFMA, Z recomputed)
//
// The reference stores Z (1.5 GB per step at B=4096, D=16, cin=(256,128,128)) for these two
// products; here both kernels rebuild their Z / dZ tiles in registers.
#include "xdfm_internal.h"

// =============================================================================================
// dOut + dbias
// =============================================================================================
// VEC = 4: N % 4 == 0 and D % 4 == 0, so a float4 never straddles two examples and every row is
// 16-byte aligned; VEC = 1 is the generic path.  One block = one row h x 1024*VEC.. columns.
template <int VEC>
__global__ __launch_bounds__(256) void cin_dout_kernel(
    const float* __restrict__ A, int H, long N, int D, int act,
    const float* __restrict__ dHid, int hid0, int hid_rows,
    const float* __restrict__ dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
    float* __restrict__ dOut, float* __restrict__ dbias, float* __restrict__ slots, unsigned* __restrict__ ticket,
    const unsigned* __restrict__ mask, long mask_ld) {
    const int h = blockIdx.y;
    const bool has_hid = dHid && h >= hid0 && h < hid0 + hid_rows;
    const bool has_dir = dDir && h >= dir0 && h < dir0 + dir_rows;
    const float* __restrict__ arow = mask ? dOut : A + (long)h * N;        // with sign bits (X3FwdEpi.mask) the saved output is not read
    const unsigned* __restrict__ mrow = mask ? mask + h : nullptr;         // word of chunk q: mrow[q * mask_ld]
    const float* __restrict__ hrow = has_hid ? dHid + (long)(h - hid0) * N : nullptr;
    const float* __restrict__ drow = (has_dir && dir_mode == 1) ? dDir + (long)(dir_off + h - dir0) * N : nullptr;
    const float* __restrict__ dres = (has_dir && dir_mode == 0) ? dDir + dir_off + (h - dir0) : nullptr;
    float* __restrict__ orow = dOut + (long)h * N;
    float part = 0.f;
    const long base = (long)blockIdx.x * (1024 * VEC);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long n = base + ((long)k * 256 + threadIdx.x) * VEC;
        if (n < N) {
            float g[VEC], a[VEC];
            if constexpr (VEC == 4) {
                if (mrow) {
                    const unsigned w = mrow[(n >> 5) * mask_ld] >> (n & 31);
                    a[0] = (w & 1u) ? 1.f : 0.f; a[1] = (w & 2u) ? 1.f : 0.f; a[2] = (w & 4u) ? 1.f : 0.f; a[3] = (w & 8u) ? 1.f : 0.f;
                } else {
                    const float4 av = *reinterpret_cast<const float4*>(arow + n);
                    a[0] = av.x; a[1] = av.y; a[2] = av.z; a[3] = av.w;
                }
                g[0] = g[1] = g[2] = g[3] = 0.f;
                if (hrow) {
                    const float4 hv = *reinterpret_cast<const float4*>(hrow + n);
                    g[0] = hv.x; g[1] = hv.y; g[2] = hv.z; g[3] = hv.w;
                }
                if (drow) {
                    const float4 dv = *reinterpret_cast<const float4*>(drow + n);
                    g[0] += dv.x; g[1] += dv.y; g[2] += dv.z; g[3] += dv.w;
                }
                if (dres) {
                    const float r = dres[(n / D) * lddir];       // the 4 columns belong to one example
                    g[0] += r; g[1] += r; g[2] += r; g[3] += r;
                }
            } else {
                a[0] = mrow ? (((mrow[(n >> 5) * mask_ld] >> (n & 31)) & 1u) ? 1.f : 0.f) : arow[n];
                g[0] = hrow ? hrow[n] : 0.f;
                if (drow) g[0] += drow[n];
                if (dres) g[0] += dres[(n / D) * lddir];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (act == XDFM_ACT_RELU && !(a[e] > 0.f)) g[e] = 0.f;
                part += g[e];
            }
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(orow + n) = make_float4(g[0], g[1], g[2], g[3]);
            else orow[n] = g[0];
        }
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float bsum = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        // with a workspace: one slot per block, added up in block order by cin_dbias_finish_kernel (the same bits on
        // every run); without: a float atomic per block, whose order -- and with it the last bits -- varies
        if (slots) xdfm_publish(&slots[(long)h * gridDim.x + blockIdx.x], bsum);
        else atomicAdd(&dbias[h], bsum);
    }
    // the last block of this ROW adds the row's slots up, in block order (one ticket per row)
    if (slots && ticket && xdfm_last_block_done(ticket + h, gridDim.x) && threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < (int)gridDim.x; ++k) s += xdfm_peer(slots + (long)h * gridDim.x + k);
        dbias[h] += s;
    }
}

__global__ void cin_dbias_finish_kernel(const float* __restrict__ part, int H, int gx, float* __restrict__ dbias) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    float s = 0.f;
    for (int k = 0; k < gx; ++k) s += part[(long)h * gx + k];
    dbias[h] += s;
}

// =============================================================================================
// dX kernel
// =============================================================================================
// Wz[g][lane][4], g = (iblk*m + j)*HS4 + q4: element e of lane (r = lane&31, s = lane>>5) is
// W[h = 2*(4*q4+e) + s][k = (iblk*32 + r)*m + j]  (0 outside the matrix); BWX_PD zero groups appended.
__global__ void cin_bwd_pack_kernel(const float* __restrict__ W, int H, int Hp, int m, int HS4,
                                    long total, long real_groups, float* __restrict__ Wz) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const long g = idx >> 8;
    float v = 0.f;
    if (g < real_groups) {
        const int q4 = (int)(g % HS4);
        const long cj = g / HS4;
        const int j = (int)(cj % m);
        const int iblk = (int)(cj / m);
        const int h = 2 * (4 * q4 + e) + (lane >> 5);
        const int i = iblk * 32 + (lane & 31);
        if (h < H && i < Hp) v = W[(long)h * ((long)Hp * m) + (long)i * m + j];
    }
    Wz[idx] = v;
}

// JB = number of (i-block, j) chains a wave runs interleaved (independent accumulators).  JB = 2 needs an
// even m (chains are paired (j, j+1)) and 32 more VGPRs.
template <int HS4, int JB>
__global__ __launch_bounds__(256, 2) void cin_bwd_x_kernel(
    const float* __restrict__ dOut, const float* xp, const float* x0, const float* __restrict__ Wz,
    int H, int Hp, int m, long N, int IB, float* dxp, float* dx0, int flags) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & 31, s = lane >> 5;
    const long n0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (n0 >= N) return;                       // wave-uniform, no barriers in this kernel
    const long n = n0 + c;
    const bool nok = n < N;
    const long nc = nok ? n : N - 1;
    const float nmask = nok ? 1.f : 0.f;

    float* dx0s = smem + wave * (m * 32);      // wave-private accumulator for dx0[j][n0..n0+31]
    for (int idx = lane; idx < m * 32; idx += 64) dx0s[idx] = 0.f;

    // B operand for the whole kernel: dOut[h = 2q+s][n], q < 4*HS4, kept in registers
    constexpr int NQ = 4 * HS4;
    float breg[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int h = 2 * q + s;
        const float v = dOut[(long)(h < H ? h : H - 1) * N + nc];
        breg[q] = v * ((h < H) ? nmask : 0.f);
    }

    constexpr int R = HS4 >= 4 ? 4 : HS4;      // ring of float4 A groups per chain
    constexpr int PD = R - 1;
    const f32x4* ap = reinterpret_cast<const f32x4*>(Wz) + lane;
    // group index of ring position q of chain b, relative to the first group g of the current chain set:
    // positions past the end of a chain belong to the same chain of the NEXT set (JB chains further on)
    auto gidx = [](int b, int q) { return q < HS4 ? b * HS4 + q : (JB + b) * HS4 + (q - HS4); };
    f32x4 ring[JB][R];
#pragma unroll
    for (int b = 0; b < JB; ++b)
#pragma unroll
        for (int k = 0; k < PD; ++k) ring[b][k] = ap[(long)gidx(b, k) * 64];
    long g = 0;

    for (int iblk = 0; iblk < IB; ++iblk) {
        float xpr[16], dxa[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = iblk * 32 + frag_row(r, s);
            const float v = xp[(long)(i < Hp ? i : Hp - 1) * N + nc];
            xpr[r] = v * ((i < Hp) ? nmask : 0.f);
            dxa[r] = 0.f;
        }
        for (int j = 0; j < m; j += JB) {
            float x0j[JB];                                   // consumed after the chains
#pragma unroll
            for (int b = 0; b < JB; ++b) x0j[b] = x0[(long)(j + b) * N + nc] * nmask;
            f32x16 acc[JB];
#pragma unroll
            for (int b = 0; b < JB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < HS4; ++q4) {
#pragma unroll
                for (int b = 0; b < JB; ++b) ring[b][(q4 + PD) % R] = ap[(g + gidx(b, q4 + PD)) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int b = 0; b < JB; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[b][q4 % R][e], breg[4 * q4 + e], acc[b], 0, 0, 0);
                // pin the order: without this hipcc sinks every ring load down to its first use
                // (one register quad, vmcnt(0) before each group) and the prefetch distance is lost
                __builtin_amdgcn_sched_barrier(0);
            }
            g += JB * HS4;
            // acc[b][r] = dZ[(i = iblk*32 + frag_row(r,s), j + b)][n]
#pragma unroll
            for (int b = 0; b < JB; ++b) {
                float sj = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dxa[r] = fmaf(acc[b][r], x0j[b], dxa[r]);
                    sj = fmaf(acc[b][r], xpr[r], sj);
                }
                sj += __shfl_xor(sj, 32);
                if (s == 0) dx0s[(j + b) * 32 + c] += sj;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = iblk * 32 + frag_row(r, s);
            if (i < Hp && nok) {
                float* d = dxp + (long)i * N + n;
                *d = (flags & XDFM_BWX_SET_DXP) ? dxa[r] : *d + dxa[r];
            }
        }
    }
    // flush the wave's dx0 slice
    for (int idx = lane; idx < m * 32; idx += 64) {
        const int j = idx >> 5;
        const long nn = n0 + (idx & 31);
        if (nn < N) {
            float* d = dx0 + (long)j * N + nn;
            *d = (flags & XDFM_BWX_SET_DX0) ? dx0s[idx] : *d + dx0s[idx];
        }
    }
}

// =============================================================================================
// dW kernel
// =============================================================================================
#define BWW_PITCH 33
// wave tile: 32*MT rows of h  x  (32 i's of block iblk) x JT values of j, accumulated over the
// block's n range; result added (fp32 atomics, 128-B row segments) into
// dWt[j][h][i]  (j < m, h < Hpad, i < IPAD) which cin_bwd_w_unpack_kernel turns into dW[h][i*m+j].
template <int MT, int JT>
__global__ __launch_bounds__(256, 2) void cin_bwd_w_kernel(
    const float* __restrict__ dOut, const float* __restrict__ xp, const float* __restrict__ x0,
    int H, int Hp, int m, long N, int IB, int JP, int TPH, long n_per_split, int Hpad, int IPAD,
    float* __restrict__ dWt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 31, s = lane >> 5;
    const int wt0 = blockIdx.x * 4;
    const int hg = wt0 / TPH;                       // same for the 4 waves (TPH % 4 == 0)
    const int tin = wt0 + wave - hg * TPH;
    const bool active = tin < JP * IB;
    const int jp = active ? tin / IB : 0;
    const int iblk = active ? tin - jp * IB : 0;
    const long n_begin = (long)blockIdx.y * n_per_split;
    const long n_end = (n_begin + n_per_split < N) ? n_begin + n_per_split : N;

    float* dOutS = smem;                                            // [32*MT][PITCH]
    float* xpS = smem + 32 * MT * BWW_PITCH + wave * ((32 + JT) * BWW_PITCH);   // [32][PITCH]
    float* x0S = xpS + 32 * BWW_PITCH;                              // [JT][PITCH]

    f32x16 acc[MT][JT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][jt][r] = 0.f;

    for (long nc0 = n_begin; nc0 < n_end; nc0 += BWW_NC) {
        __syncthreads();                    // previous chunk fully consumed
        {   // dOut rows of this h-group, 32 columns: thread -> (row = k*8 + tid/32, col = tid%32)
            const int cc = tid & 31;
            const long n = nc0 + cc;
            const float cm = (n < n_end) ? 1.f : 0.f;
            const long ncl = n < N ? n : N - 1;
            float tmp[4 * MT];
#pragma unroll
            for (int k = 0; k < 4 * MT; ++k) {
                const int rr = k * 8 + (tid >> 5);
                const int h = hg * 32 * MT + rr;
                tmp[k] = dOut[(long)(h < H ? h : H - 1) * N + ncl] * ((h < H) ? cm : 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4 * MT; ++k) dOutS[(k * 8 + (tid >> 5)) * BWW_PITCH + cc] = tmp[k];
        }
        {   // this wave's 32 x_prev rows and JT x0 rows
            const long n = nc0 + c;
            const float cm = (n < n_end) ? 1.f : 0.f;
            const long ncl = n < N ? n : N - 1;
            float tmp[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = iblk * 32 + k * 2 + s;
                tmp[k] = xp[(long)(i < Hp ? i : Hp - 1) * N + ncl] * ((i < Hp) ? cm : 0.f);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) xpS[(k * 2 + s) * BWW_PITCH + c] = tmp[k];
#pragma unroll
            for (int k = 0; k < (JT + 1) / 2; ++k) {
                const int jt = k * 2 + s;
                if (jt < JT) {
                    const int j = jp * JT + jt;
                    x0S[jt * BWW_PITCH + c] = x0[(long)(j < m ? j : m - 1) * N + ncl] * ((j < m) ? cm : 0.f);
                }
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < BWW_NC / 2; ++t) {
            const int col = 2 * t + s;
            float a[MT], b[JT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = dOutS[(mt * 32 + c) * BWW_PITCH + col];
            const float xv = xpS[c * BWW_PITCH + col];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) b[jt] = xv * x0S[jt * BWW_PITCH + col];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    acc[mt][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[jt], acc[mt][jt], 0, 0, 0);
        }
    }

    if (!active) return;
    const int i = iblk * 32 + c;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int j = jp * JT + jt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int h = hg * 32 * MT + mt * 32 + frag_row(r, s);
                if (h < H && i < Hp && j < m)
                    atomicAdd(&dWt[((long)j * Hpad + h) * IPAD + i], acc[mt][jt][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// v2 of the dW kernel: operands reach LDS by LDS-DMA (global_load_lds_dword: no staging registers,
// asynchronous), two LDS buffers, so the loads of chunk c+1 run under the MFMAs of chunk c.
// LDS image of every region: element (row, col) at row*32 + (col ^ (row & 31)) -- the DMA writes
// lane-linear, so the swizzle is applied to each lane's SOURCE column; the fragment reads (32 lanes =
// 32 rows, same logical column) then touch 32 different banks.
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void dma_dword(const float* g, float* lds_wave_base) {
    x3_lds_dma4(g, x3_lds_addr(lds_wave_base));
}

template <int MT, int JT>
__global__ __launch_bounds__(256, 2) void cin_bwd_w_dma_kernel(
    const float* __restrict__ dOut, const float* __restrict__ xp, const float* __restrict__ x0,
    int H, int Hp, int m, long N, int IB, int JP, int TPH, long n_per_split, int Hpad, int IPAD,
    float* __restrict__ dWt, long slab_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(JT == 2, "one DMA instruction stages exactly two x0 rows");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, s = lane >> 5;
    const int wt0 = blockIdx.x * 4;
    const int hg = wt0 / TPH;
    const int tin = wt0 + wave - hg * TPH;
    const bool active = tin < JP * IB;
    const int jp = active ? tin / IB : 0;
    const int iblk = active ? tin - jp * IB : 0;
    const long n_begin = (long)blockIdx.y * n_per_split;
    const long n_end = (n_begin + n_per_split < N) ? n_begin + n_per_split : N;
    const int nfull = (int)((n_end - n_begin) / BWW_NC);          // chunks with all 32 columns < n_end

    constexpr int DROWS = 32 * MT;                      // dOut rows per h-group
    constexpr int WROWS = 32 + JT;                      // per-wave rows: x_prev block + x0 rows
    constexpr int BUF = (DROWS + 4 * WROWS) * 32;       // floats per LDS buffer
    constexpr int NDMA = DROWS / 8 + 16 + 1;            // DMA instructions per wave per chunk

    // DMA instruction k of a region covers rows (2k, 2k+1): lane (c, s) fetches row 2k+s, logical
    // column c ^ ((2k+s) & 31), and the hardware drops it at word 64k + lane of the region.
    // Addresses are recomputed per instruction from a few scalars to keep the VGPR budget for acc.
    const int j_own = jp * JT + s;
    const float* src0 = x0 + (long)(j_own < m ? j_own : m - 1) * N;
    const int drow0 = wave * (DROWS / 4);               // first dOut row (within the h-group) of this wave

    // DMA instruction `k` (0 .. NDMA-1) of the chunk starting at column nc0 into buffer `buf`
    auto dma_k = [&](int k, long nc0, int buf) {
        float* base = smem + buf * BUF;
        if (k < DROWS / 8) {
            const int row = drow0 + 2 * k + s;
            int h = hg * DROWS + row;
            h = h < H ? h : H - 1;
            dma_dword(dOut + (long)h * N + nc0 + (c ^ (row & 31)), base + (drow0 + 2 * k) * 32);
        } else if (k < DROWS / 8 + 16) {
            const int kk = k - DROWS / 8;
            float* wb = base + DROWS * 32 + wave * (WROWS * 32);
            int i = iblk * 32 + 2 * kk + s;
            i = i < Hp ? i : Hp - 1;
            dma_dword(xp + (long)i * N + nc0 + (c ^ (2 * kk + s)), wb + (2 * kk) * 32);
        } else {
            float* wb = base + DROWS * 32 + wave * (WROWS * 32);
            dma_dword(src0 + nc0 + (c ^ s), wb + 32 * 32);
        }
    };

    f32x16 acc[MT][JT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][jt][r] = 0.f;

    // k-steps of one chunk; when `next` >= 0 the DMA instructions of the following chunk are issued
    // in the shadow of the MFMAs (a few per k-step) instead of in a separate phase
    constexpr int KSTEPS = BWW_NC / 2;
    constexpr int DPS = (NDMA + KSTEPS - 1) / KSTEPS;   // DMA instructions per k-step
    auto compute = [&](int buf, long next_nc0, int next_buf) {
        const float* dS = smem + buf * BUF;
        const float* xS = dS + DROWS * 32 + wave * (WROWS * 32);
        const float* zS = xS + 32 * 32;
        const bool has_next = next_nc0 >= 0;
#pragma unroll
        for (int t = 0; t < KSTEPS; ++t) {
            const int col = 2 * t + s;
            float a[MT], b[JT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = dS[(mt * 32 + c) * 32 + (col ^ c)];
            const float xv = xS[c * 32 + (col ^ c)];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) b[jt] = xv * zS[jt * 32 + (col ^ jt)];
            if (has_next) {
#pragma unroll
                for (int q = 0; q < DPS; ++q)
                    if (t * DPS + q < NDMA) dma_k(t * DPS + q, next_nc0, next_buf);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    acc[mt][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[jt], acc[mt][jt], 0, 0, 0);
        }
    };

    if (nfull > 0) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) dma_k(k, n_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int ch = 0; ch < nfull; ++ch) {
        const long nxt = (ch + 1 < nfull) ? n_begin + (long)(ch + 1) * BWW_NC : -1;
        compute(ch & 1, nxt, (ch + 1) & 1);
        // one barrier per chunk: chunk ch+1 has landed everywhere AND everyone has finished reading
        // buf[ch&1], which the DMA issued during compute(ch+1) will overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // tail chunk with fewer than 32 valid columns: masked register staging into buffer 0
    const long nt0 = n_begin + (long)nfull * BWW_NC;
    if (nt0 < n_end) {
        float* base = smem;
        const long n = nt0 + c;
        const float cm = (n < n_end) ? 1.f : 0.f;
        const long ncl = n < N ? n : N - 1;
#pragma unroll
        for (int k = 0; k < DROWS / 8; ++k) {
            const int row = drow0 + 2 * k + s;
            const int h = hg * DROWS + row;
            base[row * 32 + (c ^ (row & 31))] = dOut[(long)(h < H ? h : H - 1) * N + ncl] * ((h < H) ? cm : 0.f);
        }
        float* wb = base + DROWS * 32 + wave * (WROWS * 32);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = 2 * k + s;
            const int i = iblk * 32 + row;
            wb[row * 32 + (c ^ row)] = xp[(long)(i < Hp ? i : Hp - 1) * N + ncl] * ((i < Hp) ? cm : 0.f);
        }
        wb[(32 + s) * 32 + (c ^ s)] = src0[ncl] * ((j_own < m) ? cm : 0.f);
        __syncthreads();
        compute(0, -1, 0);
    }

    if (!active) return;
    const int i = iblk * 32 + c;
    // slab mode: this n-split owns a private copy of dWt (plain 128-B row-segment stores, summed in a
    // fixed order by the unpack kernel: deterministic); otherwise fp32 atomics into one copy
    float* __restrict__ dst = dWt + (long)blockIdx.y * slab_stride;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int j = jp * JT + jt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int h = hg * 32 * MT + mt * 32 + frag_row(r, s);
                if (h < H && i < Hp && j < m) {
                    if (slab_stride) dst[((long)j * Hpad + h) * IPAD + i] = acc[mt][jt][r];
                    else atomicAdd(&dst[((long)j * Hpad + h) * IPAD + i], acc[mt][jt][r]);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// v3 of the dW kernel (N % 4 == 0): same structure as the dword-DMA kernel, fewer instructions per
// MFMA (with two waves per SIMD the shared vector-issue port, not the matrix pipe, was the limiter):
//   * operands arrive by 16-byte LDS-DMA (global_load_lds_dwordx4: 8 rows x 32 columns per
//     instruction, 9 instructions per wave and chunk instead of 33);
//   * LDS image: row r, 16-byte chunk q (4 columns) at r*32 + (q ^ (r & 7))*4 floats;
//   * the contraction index is re-ordered so that one ds_read_b64 feeds two k-steps: k-step t,
//     lane half s  <->  column 4*(t>>1) + 2*s + (t&1)  (A and B operands use the same map).
__device__ __forceinline__ void dma_x4(const float* g, float* lds_wave_base) {
    x3_lds_dma16(g, x3_lds_addr(lds_wave_base));
}

template <int MT, int JT>
__global__ __launch_bounds__(256, 2) void cin_bwd_w_dma4_kernel(
    const float* __restrict__ dOut, const float* __restrict__ xp, const float* __restrict__ x0,
    int H, int Hp, int m, long N, int IB, int JP, int TPH, long n_per_split, int Hpad, int IPAD,
    float* __restrict__ dWt, long slab_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(JT == 2, "x0 rows are staged as one (half-masked) DMA instruction");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, s = lane >> 5;
    const int wt0 = blockIdx.x * 4;
    const int hg = wt0 / TPH;
    const int tin = wt0 + wave - hg * TPH;
    const bool active = tin < JP * IB;
    const int jp = active ? tin / IB : 0;
    const int iblk = active ? tin - jp * IB : 0;
    const long n_begin = (long)blockIdx.y * n_per_split;
    const long n_end = (n_begin + n_per_split < N) ? n_begin + n_per_split : N;
    const int nfull = (int)((n_end - n_begin) / BWW_NC);

    constexpr int DROWS = 32 * MT;
    constexpr int WROWS = 32 + 8;                       // x_prev block + one 8-row group holding the x0 rows
    constexpr int BUF = (DROWS + 4 * WROWS) * 32;
    constexpr int ND_D = DROWS / 32;                    // dOut DMA instructions per wave (8 rows each)
    constexpr int NDMA = ND_D + 4 + 1;

    // lane -> (row within the 8-row group, physical chunk); logical chunk = pc ^ (row & 7)
    const int lrow = lane >> 3, pc = lane & 7;
    const int lcol = (pc ^ lrow) * 4;                   // (row & 7) == lrow because groups start at multiples of 8
    const int drow0 = wave * (DROWS / 4);

    auto dma_k = [&](int k, long nc0, int buf) {
        float* base = smem + buf * BUF;
        if (k < ND_D) {
            const int row = drow0 + 8 * k + lrow;
            int h = hg * DROWS + row;
            h = h < H ? h : H - 1;
            dma_x4(dOut + (long)h * N + nc0 + lcol, base + (drow0 + 8 * k) * 32);
        } else if (k < ND_D + 4) {
            const int kk = k - ND_D;
            float* wb = base + DROWS * 32 + wave * (WROWS * 32);
            int i = iblk * 32 + 8 * kk + lrow;
            i = i < Hp ? i : Hp - 1;
            dma_x4(xp + (long)i * N + nc0 + lcol, wb + (8 * kk) * 32);
        } else {
            float* wb = base + DROWS * 32 + wave * (WROWS * 32) + 32 * 32;
            int j = jp * JT + (lrow < JT ? lrow : JT - 1);     // rows >= JT of the group are never read
            j = j < m ? j : m - 1;
            dma_x4(x0 + (long)j * N + nc0 + lcol, wb);
        }
    };

    f32x16 acc[MT][JT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][jt][r] = 0.f;

    constexpr int USTEPS = BWW_NC / 4;                  // pairs of k-steps per chunk
    auto compute = [&](int buf, long next_nc0, int next_buf) {
        const float* dS = smem + buf * BUF;
        const float* xS = dS + DROWS * 32 + wave * (WROWS * 32);
        const float* zS = xS + 32 * 32;
        const bool has_next = next_nc0 >= 0;
#pragma unroll
        for (int u = 0; u < USTEPS; ++u) {
            // 8-byte pair s of logical chunk u of row r sits at r*32 + (u ^ (r&7))*4 + 2*s
            float2 a[MT], b2[JT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *reinterpret_cast<const float2*>(dS + (mt * 32 + c) * 32 + ((u ^ (c & 7)) * 4 + 2 * s));
            const float2 xv = *reinterpret_cast<const float2*>(xS + c * 32 + ((u ^ (c & 7)) * 4 + 2 * s));
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const float2 z = *reinterpret_cast<const float2*>(zS + jt * 32 + ((u ^ jt) * 4 + 2 * s));
                b2[jt] = make_float2(xv.x * z.x, xv.y * z.y);
            }
            if (has_next) {
                if (u < NDMA) dma_k(u, next_nc0, next_buf);
                if (u == USTEPS - 1) {
#pragma unroll
                    for (int k = USTEPS; k < NDMA; ++k) dma_k(k, next_nc0, next_buf);
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    acc[mt][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, b2[jt].x, acc[mt][jt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    acc[mt][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, b2[jt].y, acc[mt][jt], 0, 0, 0);
        }
    };

    if (nfull > 0) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) dma_k(k, n_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int ch = 0; ch < nfull; ++ch) {
        const long nxt = (ch + 1 < nfull) ? n_begin + (long)(ch + 1) * BWW_NC : -1;
        compute(ch & 1, nxt, (ch + 1) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // tail chunk (fewer than 32 valid columns): masked register staging into buffer 0, same image
    const long nt0 = n_begin + (long)nfull * BWW_NC;
    if (nt0 < n_end) {
        float* base = smem;
        const long n = nt0 + c;
        const float cm = (n < n_end) ? 1.f : 0.f;
        const long ncl = n < N ? n : N - 1;
        auto img = [&](int row, int col) { return row * 32 + (((col >> 2) ^ (row & 7)) * 4 + (col & 3)); };
#pragma unroll
        for (int k = 0; k < DROWS / 8; ++k) {
            const int row = drow0 + 2 * k + s;
            const int h = hg * DROWS + row;
            base[img(row, c)] = dOut[(long)(h < H ? h : H - 1) * N + ncl] * ((h < H) ? cm : 0.f);
        }
        float* wb = base + DROWS * 32 + wave * (WROWS * 32);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = 2 * k + s;
            const int i = iblk * 32 + row;
            wb[img(row, c)] = xp[(long)(i < Hp ? i : Hp - 1) * N + ncl] * ((i < Hp) ? cm : 0.f);
        }
        {
            const int j = jp * JT + s;
            wb[32 * 32 + img(s, c)] = x0[(long)(j < m ? j : m - 1) * N + ncl] * ((j < m) ? cm : 0.f);
        }
        __syncthreads();
        compute(0, -1, 0);
    }

    if (!active) return;
    const int i = iblk * 32 + c;
    float* __restrict__ dst = dWt + (long)blockIdx.y * slab_stride;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int j = jp * JT + jt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int h = hg * 32 * MT + mt * 32 + frag_row(r, s);
                if (h < H && i < Hp && j < m) {
                    if (slab_stride) dst[((long)j * Hpad + h) * IPAD + i] = acc[mt][jt][r];
                    else atomicAdd(&dst[((long)j * Hpad + h) * IPAD + i], acc[mt][jt][r]);
                }
            }
        }
}

// dW[h][i*m+j] = sum over n-splits of dWt[split][j][h][i]; threads walk the dWt layout (i fastest) so
// the slab reads are coalesced.
__global__ void cin_bwd_w_unpack_kernel(const float* __restrict__ dWt, int H, int Hp, int m, int Hpad, int IPAD,
                                        int nslab, long slab_stride, float* __restrict__ dW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)m * Hpad * IPAD;
    if (idx >= total) return;
    const int i = (int)(idx % IPAD);
    const long jh = idx / IPAD;
    const int h = (int)(jh % Hpad), j = (int)(jh / Hpad);
    if (i >= Hp || h >= H) return;
    float acc = 0.f;
    for (int k = 0; k < nslab; ++k) acc += dWt[(long)k * slab_stride + idx];
    dW[(long)h * ((long)Hp * m) + (long)i * m + j] = acc;
}

// =============================================================================================
// host side
// =============================================================================================
template <int HS4>
static int launch_bwd_x(const float* dOut, const float* xp, const float* x0, const float* Wz, int H, int Hp,
                        int m, long N, float* dxp, float* dx0, int flags, hipStream_t st) {
    const int IB = ceil_div(Hp, 32);
    size_t lds = (size_t)4 * m * 32 * sizeof(float);
    if (lds > 160 * 1024) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_x: m=%d needs %zu B of LDS", m, lds);
    if (xdfm_opt(OPT_DBG) & 4) lds = 80 * 1024;
    // optional: two interleaved chains per wave (even m, H <= 128).  Measured 110 vs 112 TFLOP/s for the
    // single dependent chain at config 2, so it stays off by default (A/B knob: dbg bit 5).
    if (HS4 <= 16 && HS4 >= 4 && m % 2 == 0 && (xdfm_opt(OPT_DBG) & 32))
        hipLaunchKernelGGL((cin_bwd_x_kernel<HS4, 2>), dim3(ceil_div(N, 128)), dim3(256), lds, st, dOut, xp, x0, Wz, H,
                           Hp, m, N, IB, dxp, dx0, flags);
    else
        hipLaunchKernelGGL((cin_bwd_x_kernel<HS4, 1>), dim3(ceil_div(N, 128)), dim3(256), lds, st, dOut, xp, x0, Wz, H,
                           Hp, m, N, IB, dxp, dx0, flags);
    return xdfm_check_launch("cin_level_bwd_x");
}

template <int MT>
static int launch_bwd_w(const float* dOut, const float* xp, const float* x0, int H, int Hp, int m, long N,
                        float* ws, float* dW, hipStream_t st) {
    constexpr int JT = BWW_JT;
    const BwwGeom g = bww_geometry(H, Hp, m, N);
    const bool slab = xdfm_opt(OPT_BWW_SLAB) != 0;
    if (!slab) {
        hipError_t e = hipMemsetAsync(ws, 0, (size_t)g.slab * sizeof(float), st);
        if (e != hipSuccess) return xdfm_fail(XDFM_ERR_LAUNCH, "cin_level_bwd_w memset: %s", hipGetErrorString(e));
    }
    const long stride = slab ? g.slab : 0;
    if (xdfm_opt(OPT_DBG) & 8) {      // v1: register staging, single LDS buffer (atomics only)
        if (slab) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_bwd_w: dbg 8 needs bww_slab=0");
        const size_t lds = (size_t)(32 * MT + 4 * (32 + JT)) * BWW_PITCH * sizeof(float);
        hipLaunchKernelGGL((cin_bwd_w_kernel<MT, JT>), dim3(g.gx, g.nsplit), dim3(256), lds, st, dOut, xp, x0, H, Hp, m,
                           N, g.IB, g.JP, g.TPH, g.n_per_split, g.Hpad, g.IPAD, ws);
    } else if (MT == 4 && N % 4 == 0 && !(xdfm_opt(OPT_DBG) & 16) &&
               ((((size_t)dOut) | ((size_t)xp) | ((size_t)x0)) & 15) == 0) {
        const size_t lds = (size_t)2 * (32 * MT + 4 * (32 + 8)) * 32 * sizeof(float);
        hipLaunchKernelGGL((cin_bwd_w_dma4_kernel<MT, JT>), dim3(g.gx, g.nsplit), dim3(256), lds, st, dOut, xp, x0, H, Hp,
                           m, N, g.IB, g.JP, g.TPH, g.n_per_split, g.Hpad, g.IPAD, ws, stride);
    } else {
        const size_t lds = (size_t)2 * (32 * MT + 4 * (32 + JT)) * 32 * sizeof(float);
        hipLaunchKernelGGL((cin_bwd_w_dma_kernel<MT, JT>), dim3(g.gx, g.nsplit), dim3(256), lds, st, dOut, xp, x0, H, Hp,
                           m, N, g.IB, g.JP, g.TPH, g.n_per_split, g.Hpad, g.IPAD, ws, stride);
    }
    int rc = xdfm_check_launch("cin_level_bwd_w");
    if (rc) return rc;
    hipLaunchKernelGGL(cin_bwd_w_unpack_kernel, dim3(ceil_div(g.slab, 256)), dim3(256), 0, st, ws, H, Hp, m, g.Hpad,
                       g.IPAD, slab ? g.nsplit : 1, stride, dW);
    return xdfm_check_launch("cin_level_bwd_w unpack");
}

extern "C" {

static bool cin_dout_vec(const float* A, long N, int D, const float* dOut, const float* dh, const float* dd) {
    return (N % 4 == 0) && (D % 4 == 0) && ((((size_t)A) | ((size_t)dOut) | ((size_t)dh) | ((size_t)dd)) & 15) == 0;
}

size_t xdfm_cin_dout_ws_elems(int H, int B, int D) {
    if (H <= 0 || B <= 0 || D <= 0) return 0;
    return (size_t)H * ceil_div((long)B * D, 1024);      // enough for either vector width
}

int xdfm_cin_dout(const float* A, int H, int B, int D, int act, const float* dHid, int hid0, int hid_rows,
                  const float* dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
                  float* dOut, float* dbias, void* stream) {
    return xdfm_cin_dout_det(A, H, B, D, act, dHid, hid0, hid_rows, dDir, dir_mode, lddir, dir_off, dir0, dir_rows, dOut, dbias,
                             nullptr, stream);
}

static int cin_dout_impl(const float* A, const unsigned* mask, long mask_ld, int H, int B, int D, int act, const float* dHid,
                         int hid0, int hid_rows, const float* dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
                         float* dOut, float* dbias, float* ws, void* stream);

int xdfm_cin_dout_det(const float* A, int H, int B, int D, int act, const float* dHid, int hid0, int hid_rows,
                      const float* dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
                      float* dOut, float* dbias, float* ws, void* stream) {
    return cin_dout_impl(A, nullptr, 0, H, B, D, act, dHid, hid0, hid_rows, dDir, dir_mode, lddir, dir_off, dir0, dir_rows, dOut,
                         dbias, ws, stream);
}

static int cin_dout_impl(const float* A, const unsigned* mask, long mask_ld, int H, int B, int D, int act, const float* dHid,
                         int hid0, int hid_rows, const float* dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
                         float* dOut, float* dbias, float* ws, void* stream) {
    XDFM_REQUIRE((A || mask) && dOut && dbias, "cin_dout: null pointer");
    XDFM_REQUIRE(H > 0 && B > 0 && D > 0, "cin_dout: bad shape H=%d B=%d D=%d", H, B, D);
    XDFM_REQUIRE(act == XDFM_ACT_LINEAR || act == XDFM_ACT_RELU, "cin_dout: unsupported activation %d", act);
    XDFM_REQUIRE(hid_rows >= 0 && dir_rows >= 0 && hid0 >= 0 && dir0 >= 0 && hid0 + hid_rows <= H &&
                     dir0 + dir_rows <= H, "cin_dout: row ranges outside [0,%d)", H);
    XDFM_REQUIRE(dir_mode == 0 || dir_mode == 1, "cin_dout: dir_mode %d", dir_mode);
    const long N = (long)B * D;
    const float* dh = hid_rows > 0 ? dHid : nullptr;
    const float* dd = dir_rows > 0 ? dDir : nullptr;
    const bool vec = cin_dout_vec(mask ? dOut : A, N, D, dOut, dh, dd);
    const int gx = ceil_div(N, vec ? 4096 : 1024);
    unsigned* ticket = (ws && H <= TK_ROWS) ? xdfm_ticket(TK_ROW0) : nullptr;
    if (vec)
        hipLaunchKernelGGL(cin_dout_kernel<4>, dim3(gx, H), dim3(256), 0, (hipStream_t)stream, A, H, N, D,
                           act, dh, hid0, hid_rows, dd, dir_mode, lddir, dir_off, dir0, dir_rows, dOut, dbias, ws, ticket, mask, mask_ld);
    else
        hipLaunchKernelGGL(cin_dout_kernel<1>, dim3(gx, H), dim3(256), 0, (hipStream_t)stream, A, H, N, D,
                           act, dh, hid0, hid_rows, dd, dir_mode, lddir, dir_off, dir0, dir_rows, dOut, dbias, ws, ticket, mask, mask_ld);
    if (ws && !ticket) hipLaunchKernelGGL(cin_dbias_finish_kernel, dim3(ceil_div(H, 64)), dim3(64), 0, (hipStream_t)stream, ws, H, gx, dbias);
    return xdfm_check_launch("cin_dout");
}

size_t xdfm_cin_bwd_pack_elems(int H, int Hp, int m) {
    if (H <= 0 || Hp <= 0 || m <= 0 || H > 256) return 0;
    if (x3_bwx_usable(H, Hp, m)) return x3_bwx_pack_elems(H, Hp, m);
    return ((size_t)ceil_div(Hp, 32) * m * bwx_hs4(H) + BWX_TAIL(bwx_hs4(H))) * 256;
}

int xdfm_cin_bwd_pack(const float* W, int H, int Hp, int m, float* Wz, void* stream) {
    XDFM_REQUIRE(W && Wz, "cin_bwd_pack: null pointer");
    XDFM_REQUIRE(H > 0 && H <= 256 && Hp > 0 && m > 0, "cin_bwd_pack: bad shape H=%d (<=256) Hp=%d m=%d", H, Hp, m);
    if (x3_bwx_usable(H, Hp, m)) return x3_bwx_pack(W, H, Hp, m, Wz, (hipStream_t)stream);
    const int HS4 = bwx_hs4(H);
    const long real_groups = (long)ceil_div(Hp, 32) * m * HS4;
    const long total = (real_groups + BWX_TAIL(HS4)) * 256;
    hipLaunchKernelGGL(cin_bwd_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, H, Hp,
                       m, HS4, total, real_groups, Wz);
    return xdfm_check_launch("cin_bwd_pack");
}

int xdfm_cin_level_bwd_x(const float* dOut, const float* xp, const float* x0, const float* Wz, int H, int Hp,
                         int m, long N, float* dxp, float* dx0, void* stream) {
    return xdfm_cin_level_bwd_x_ex(dOut, xp, x0, Wz, H, Hp, m, N, dxp, dx0, 0, stream);
}

int xdfm_cin_level_bwd_x_ex(const float* dOut, const float* xp, const float* x0, const float* Wz, int H, int Hp,
                            int m, long N, float* dxp, float* dx0, int flags, void* stream) {
    XDFM_REQUIRE(dOut && xp && x0 && Wz && dxp && dx0, "cin_level_bwd_x: null pointer");
    XDFM_REQUIRE((flags & ~(XDFM_BWX_SET_DXP | XDFM_BWX_SET_DX0)) == 0, "cin_level_bwd_x: unknown flags 0x%x", flags);
    XDFM_REQUIRE(H > 0 && H <= 256 && Hp > 0 && m > 0 && N > 0, "cin_level_bwd_x: bad shape H=%d (<=256) Hp=%d m=%d",
                 H, Hp, m);
    hipStream_t st = (hipStream_t)stream;
    if (x3_bwx_usable(H, Hp, m)) { xdfm_opt_note(OPT_LAST_BWX, xdfm_opt(OPT_CIN_MATH)); return x3_level_bwd_x(x3_dout_plain(dOut), xp, x0, Wz, H, Hp, m, N, dxp, dx0, flags, st); }
    xdfm_opt_note(OPT_LAST_BWX, 0);
    xdfm_opt_note(OPT_LAST_SYM, xdfm_opt(OPT_LAST_SYM) & ~2);
    switch (bwx_hs4(H)) {
        case 1: return launch_bwd_x<1>(dOut, xp, x0, Wz, H, Hp, m, N, dxp, dx0, flags, st);
        case 2: return launch_bwd_x<2>(dOut, xp, x0, Wz, H, Hp, m, N, dxp, dx0, flags, st);
        case 4: return launch_bwd_x<4>(dOut, xp, x0, Wz, H, Hp, m, N, dxp, dx0, flags, st);
        case 8: return launch_bwd_x<8>(dOut, xp, x0, Wz, H, Hp, m, N, dxp, dx0, flags, st);
        case 16: return launch_bwd_x<16>(dOut, xp, x0, Wz, H, Hp, m, N, dxp, dx0, flags, st);
        default: return launch_bwd_x<32>(dOut, xp, x0, Wz, H, Hp, m, N, dxp, dx0, flags, st);
    }
}

int xdfm_cin_bwd_x_is_folded(int H, int Hp, int m, int xp_is_x0) {
    // the condition of x3_level_bwd_x (cin_x3.hip) for the folded level-0 kernel, as a function of the arguments
    return (H > 0 && H <= 256 && Hp > 0 && m > 0 && xp_is_x0 && x3_bwx_usable(H, Hp, m) && Hp == m && x3_sym_m(m) &&
            xdfm_opt(OPT_X3_SYM) != 0) ? 1 : 0;
}

size_t xdfm_cin_bwd_w_ws_elems(int H, int Hp, int m, long N) {
    if (H <= 0 || Hp <= 0 || m <= 0 || N <= 0) return 0;
    const BwwGeom g = bww_geometry(H, Hp, m, N);
    const size_t f32 = (size_t)g.slab * (size_t)(xdfm_opt(OPT_BWW_SLAB) ? g.nsplit : 1);
    if (x3_terms() != 0) {      // whether the f16x3 / bf16 kernel runs also depends on pointer alignment
        const size_t x3 = x3_bww_ws_elems(H, Hp, m, N);
        return x3 > f32 ? x3 : f32;
    }
    return f32;
}

int xdfm_cin_level_bwd_w(const float* dOut, const float* xp, const float* x0, int H, int Hp, int m, long N,
                         float* ws, float* dW, void* stream) {
    XDFM_REQUIRE(dOut && xp && x0 && ws && dW, "cin_level_bwd_w: null pointer");
    XDFM_REQUIRE(H > 0 && Hp > 0 && m > 0 && N > 0, "cin_level_bwd_w: bad shape H=%d Hp=%d m=%d", H, Hp, m);
    hipStream_t st = (hipStream_t)stream;
    if (x3_bww_usable(dOut, xp, x0, H, N)) { xdfm_opt_note(OPT_LAST_BWW, xdfm_opt(OPT_CIN_MATH)); return x3_level_bwd_w(dOut, xp, x0, H, Hp, m, N, ws, dW, false, st); }
    xdfm_opt_note(OPT_LAST_BWW, 0);
    xdfm_opt_note(OPT_LAST_SYM, xdfm_opt(OPT_LAST_SYM) & ~4);
    const int phase = xdfm_opt(OPT_BWW_PHASE);            // fp32 kernels: the whole call counts as phase 2
    if (phase == 1 || phase == 3) return XDFM_OK;
    switch (bww_mt(H)) {
        case 1: return launch_bwd_w<1>(dOut, xp, x0, H, Hp, m, N, ws, dW, st);
        case 2: return launch_bwd_w<2>(dOut, xp, x0, H, Hp, m, N, ws, dW, st);
        default: return launch_bwd_w<4>(dOut, xp, x0, H, Hp, m, N, ws, dW, st);
    }
}

// ---- cin_dout + the dW kernel's operands in one pass (f16x3 / bf16 arithmetic) ----------------------------------
size_t xdfm_cin_bwd_prep_ws_elems(int H, int Hp, int m, int B, int D) {
    if (H <= 0 || Hp <= 0 || m <= 0 || B <= 0 || D <= 0) return 0;
    const long N = (long)B * D;
    size_t n = xdfm_cin_dout_ws_elems(H, B, D);
    if (x3_terms() != 0 && N >= 32) {          // blocks per row of the fused pass: the larger of the two tilings a level may take
        int blocks = x3_bwd_prep_blocks(false, H, Hp, m, N);
        const int bs = x3_bwd_prep_blocks(true, H, Hp, m, N);
        if (bs > blocks) blocks = bs;
        if ((size_t)H * blocks > n) n = (size_t)H * blocks;
    }
    return n;
}

int xdfm_cin_bwd_prep(const float* A, const unsigned* mask, long mask_ld, int H, int B, int D, int act, const float* dHid,
                      int hid0, int hid_rows, const float* dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
                      float* dOut, float* dbias, float* dout_ws, const float* xp, const float* x0, int Hp, int m, float* bww_ws,
                      int* prepared, void* stream) {
    XDFM_REQUIRE(prepared, "cin_bwd_prep: null pointer");
    XDFM_REQUIRE(!mask || (mask_ld >= H && mask_ld % 4 == 0), "cin_bwd_prep: mask pitch %ld", mask_ld);
    *prepared = 0;
    const long N = (long)B * D;
    const float* dh = hid_rows > 0 ? dHid : nullptr;
    const float* dd = dir_rows > 0 ? dDir : nullptr;
    const bool fused = dout_ws && bww_ws && xp && x0 && Hp > 0 && m > 0 && H > 0 && B > 0 && D > 0 && (A || mask) && (dOut || mask) &&
                       x3_bww_usable(dOut, xp, x0, H, N) && cin_dout_vec(mask ? dOut : A, N, D, dOut, dh, dir_mode == 1 ? dd : nullptr);
    XDFM_REQUIRE(fused || dOut, "cin_bwd_prep: dOut may only be omitted when the pass is fused (xdfm_cin_bwd_nodout_supported)");
    if (!fused)
        return cin_dout_impl(A, mask, mask_ld, H, B, D, act, dHid, hid0, hid_rows, dDir, dir_mode, lddir, dir_off, dir0, dir_rows, dOut,
                             dbias, dout_ws, stream);
    XDFM_REQUIRE(dbias, "cin_bwd_prep: null pointer");
    XDFM_REQUIRE(act == XDFM_ACT_LINEAR || act == XDFM_ACT_RELU, "cin_bwd_prep: unsupported activation %d", act);
    XDFM_REQUIRE(hid_rows >= 0 && dir_rows >= 0 && hid0 >= 0 && dir0 >= 0 && hid0 + hid_rows <= H && dir0 + dir_rows <= H,
                 "cin_bwd_prep: row ranges outside [0,%d)", H);
    XDFM_REQUIRE(dir_mode == 0 || dir_mode == 1, "cin_bwd_prep: dir_mode %d", dir_mode);
    hipStream_t st = (hipStream_t)stream;
    unsigned* ticket = H <= TK_ROWS ? xdfm_ticket(TK_ROW0) : nullptr;
    int rc = x3_bwd_prep(A, mask, mask_ld, H, N, D, act, dh, hid0, hid_rows, dd, dir_mode, lddir, dir_off, dir0, dir_rows, dOut, dout_ws, dbias,
                         ticket, xp, x0, Hp, m, bww_ws, st);
    if (rc) return rc;
    if (!ticket)
        hipLaunchKernelGGL(cin_dbias_finish_kernel, dim3(ceil_div(H, 64)), dim3(64), 0, st, dout_ws, H,
                           x3_bwd_prep_blocks(xp == x0, H, Hp, m, N), dbias);
    *prepared = 1;
    return xdfm_check_launch("cin_bwd_prep");
}

int xdfm_cin_bwd_nodout_supported(int H, int Hp, int m, int B, int D) {
    const long N = (long)B * D;
    return (H > 0 && Hp > 0 && m > 0 && B > 0 && (D == 4 || D == 8 || D == 16 || D == 32) && x3_terms() != 0 && bww_mt(H) == 4 && N % 4 == 0 &&
            N >= 32 && x3_bwx_usable(H < 256 ? H : 256, Hp, m) && (H <= 256 || H % 256 == 0 || x3_bwx_usable(H % 256, Hp, m)) &&
            H <= (1 << 16)) ? 1 : 0;
}

int xdfm_cin_level_bwd_x_src(const unsigned* mask, long mask_ld, const float* dHid, int hid_rows, const float* dDir, int dir_mode,
                             long lddir, int dir_off, int dir0, int dir_rows, int D, int h0, const float* xp, const float* x0,
                             const float* Wz, int H, int Hp, int m, long N, float* dxp, float* dx0, int flags, void* stream) {
    XDFM_REQUIRE(xp && x0 && Wz && dxp && dx0 && (dHid || dDir), "cin_level_bwd_x_src: null pointer");
    XDFM_REQUIRE((flags & ~(XDFM_BWX_SET_DXP | XDFM_BWX_SET_DX0)) == 0, "cin_level_bwd_x_src: unknown flags 0x%x", flags);
    XDFM_REQUIRE(H > 0 && H <= 256 && Hp > 0 && m > 0 && N > 0 && h0 >= 0 && h0 % 4 == 0, "cin_level_bwd_x_src: bad shape H=%d (<=256) Hp=%d m=%d h0=%d",
                 H, Hp, m, h0);
    XDFM_REQUIRE(x3_bwx_usable(H, Hp, m), "cin_level_bwd_x_src: no f16x3 / bf16 dX kernel for H=%d", H);
    XDFM_REQUIRE(D == 4 || D == 8 || D == 16 || D == 32, "cin_level_bwd_x_src: D=%d", D);
    XDFM_REQUIRE(!mask || (mask_ld >= h0 + H && mask_ld % 4 == 0 && (((size_t)mask) & 15) == 0), "cin_level_bwd_x_src: mask pitch %ld", mask_ld);
    XDFM_REQUIRE(dir_mode == 0 || dir_mode == 1, "cin_level_bwd_x_src: dir_mode %d", dir_mode);
    int logD = 0;
    while ((1 << logD) < D) ++logD;
    const X3DoutSrc S = {nullptr, mask, mask_ld, hid_rows > 0 ? dHid : nullptr, hid_rows, dir_rows > 0 ? dDir : nullptr, dir_mode, lddir,
                         dir_off, dir0, dir_rows, logD, h0};
    xdfm_opt_note(OPT_LAST_BWX, xdfm_opt(OPT_CIN_MATH));
    return x3_level_bwd_x(S, xp, x0, Wz, H, Hp, m, N, dxp, dx0, flags, (hipStream_t)stream);
}

int xdfm_cin_level_bwd_w_prepared(const float* dOut, const float* xp, const float* x0, int H, int Hp, int m, long N,
                                  float* ws, float* dW, void* stream) {
    XDFM_REQUIRE(xp && x0 && ws && dW, "cin_level_bwd_w: null pointer");       // dOut itself is not read (may be NULL): the planes are
    XDFM_REQUIRE(H > 0 && Hp > 0 && m > 0 && N > 0, "cin_level_bwd_w: bad shape H=%d Hp=%d m=%d", H, Hp, m);
    XDFM_REQUIRE(x3_bww_usable(dOut, xp, x0, H, N), "cin_level_bwd_w_prepared: no f16x3 / bf16 dW kernel for this call (H=%d)", H);
    xdfm_opt_note(OPT_LAST_BWW, xdfm_opt(OPT_CIN_MATH));
    return x3_level_bwd_w(dOut, xp, x0, H, Hp, m, N, ws, dW, true, (hipStream_t)stream);
}

}  // extern "C"
