// K1 / K2: sparse embedding gather (forward) and dense-gradient scatter (backward).
//
// replaces deepctr/models/basemodel.py:368-370 (26 x slice -> .long() -> nn.Embedding),
// deepctr/models/basemodel.py:63-92 (Linear), deepctr/models/xdeepfm.py:86 and
// deepctr/inputs.py:126-132 (the two concatenations) with ONE launch, and the 52
// aten::embedding_dense_backward calls of their autograd with one more.
//
// Forward data flow per workgroup (EB consecutive examples):
//   HBM rows --(16-B or 4-B loads, one row = D floats)--> LDS tile [EB][m][D]
//   LDS tile --> dnn_in[b][0 .. m*D)        contiguous m*D floats per example
//            --> emb_fm[j][b0*D .. (b0+EB)*D)  contiguous EB*D floats per field   (FM layout)
// so both outputs are written in long contiguous runs although the rows arrive in id order.
#include "xdfm_internal.h"

#define EMB_THREADS 256

// Index arithmetic: no division by a run-time value sits in a loop (a v_div sequence is ~40 instructions; with one
// per copied element the kernel was instruction-bound at 2.2 TB/s however large the batch).  Phase 0 also parks
// everything later phases would otherwise fetch through dependent global loads -- row addresses, the dense
// values of X, the dense weights -- in LDS, so that phases 1 and 2 issue independent loads / stores only.
template <int VEC>
__global__ __launch_bounds__(EMB_THREADS) void embed_gather_kernel(
    const float* __restrict__ X, long ldx, int B, const float* const* __restrict__ tables,
    const float* const* __restrict__ lin_tables, const int* __restrict__ cols, const int* __restrict__ vocab,
    int m, int D, const int* __restrict__ dense_cols, const float* __restrict__ dense_w, int nd, int EB,
    float* __restrict__ emb_fm, float* __restrict__ dnn_in, float* __restrict__ lin_out, int* __restrict__ err_flag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                                    // [EB][m][D]
    float* linv = smem + (size_t)EB * m * D;                               // [EB][m]
    const float** rowp = reinterpret_cast<const float**>(smem + ((((size_t)EB * m * (D + 1)) + 1) & ~(size_t)1));   // [EB][m]
    float* densev = reinterpret_cast<float*>(rowp + (size_t)EB * m);       // [EB][nd]
    float* wl = densev + (size_t)EB * nd;                                  // [nd]
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * EB;
    const int nb = (B - b0 < EB) ? B - b0 : EB;
    const int DV = D / VEC;
    const long N = (long)B * D;

    // phase 0a: dense values of the block's rows and the dense weights -> LDS (loads issued first, parked last)
    const int ndense = nb * nd;
    float dval[2] = {0.f, 0.f};
    float wval = 0.f;
    if (nd > 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int idx = tid + t * EMB_THREADS;
            const int c = idx < ndense ? idx : 0;
            const int bl = c / nd, k = c - bl * nd;
            dval[t] = X[(long)(b0 + bl) * ldx + dense_cols[k]];
        }
        if (dense_w) wval = dense_w[tid < nd ? tid : 0];
    }
    // phase 0b: ids (clamped; out-of-range ids raise the flag) -> row addresses and linear-table values.
    // m <= 32: lane & 31 is the field, 8 examples per pass; wider models take the division
    const int npass = m <= 32 ? (nb + 7) / 8 : (nb * m + EMB_THREADS - 1) / EMB_THREADS;
    for (int p = 0; p < npass; ++p) {
        int jf, bl;
        if (m <= 32) { jf = tid & 31; bl = (tid >> 5) + 8 * p; }
        else { const int rj = tid + p * EMB_THREADS; bl = rj / m; jf = rj - bl * m; }
        const bool live = jf < m && bl < nb;
        const int jc = live ? jf : 0, bc = live ? bl : 0;
        const float* tab = tables[jc];
        const float* ltab = lin_tables ? lin_tables[jc] : tab;
        const int V = vocab[jc];
        long id = (long)X[(long)(b0 + bc) * ldx + cols[jc]];        // truncation, as Tensor.long() (basemodel.py:369)
        if (id < 0 || id >= V) {
            if (err_flag && live) atomicOr(err_flag, 1);
            id = id < 0 ? 0 : V - 1;
        }
        const float lv = ltab[lin_tables ? id : 0];
        if (live) {
            rowp[bl * m + jf] = tab + id * D;
            if (lin_tables) linv[bl * m + jf] = lv;
        }
    }
    if (nd > 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (tid + t * EMB_THREADS < ndense) densev[tid + t * EMB_THREADS] = dval[t];
        for (int idx = tid + 2 * EMB_THREADS; idx < ndense; idx += EMB_THREADS) {       // > 512 dense values: rare
            const int bl = idx / nd, k = idx - bl * nd;
            densev[idx] = X[(long)(b0 + bl) * ldx + dense_cols[k]];
        }
        if (dense_w) {
            if (tid < nd) wl[tid] = wval;
            for (int k = tid + EMB_THREADS; k < nd; k += EMB_THREADS) wl[k] = dense_w[k];
        }
    }
    __syncthreads();
    // phase 1: gather rows into the LDS tile (chunk = VEC floats of one row).  GB chunks per thread are loaded
    // back to back before any of them is stored, so a block pays about one HBM round trip for all its rows
    // instead of one per loop iteration.
    const int nchunks = nb * m * DV;
    const bool dv_pow2 = (DV & (DV - 1)) == 0;
    const int dv_sh = 31 - __builtin_clz(DV);
    constexpr int GB = 8;
    for (int base = 0; base < nchunks; base += GB * EMB_THREADS) {
        float4 v[GB];
        int at[GB];
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            const int idx = base + k * EMB_THREADS + tid;
            const int cidx = idx < nchunks ? idx : nchunks - 1;
            const int rj = dv_pow2 ? (cidx >> dv_sh) : (cidx / DV);
            const int q = cidx - rj * DV;
            const float* src = rowp[rj] + q * VEC;
            at[k] = idx < nchunks ? rj * D + q * VEC : -1;
            if constexpr (VEC == 4) {
                v[k] = *reinterpret_cast<const float4*>(src);
            } else if constexpr (VEC == 2) {
                const float2 t = *reinterpret_cast<const float2*>(src);
                v[k] = make_float4(t.x, t.y, 0.f, 0.f);
            } else {
                v[k] = make_float4(*src, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            if (at[k] < 0) continue;
            float* dst = tile + at[k];
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst) = v[k];
            else if constexpr (VEC == 2) *reinterpret_cast<float2*>(dst) = make_float2(v[k].x, v[k].y);
            else *dst = v[k].x;
        }
    }
    __syncthreads();

    // phase 2a: dnn_in rows (sparse part is the tile verbatim, dense part from the staged values)
    if (dnn_in) {
        const int ldd = m * D + nd;
        const int per = m * D;
        for (int bl = 0; bl < nb; ++bl) {
            float* drow = dnn_in + (long)(b0 + bl) * ldd;
            const float* trow = tile + (size_t)bl * per;
            for (int k = tid; k < per; k += EMB_THREADS) drow[k] = trow[k];
            for (int k = tid; k < nd; k += EMB_THREADS) drow[per + k] = densev[bl * nd + k];
        }
    }
    // phase 2b: FM layout, field-major: for field j the nb*D floats of this block are contiguous
    {
        const int per = nb * D;
        const bool d_pow2 = (D & (D - 1)) == 0;
        const int d_sh = 31 - __builtin_clz(D);
        for (int k = tid; k < per; k += EMB_THREADS) {
            const int bl = d_pow2 ? (k >> d_sh) : (k / D);
            const float* src = tile + (size_t)bl * m * D + (k - bl * D);
            float* dst = emb_fm + (long)b0 * D + k;
            for (int j = 0; j < m; ++j) dst[(long)j * N] = src[j * D];
        }
    }
    // phase 2c: linear logit, summed in field order then dense columns in column order
    if (lin_out && tid < nb) {
        float acc = 0.f;
        if (lin_tables)
            for (int j = 0; j < m; ++j) acc += linv[tid * m + j];
        float dacc = 0.f;
        for (int k = 0; k < nd; ++k) dacc += densev[tid * nd + k] * wl[k];
        lin_out[b0 + tid] = acc + dacc;
    }
}

// ---------------------------------------------------------------------------------------------
// K2 backward: dense table gradients from the row gradients, WITHOUT atomics -- a sorted, segmented, exact reduce.
//
// The reference's embedding_dense_backward (deepctr/inputs.py:168, sparse=False) adds the rows of one id in example
// order on the CPU: deterministic.  Arrival-order fp32 atomics are not (replicated tables of a row-parallel run
// drift apart in the last bits), and a hot id serialises thousands of atomics on one row.  Here one workgroup owns
// one (field, 16-column slice) of a chunk of <= SC_ROWS examples:
//   S   sort the chunk's (id, example) keys of the field in LDS (bitonic network on 64-bit composite keys);
//   1   walk the sorted list in windows of 8 positions (4 lanes per window, one float4 of the row each, all 8 row
//       loads of a window in flight): runs of equal ids that lie inside a window are summed and written at once;
//   2-4 runs that cross windows: their owner (the window where the run starts) collects the pieces through LDS.
// Every sum is EXACT: the addends of a run are rounded once to a fixed-point grid 2^-s derived from the run's
// largest magnitude (s = 37 - exponent: the grid is 2^-13 of an fp32 ulp of that magnitude or finer) and added as
// integers held in doubles (< 2^52, so every addition is exact), then rounded once to fp32.  The result is a pure
// function of the MULTISET of rows of an id: independent of the order of the examples, of the window cuts and of
// the hardware -- bit-identical on every rank and under any permutation of the batch (tests/test_gpu_parity.py).
// One row is written by exactly one lane (read-modify-write of d_flat, which holds zeros or the L2 gradient).
#define SC_THREADS 512
#define SC_ROWS 4096          // examples per chunk (LDS: 8 B of key per example + 25.5 B of slots per example)
#define SC_W 8                // sorted positions per window
#define SC_SW 16              // embedding columns per workgroup slice (4 lanes x float4)
#define SC_NC 5               // components per lane: 4 columns + the linear-table gradient (lane 0 of slice 0)
#define SC_NOID 0xffffffffu
// keys live at padded positions (one spare slot after every 8): a thread's 8 consecutive keys and a wave's
// consecutive keys are both free of LDS bank conflicts
#define SC_K(i) ((i) + ((i) >> 3))

__device__ __forceinline__ int sc_scale_exp(float amax) {
    // s with amax * 2^s in [2^37, 2^38): the sum of <= 2^13 such addends stays below 2^51
    const int E = (int)((__float_as_uint(amax) >> 23) & 0xff);          // biased exponent (0: zero / denormal)
    return 37 - ((E == 0 ? 1 : E) - 127);
}
__device__ __forceinline__ double sc_pow2(int s) { return __longlong_as_double((long long)(1023 + s) << 52); }
__device__ __forceinline__ double sc_fix(float v, double sc) { return __builtin_rint((double)v * sc); }

__host__ __device__ inline size_t sc_keys_bytes(int npad) { return (size_t)(SC_K(npad + 8) + 1) * 8; }

struct ScWin {                 // one window of the sorted list, as every lane of its group sees it
    unsigned id[SC_W];
    int b[SC_W];
    int count;                 // live positions (the last window of a chunk may be short)
    bool contL, contR;         // first run continues from the previous window / last run continues into the next
};

__device__ __forceinline__ void sc_window(const unsigned long long* keys, int w, int nb, ScWin& W) {
    const int base = w * SC_W;
    W.count = nb - base < SC_W ? nb - base : SC_W;
#pragma unroll
    for (int q = 0; q < SC_W; ++q) {
        const unsigned long long k = keys[SC_K(base + q)];    // padding keys are all ones: id = SC_NOID
        W.id[q] = (unsigned)(k >> 32);
        W.b[q] = (int)(unsigned)k;
    }
    W.contL = base > 0 && (unsigned)(keys[SC_K(base - 1)] >> 32) == W.id[0];
    W.contR = W.count == SC_W && base + SC_W < nb && (unsigned)(keys[SC_K(base + SC_W)] >> 32) == W.id[SC_W - 1];
}
// window w is one single run that comes from the left and goes on to the right
__device__ __forceinline__ bool sc_is_middle(const unsigned long long* keys, int w, int nb) {
    const int base = w * SC_W;
    if (base + SC_W >= nb) return false;
    const unsigned a = (unsigned)(keys[SC_K(base)] >> 32);
    return a == (unsigned)(keys[SC_K(base + SC_W - 1)] >> 32) && a == (unsigned)(keys[SC_K(base + SC_W)] >> 32) &&
           a == (unsigned)(keys[SC_K(base - 1)] >> 32);
}

struct ScSrc {
    const float* de; const float* dd; const float* dl;
    long ld_dnn, ld_lin, Btot;
    int b0, j, D, c0;          // c0: first column of this lane; columns c0 .. c0+3 (those < D are live)
    bool lin;                  // this lane also carries the linear-table gradient
    bool dd_vec;               // rows of d_dnn_in are 16-byte aligned (ld_dnn % 4 == 0): float4 loads
};

// the lane's 5 components of the row gradient of local example bl: d_emb_fm + d_dnn_in (fp32 add, as autograd's
// accumulation of the two uses of the embedding), and d_lin
template <int VEC>
__device__ __forceinline__ void sc_load(const ScSrc& S, int bl, float (&v)[SC_NC]) {
    const long b = S.b0 + bl;
    float e[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
    if (S.c0 < S.D) {
        if constexpr (VEC == 4) {
            if (S.de) { const float4 t = *reinterpret_cast<const float4*>(S.de + ((long)S.j * S.Btot + b) * S.D + S.c0); e[0] = t.x; e[1] = t.y; e[2] = t.z; e[3] = t.w; }
            if (S.dd) {
                const float* p = S.dd + b * S.ld_dnn + (long)S.j * S.D + S.c0;
                if (S.dd_vec) { const float4 t = *reinterpret_cast<const float4*>(p); d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
                else { d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3]; }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (S.c0 + c < S.D) {
                    if (S.de) e[c] = S.de[((long)S.j * S.Btot + b) * S.D + S.c0 + c];
                    if (S.dd) d[c] = S.dd[b * S.ld_dnn + (long)S.j * S.D + S.c0 + c];
                }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (S.de ? e[c] : 0.f) + (S.dd ? d[c] : 0.f);
    v[4] = S.lin ? S.dl[b * S.ld_lin] : 0.f;
}

struct ScDst {
    float* d_flat; unsigned char* marks;
    long tab_base, lin_base;   // element offsets of the field's table / linear table in d_flat (-1: absent)
    int D, c0;
    bool lin;
};

// d_flat[row id] += the finished sums of a run (one writer per element: plain read-modify-write)
template <int VEC>
__device__ __forceinline__ void sc_store(const ScDst& T, unsigned id, const double (&tot)[SC_NC], const int (&s)[SC_NC]) {
    float r[SC_NC];
#pragma unroll
    for (int c = 0; c < SC_NC; ++c) r[c] = (float)(tot[c] * sc_pow2(-s[c]));
    if (T.tab_base >= 0 && T.c0 < T.D) {
        const long e = T.tab_base + (long)id * T.D + T.c0;
        if (VEC == 4 && (e & 3) == 0) {
            float4* p = reinterpret_cast<float4*>(T.d_flat + e);
            float4 o = *p;
            o.x += r[0]; o.y += r[1]; o.z += r[2]; o.w += r[3];
            *p = o;
            if (T.marks) T.marks[e >> 2] = 1;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (T.c0 + c < T.D) {
                    T.d_flat[e + c] += r[c];
                    if (T.marks) T.marks[(e + c) >> 2] = 1;
                }
        }
    }
    if (T.lin && T.lin_base >= 0) {
        const long e = T.lin_base + (long)id;
        T.d_flat[e] += r[4];
        if (T.marks) T.marks[e >> 2] = 1;
    }
}

template <int VEC>
__global__ __launch_bounds__(SC_THREADS) void embed_scatter_sorted_kernel(
    const float* __restrict__ X, long ldx, int b0, int nb, long Btot, const int* __restrict__ cols,
    const int* __restrict__ vocab, int m, int D, int nslice, const float* __restrict__ d_emb_fm,
    const float* __restrict__ d_dnn_in, long ld_dnn, const float* __restrict__ d_lin, long ld_lin,
    float* __restrict__ d_flat, const long* __restrict__ tab_off, const long* __restrict__ lin_off,
    unsigned char* __restrict__ marks, int npad, const int* __restrict__ dense_cols, int nd,
    float* __restrict__ d_dense_w, long dense_mark_base) {
    extern __shared__ __attribute__((aligned(16))) char sc_smem[];
    const int tid = threadIdx.x;

    if ((int)blockIdx.x >= m * nslice) {
        // ---- d(linear_model.weight)[k] += sum_b X[b][dense_cols[k]] * d_lin[b]: exact sum of the fp32 products ----
        const int k = blockIdx.x - m * nslice;
        const int col = dense_cols[k];
        float* red = reinterpret_cast<float*>(sc_smem);
        double* redd = reinterpret_cast<double*>(sc_smem + 64);
        constexpr int PER = SC_ROWS / SC_THREADS;
        float p[PER];
        float mx = 0.f;
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int bl = tid + t * SC_THREADS;
            const long b = b0 + (bl < nb ? bl : 0);
            const float x = X[b * ldx + col], g = d_lin[b * ld_lin];
            p[t] = bl < nb ? x * g : 0.f;
            mx = fmaxf(mx, fabsf(p[t]));
        }
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = 0.f;
#pragma unroll
        for (int t = 0; t < SC_THREADS / 64; ++t) mx = fmaxf(mx, red[t]);
        const int s = sc_scale_exp(mx);
        const double sc = sc_pow2(s);
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < PER; ++t) acc += sc_fix(p[t], sc);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);       // exact: any order gives the same bits
        if ((tid & 63) == 0) redd[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            acc = 0.0;
#pragma unroll
            for (int t = 0; t < SC_THREADS / 64; ++t) acc += redd[t];
            d_dense_w[k] += (float)(acc * sc_pow2(-s));
            if (marks) marks[(dense_mark_base + k) >> 2] = 1;
        }
        return;
    }

    const int j = blockIdx.x / nslice, slice = blockIdx.x - j * nslice;
    const int nwin = npad / SC_W;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(sc_smem);            // [SC_K(npad + 8) + 1]
    double* slotF = reinterpret_cast<double*>(sc_smem + sc_keys_bytes(npad));              // [nwin][17]: continuation pieces
    int* slotO = reinterpret_cast<int*>(slotF + (size_t)nwin * 17);                        // [nwin][17]: owner pieces
    // ---- S: keys and bitonic sort ------------------------------------------------------------
    {
        const int V = vocab[j];
        const int col = cols[j];
        for (int i = tid; i < npad + 8; i += SC_THREADS) {
            unsigned long long key = ~0ull;
            if (i < nb) {
                long id = (long)X[(long)(b0 + i) * ldx + col];                 // truncation as Tensor.long() (basemodel.py:369)
                id = id < 0 ? 0 : (id >= V ? V - 1 : id);                      // the gather raised the error flag for these
                key = ((unsigned long long)id << 32) | (unsigned)i;
            }
            keys[SC_K(i)] = key;
        }
        __syncthreads();
        // levels with compare distance >= 8 through LDS, one barrier each; distances 4, 2, 1 of a merge step inside
        // one thread's 8 consecutive elements
        for (int k = 2; k <= npad; k <<= 1) {
            for (int jj = k >> 1; jj >= 8; jj >>= 1) {
                for (int t = tid; t < (npad >> 1); t += SC_THREADS) {
                    const int lo = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), hi = lo | jj;
                    const bool up = (lo & k) == 0;
                    const unsigned long long a = keys[SC_K(lo)], b = keys[SC_K(hi)];
                    if ((a > b) == up) { keys[SC_K(lo)] = b; keys[SC_K(hi)] = a; }
                }
                __syncthreads();
            }
            for (int blk = tid; blk < (npad >> 3); blk += SC_THREADS) {
                unsigned long long e[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) e[q] = keys[blk * 9 + q];
#pragma unroll
                for (int jj = 4; jj > 0; jj >>= 1) {
                    if (jj < k) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            if ((q & jj) == 0) {
                                const bool up = ((blk * 8 + q) & k) == 0;
                                const unsigned long long a = e[q], b = e[q | jj];
                                const bool sw = (a > b) == up;
                                e[q] = sw ? b : a;
                                e[q | jj] = sw ? a : b;
                            }
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) keys[blk * 9 + q] = e[q];
            }
            __syncthreads();
        }
    }

    const int grp = tid >> 2, gl = tid & 3;
    constexpr int NGRP = SC_THREADS / 4;
    ScSrc S;
    S.de = d_emb_fm; S.dd = d_dnn_in; S.dl = d_lin; S.ld_dnn = ld_dnn; S.ld_lin = ld_lin; S.Btot = Btot;
    S.b0 = b0; S.j = j; S.D = D; S.c0 = slice * SC_SW + gl * 4;
    S.lin = d_lin != nullptr && lin_off != nullptr && slice == 0 && gl == 0;
    S.dd_vec = d_dnn_in != nullptr && (ld_dnn & 3) == 0 && ((((size_t)d_dnn_in) & 15) == 0);
    ScDst T;
    T.d_flat = d_flat; T.marks = marks; T.tab_base = tab_off ? tab_off[j] : -1; T.lin_base = lin_off ? lin_off[j] : -1;
    T.D = D; T.c0 = S.c0; T.lin = S.lin;
    const int sl = gl * 4;                                   // this lane's first slot component (component 16 = linear)

    // ---- 1: every window; closed runs are finished, open pieces leave their maxima in the slots -------------
    for (int w = grp; w * SC_W < nb; w += NGRP) {
        ScWin W;
        sc_window(keys, w, nb, W);
        float v[SC_W][SC_NC];
#pragma unroll
        for (int q = 0; q < SC_W; ++q) sc_load<VEC>(S, q < W.count ? W.b[q] : 0, v[q]);
        // piece maximum of every position: forward then backward over the run structure
        float pm[SC_W][SC_NC];
#pragma unroll
        for (int q = 0; q < SC_W; ++q)
#pragma unroll
            for (int c = 0; c < SC_NC; ++c) {
                const float a = q < W.count ? fabsf(v[q][c]) : 0.f;
                pm[q][c] = (q > 0 && W.id[q] == W.id[q - 1]) ? fmaxf(pm[q - 1][c], a) : a;
            }
#pragma unroll
        for (int q = SC_W - 2; q >= 0; --q)
#pragma unroll
            for (int c = 0; c < SC_NC; ++c)
                if (q + 1 < W.count && W.id[q] == W.id[q + 1]) pm[q][c] = pm[q + 1][c];
        double acc[SC_NC];
#pragma unroll
        for (int q = 0; q < SC_W; ++q) {
            if (q < W.count) {
                const bool first = q == 0 || W.id[q] != W.id[q - 1];
                const bool last = q == W.count - 1 || W.id[q] != W.id[q + 1];
                const bool pieceL = W.contL && W.id[q] == W.id[0];
                const bool pieceR = W.contR && W.id[q] == W.id[SC_W - 1];
                if (!pieceL && !pieceR) {
                    int s[SC_NC];
#pragma unroll
                    for (int c = 0; c < SC_NC; ++c) {
                        s[c] = sc_scale_exp(pm[q][c]);
                        const double f = sc_fix(v[q][c], sc_pow2(s[c]));
                        acc[c] = first ? f : acc[c] + f;
                    }
                    if (last) sc_store<VEC>(T, W.id[q], acc, s);
                } else if (last) {
                    // open piece: its maximum goes to the slot of its kind (a piece open on both sides is a middle one)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (pieceL) reinterpret_cast<float*>(slotF + (size_t)w * 17 + sl + c)[0] = pm[q][c];
                        else slotO[(size_t)w * 17 + sl + c] = __float_as_int(pm[q][c]);
                    }
                    if (S.lin) {
                        if (pieceL) reinterpret_cast<float*>(slotF + (size_t)w * 17 + 16)[0] = pm[q][4];
                        else slotO[(size_t)w * 17 + 16] = __float_as_int(pm[q][4]);
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- 2: owners (a run that starts in window w and goes on): run maximum -> exponent, handed to every piece ----
    for (int w = grp; w * SC_W < nb; w += NGRP) {
        ScWin W;
        sc_window(keys, w, nb, W);
        if (!W.contR || (W.contL && W.id[0] == W.id[SC_W - 1])) continue;
        float mx[SC_NC];
#pragma unroll
        for (int c = 0; c < 4; ++c) mx[c] = __int_as_float(slotO[(size_t)w * 17 + sl + c]);
        mx[4] = S.lin ? __int_as_float(slotO[(size_t)w * 17 + 16]) : 0.f;
        int we = w + 1;
        for (;; ++we) {
#pragma unroll
            for (int c = 0; c < 4; ++c) mx[c] = fmaxf(mx[c], reinterpret_cast<const float*>(slotF + (size_t)we * 17 + sl + c)[0]);
            if (S.lin) mx[4] = fmaxf(mx[4], reinterpret_cast<const float*>(slotF + (size_t)we * 17 + 16)[0]);
            if (!sc_is_middle(keys, we, nb)) break;
        }
        int s[SC_NC];
#pragma unroll
        for (int c = 0; c < SC_NC; ++c) s[c] = sc_scale_exp(mx[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) slotO[(size_t)w * 17 + sl + c] = s[c];
        if (S.lin) slotO[(size_t)w * 17 + 16] = s[4];
        for (int u = w + 1; u <= we; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) reinterpret_cast<int*>(slotF + (size_t)u * 17 + sl + c)[0] = s[c];
            if (S.lin) reinterpret_cast<int*>(slotF + (size_t)u * 17 + 16)[0] = s[4];
        }
    }
    __syncthreads();
    // ---- 3: continuation pieces: exact partial sums on the run's grid -> slots ----------------------------------
    for (int w = grp; w * SC_W < nb; w += NGRP) {
        ScWin W;
        sc_window(keys, w, nb, W);
        if (!W.contL) continue;
        int s[SC_NC];
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = reinterpret_cast<const int*>(slotF + (size_t)w * 17 + sl + c)[0];
        s[4] = S.lin ? reinterpret_cast<const int*>(slotF + (size_t)w * 17 + 16)[0] : 0;
        float v[SC_W][SC_NC];
#pragma unroll
        for (int q = 0; q < SC_W; ++q) sc_load<VEC>(S, (q < W.count && W.id[q] == W.id[0]) ? W.b[q] : W.b[0], v[q]);
        double acc[SC_NC] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < SC_W; ++q)
            if (q < W.count && W.id[q] == W.id[0])
#pragma unroll
                for (int c = 0; c < SC_NC; ++c) acc[c] += sc_fix(v[q][c], sc_pow2(s[c]));
#pragma unroll
        for (int c = 0; c < 4; ++c) slotF[(size_t)w * 17 + sl + c] = acc[c];
        if (S.lin) slotF[(size_t)w * 17 + 16] = acc[4];
    }
    __syncthreads();
    // ---- 4: owners: own piece + the partial sums of the windows the run goes through ------------------------------
    for (int w = grp; w * SC_W < nb; w += NGRP) {
        ScWin W;
        sc_window(keys, w, nb, W);
        if (!W.contR || (W.contL && W.id[0] == W.id[SC_W - 1])) continue;
        int s[SC_NC];
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = slotO[(size_t)w * 17 + sl + c];
        s[4] = S.lin ? slotO[(size_t)w * 17 + 16] : 0;
        const unsigned rid = W.id[SC_W - 1];
        float v[SC_W][SC_NC];
#pragma unroll
        for (int q = 0; q < SC_W; ++q) sc_load<VEC>(S, W.id[q] == rid ? W.b[q] : W.b[SC_W - 1], v[q]);
        double acc[SC_NC] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < SC_W; ++q)
            if (W.id[q] == rid)
#pragma unroll
                for (int c = 0; c < SC_NC; ++c) acc[c] += sc_fix(v[q][c], sc_pow2(s[c]));
        for (int we = w + 1;; ++we) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += slotF[(size_t)we * 17 + sl + c];
            if (S.lin) acc[4] += slotF[(size_t)we * 17 + 16];
            if (!sc_is_middle(keys, we, nb)) break;
        }
        sc_store<VEC>(T, rid, acc, s);
    }
}

extern "C" {

int xdfm_embed_gather_fwd(const float* X, long ldx, int B, const float* const* tables,
                          const float* const* lin_tables, const int* cols, const int* vocab, int m, int D,
                          const int* dense_cols, const float* dense_w, int nd, float* emb_fm, float* dnn_in,
                          float* lin_out, int* err_flag, void* stream) {
    XDFM_REQUIRE(X && tables && cols && vocab && emb_fm, "embed_gather_fwd: null pointer");
    XDFM_REQUIRE(B > 0 && m > 0 && D > 0 && nd >= 0 && ldx >= m + nd, "embed_gather_fwd: bad shape B=%d m=%d D=%d nd=%d ldx=%ld",
                 B, m, D, nd, ldx);
    XDFM_REQUIRE(nd == 0 || dense_cols, "embed_gather_fwd: dense_cols missing");
    XDFM_REQUIRE(!(lin_out && nd > 0) || dense_w, "embed_gather_fwd: dense_w missing");
    // examples per block: tile <= 32 KiB, at most 16, at least 1
    int EB = (int)(8192 / ((long)m * D));
    if (EB > 16) EB = 16;
    // A/B knob: 32 examples (2 KB runs per field, 3 workgroups per CU) measured 19 % slower at large batches
    if (xdfm_opt(OPT_DBG) & 1024) { EB = (int)(16384 / ((long)m * D)); if (EB > 32) EB = 32; }
    if ((xdfm_opt(OPT_DBG) & 256) && EB > 4) EB = 4;      // A/B: more, smaller workgroups
    if ((xdfm_opt(OPT_DBG) & 512) && EB > 8) EB = 8;
    if (EB < 1) EB = 1;
    // tile + linear values + row addresses (8 bytes each; the float part is kept even so that they are 8-byte
    // aligned) + staged dense values and weights
    const size_t lds = ((size_t)EB * m * D + (size_t)EB * m) * sizeof(float) + (size_t)EB * m * sizeof(void*) + 8 +
                       ((size_t)EB * nd + (size_t)nd) * sizeof(float);
    XDFM_REQUIRE(lds <= 160 * 1024, "embed_gather_fwd: m*D=%ld too large for one LDS tile", (long)m * D);
    XDFM_REQUIRE(EB <= EMB_THREADS, "embed_gather_fwd: internal");
    dim3 grid(ceil_div(B, EB));
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(V)                                                                                              \
    hipLaunchKernelGGL((embed_gather_kernel<V>), grid, dim3(EMB_THREADS), lds, st, X, ldx, B, tables, lin_tables, \
                       cols, vocab, m, D, dense_cols, dense_w, nd, EB, emb_fm, dnn_in, lin_out, err_flag)
    if (D % 4 == 0) LAUNCH(4);
    else if (D % 2 == 0) LAUNCH(2);
    else LAUNCH(1);
#undef LAUNCH
    return xdfm_check_launch("embed_gather_fwd");
}

int xdfm_embed_scatter_bwd(const float* X, long ldx, int B, const int* cols, const int* vocab, int m, int D,
                           const int* dense_cols, int nd, const float* d_emb_fm, const float* d_dnn_in,
                           const float* d_lin, float* d_flat, const long* tab_off, const long* lin_off,
                           float* d_dense_w, void* stream) {
    return xdfm_embed_scatter_bwd_marked(X, ldx, B, cols, vocab, m, D, dense_cols, nd, d_emb_fm, d_dnn_in, 0, d_lin, 0,
                                         d_flat, tab_off, lin_off, d_dense_w, nullptr, stream);
}

int xdfm_embed_scatter_bwd_marked(const float* X, long ldx, int B, const int* cols, const int* vocab, int m, int D,
                                  const int* dense_cols, int nd, const float* d_emb_fm, const float* d_dnn_in,
                                  long ld_dnn, const float* d_lin, long ld_lin, float* d_flat, const long* tab_off,
                                  const long* lin_off, float* d_dense_w, unsigned char* marks, void* stream) {
    XDFM_REQUIRE(X && cols && vocab, "embed_scatter_bwd: null pointer");
    if (ld_dnn <= 0) ld_dnn = (long)m * D + nd;
    if (ld_lin <= 0) ld_lin = 1;
    XDFM_REQUIRE(ld_dnn >= (long)m * D, "embed_scatter_bwd: ld_dnn %ld smaller than m*D", ld_dnn);
    XDFM_REQUIRE(!marks || (d_flat && (((size_t)d_flat) & 15) == 0), "embed_scatter_bwd: marks need a 16-byte aligned d_flat");
    XDFM_REQUIRE(!marks || !d_dense_w || d_dense_w >= d_flat, "embed_scatter_bwd: with marks d_dense_w must lie inside d_flat");
    XDFM_REQUIRE(d_flat || (!tab_off && !lin_off), "embed_scatter_bwd: offsets without a gradient buffer");
    XDFM_REQUIRE(B > 0 && m > 0 && D > 0 && nd >= 0, "embed_scatter_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const bool rows = tab_off || (d_lin && lin_off);
    const bool dense = nd > 0 && d_lin && d_dense_w;
    if (dense) XDFM_REQUIRE(dense_cols, "embed_scatter_bwd: dense_cols missing");
    if (!rows && !dense) return XDFM_OK;
    XDFM_REQUIRE(!tab_off || d_emb_fm || d_dnn_in, "embed_scatter_bwd: table offsets without row gradients");
    // float4 path: rows of D floats that start on 16-byte boundaries in every buffer the lanes touch
    const bool vec4 = D % 4 == 0 && ((((size_t)d_flat) | ((size_t)d_emb_fm)) & 15) == 0;
    const int nslice = rows ? ceil_div(D, SC_SW) : 0;
    // chunks of SC_ROWS examples, one launch each, in ascending order: a launch owns every row it writes (one
    // workgroup per field and column slice), launches of one stream run in order -- no two writers ever race
    for (long b0 = 0; b0 < B; b0 += SC_ROWS) {
        const int nb = (int)(B - b0 < SC_ROWS ? B - b0 : SC_ROWS);
        int npad = 8;
        while (npad < nb) npad <<= 1;
        const int nwin = npad / SC_W;
        const size_t lds = sc_keys_bytes(npad) + (size_t)nwin * 17 * (sizeof(double) + sizeof(int));
        const dim3 grid(m * nslice + (dense ? nd : 0));
        const long mark_base = marks && dense ? (long)(d_dense_w - d_flat) : 0L;
        if (vec4)
            hipLaunchKernelGGL((embed_scatter_sorted_kernel<4>), grid, dim3(SC_THREADS), lds, st, X, ldx, (int)b0, nb, (long)B,
                               cols, vocab, m, D, nslice, d_emb_fm, d_dnn_in, ld_dnn, d_lin, ld_lin, d_flat, tab_off, lin_off,
                               marks, npad, dense_cols, dense ? nd : 0, d_dense_w, mark_base);
        else
            hipLaunchKernelGGL((embed_scatter_sorted_kernel<1>), grid, dim3(SC_THREADS), lds, st, X, ldx, (int)b0, nb, (long)B,
                               cols, vocab, m, D, nslice, d_emb_fm, d_dnn_in, ld_dnn, d_lin, ld_lin, d_flat, tab_off, lin_off,
                               marks, npad, dense_cols, dense ? nd : 0, d_dense_w, mark_base);
        const int rc = xdfm_check_launch("embed_scatter_bwd");
        if (rc) return rc;
    }
    return XDFM_OK;
}

}  // extern "C"
