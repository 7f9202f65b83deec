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
// backward: one thread per (b, j, d): atomic add of the row gradient into the dense table grad.
__global__ __launch_bounds__(256) void embed_scatter_kernel(
    const float* __restrict__ X, long ldx, int B, const int* __restrict__ cols, const int* __restrict__ vocab, int m,
    int D, int nd, const float* __restrict__ d_emb_fm, const float* __restrict__ d_dnn_in,
    const float* __restrict__ d_lin, float* __restrict__ d_flat, const long* __restrict__ tab_off,
    const long* __restrict__ lin_off, unsigned char* __restrict__ marks, long ld_dnn, long ld_lin) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * m * D;
    if (idx >= total) return;
    const int d = (int)(idx % D);
    const long r = idx / D;
    const int j = (int)(r % m);
    const int b = (int)(r / m);
    long id = (long)X[(long)b * ldx + cols[j]];
    const int V = vocab[j];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    // both gradient sources are loaded unconditionally (a missing one aliases X and is masked): behind the two
    // conditions the loads were issued one after the other
    const float* pe = d_emb_fm ? d_emb_fm + ((long)j * B + b) * D + d : X;
    const float* pd = d_dnn_in ? d_dnn_in + (long)b * ld_dnn + (long)j * D + d : X;
    const float ge = *pe, gd = *pd;
    const float g = (d_emb_fm ? ge : 0.f) + (d_dnn_in ? gd : 0.f);
    // marks: one byte per 16-byte chunk of d_flat, set where a gradient landed (several threads may store the
    // same 1) -- K7 then reads, and re-zeroes, only the marked chunks (xdfm_adam_tensor.grad_marks)
    if (tab_off) {
        const long e = tab_off[j] + id * D + d;
        atomicAdd(d_flat + e, g);
        if (marks) marks[e >> 2] = 1;
    }
    if (d == 0 && d_lin && lin_off) {
        const long e = lin_off[j] + id;
        atomicAdd(d_flat + e, d_lin[(long)b * ld_lin]);
        if (marks) marks[e >> 2] = 1;
    }
}

// d(linear_model.weight)[k] += sum_b X[b][dense_cols[k]] * d_lin[b]
__global__ __launch_bounds__(256) void dense_w_grad_kernel(const float* __restrict__ X, long ldx, int B,
                                                          const int* __restrict__ dense_cols, int nd,
                                                          const float* __restrict__ d_lin,
                                                          float* __restrict__ d_dense_w,
                                                          unsigned char* __restrict__ marks, long mark_base,
                                                          long ld_lin) {
    const int k = blockIdx.y;
    const int col = dense_cols[k];
    float part = 0.f;
    for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (long)gridDim.x * blockDim.x)
        part += X[b * ldx + col] * d_lin[b * ld_lin];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&d_dense_w[k], wsum[0] + wsum[1] + wsum[2] + wsum[3]);
        if (marks) marks[(mark_base + k) >> 2] = 1;
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
    if (tab_off || (d_lin && lin_off)) {
        const long total = (long)B * m * D;
        hipLaunchKernelGGL(embed_scatter_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, X, ldx, B, cols, vocab,
                           m, D, nd, d_emb_fm, d_dnn_in, d_lin, d_flat, tab_off, lin_off, marks, ld_dnn, ld_lin);
        int rc = xdfm_check_launch("embed_scatter_bwd");
        if (rc) return rc;
    }
    if (nd > 0 && d_lin && d_dense_w) {
        XDFM_REQUIRE(dense_cols, "embed_scatter_bwd: dense_cols missing");
        int gx = ceil_div(B, 256);
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL(dense_w_grad_kernel, dim3(gx, nd), dim3(256), 0, st, X, ldx, B, dense_cols, nd, d_lin,
                           d_dense_w, marks, marks ? (long)(d_dense_w - d_flat) : 0L, ld_lin);
        return xdfm_check_launch("embed_scatter_bwd dense_w");
    }
    return XDFM_OK;
}

}  // extern "C"
