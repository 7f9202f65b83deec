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
// K2 backward: dense table gradients from the row gradients, WITHOUT atomics on the gradients -- a grouped,
// segmented, exact reduce.
//
// The reference's embedding_dense_backward (deepctr/inputs.py:168, sparse=False) adds the rows of one id in example
// order on the CPU: deterministic.  Arrival-order fp32 atomics are not (replicated tables of a row-parallel run
// drift apart in the last bits), and a hot id serialises thousands of atomics on one row.  Here one workgroup owns
// one (field, 4-column slice) -- or one field's linear table -- of a chunk of <= SC_ROWS examples:
//   G   group the chunk's examples by id in LDS: open-addressing hash slots (atomicCAS on integers), a count per
//       slot, an exclusive scan, a placement pass -- O(n); equal ids end up adjacent (a "run"), in no particular order;
//   1   one thread per window of 8 list positions: its 8 row pieces (float4) are loaded at once and stay in
//       registers; runs that lie inside the window are summed, and all their read-modify-writes of d_flat are
//       issued together;
//   2-4 runs that cross windows: their owner (the window where the run starts) collects the pieces through LDS.
// Every sum is EXACT: the addends of a run are rounded once to a fixed-point grid 2^-s derived from the run's
// largest magnitude (s = 37 - exponent: the grid is 2^-13 of an fp32 ulp of that magnitude or finer) and added as
// integers held in doubles (< 2^52, so every addition is exact), then rounded once to fp32.  The result is a pure
// function of the MULTISET of rows of an id -- which is why the order inside the grouped list (it depends on
// which lane wins a hash slot) does not matter: bit-identical from run to run, on every rank and under any
// permutation of the batch (tests/test_gpu_parity.py).  One row is written by exactly one lane (read-modify-write
// of d_flat, which holds zeros or the L2 gradient).
#define SC_THREADS 512
#define SC_ROWS 4096          // examples per chunk = SC_THREADS windows of SC_W
#define SC_W 8                // list positions per window
#define SC_SW 4               // embedding columns per workgroup slice (one float4 per row and lane)
#define SC_NC 4               // components per lane (a linear-table workgroup uses component 0 only)
#define SC_EMPTY 0xffffffffu

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte load from a 4-byte aligned address

__device__ __forceinline__ int sc_scale_exp(float amax) {
    // s with amax * 2^s in [2^37, 2^38): the sum of <= 2^13 such addends stays below 2^51
    const int E = (int)((__float_as_uint(amax) >> 23) & 0xff);          // biased exponent (0: zero / denormal)
    return 37 - ((E == 0 ? 1 : E) - 127);
}
__device__ __forceinline__ double sc_pow2(int s) { return __longlong_as_double((long long)(1023 + s) << 52); }
__device__ __forceinline__ double sc_fix(float v, double sc) { return __builtin_rint((double)v * sc); }

// LDS bytes: hash slots (2 per example: key + count) that the window slots reuse, the grouped list, scan scratch
__host__ __device__ inline size_t sc_lds_bytes(int npad) {
    const size_t hash = (size_t)2 * npad * 2 * sizeof(unsigned);
    const size_t slots = (size_t)(npad / SC_W) * 2 * (SC_NC * (sizeof(double) + sizeof(float)) + 2 * sizeof(int));   // open pieces: 2 per window
    return (hash > slots ? hash : slots) + (size_t)(npad + 8) * (sizeof(unsigned) + sizeof(unsigned short)) + 64 +
           (size_t)npad * 4 * sizeof(float);                   // + the staged row pieces [npad][4]
}

template <int VEC>
__global__ __launch_bounds__(SC_THREADS) void embed_scatter_grouped_kernel(
    const float* __restrict__ X, long ldx, int b0, int nb, long Btot, const int* __restrict__ cols,
    const int* __restrict__ vocab, int m, int D, int nslice, const float* __restrict__ d_emb_fm,
    const float* __restrict__ d_dnn_in, long ld_dnn, const float* __restrict__ d_lin, long ld_lin,
    float* __restrict__ d_flat, const long* __restrict__ tab_off, const long* __restrict__ lin_off,
    unsigned char* __restrict__ marks, int npad, int nlin, const int* __restrict__ dense_cols, int nd,
    float* __restrict__ d_dense_w, long dense_mark_base, int dbg) {
    // dbg: timing experiments only (xdfm option "dbg" bits 12..16; results become wrong): 1 = no grouping,
    // 2 = no row loads, 4 = no read-modify-writes, 8 = no cross-window phases, 16 = ids without reading X
    extern __shared__ __attribute__((aligned(16))) char sc_smem[];
    const int tid = threadIdx.x;

    if ((int)blockIdx.x >= m * nslice + nlin) {
        // ---- d(linear_model.weight)[k] += sum_b X[b][dense_cols[k]] * d_lin[b]: exact sum of the fp32 products ----
        const int k = blockIdx.x - m * nslice - nlin;
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

    const bool lin = (int)blockIdx.x >= m * nslice;                                        // linear-table workgroup
    const int j = lin ? (int)blockIdx.x - m * nslice : (int)blockIdx.x / nslice;
    const int slice = lin ? 0 : (int)blockIdx.x - j * nslice;
    const int nwin = npad / SC_W;                                                          // <= SC_THREADS
    const int HS = 2 * npad;                                                               // hash slots (load <= 0.5)
    unsigned* tkey = reinterpret_cast<unsigned*>(sc_smem);                                 // [HS]
    unsigned* tcnt = tkey + HS;                                                            // [HS]
    // after grouping the same LDS holds the OPEN pieces, two elements per window (2w: the window's first piece when it
    // continues a run from the left; 2w+1: its last piece when the run goes on to the right)
    double* seqS = reinterpret_cast<double*>(sc_smem);                                     // [2*nwin][4] partial sums
    float* seqM = reinterpret_cast<float*>(seqS + (size_t)2 * nwin * SC_NC);               // [2*nwin][4] maxima
    int* seqN = reinterpret_cast<int*>(seqM + (size_t)2 * nwin * SC_NC);                   // [2*nwin] next element of the run, -1 = last
    int* seqH = seqN + 2 * nwin;                                                           // [2*nwin] head element of the run
    const size_t hash_b = (size_t)HS * 2 * sizeof(unsigned), slot_b = (size_t)nwin * 2 * (SC_NC * 12 + 8);
    unsigned* gid = reinterpret_cast<unsigned*>(sc_smem + (hash_b > slot_b ? hash_b : slot_b));   // [npad + 8] grouped ids
    unsigned short* gb = reinterpret_cast<unsigned short*>(gid + npad + 8);                        // [npad + 8] example in chunk
    unsigned* wsum = reinterpret_cast<unsigned*>(gb + npad + 8);                                   // [8] scan scratch
    float4* tile = reinterpret_cast<float4*>(reinterpret_cast<char*>(wsum) + 64);                 // [npad] row pieces by example

    const int c0 = slice * SC_SW;                              // columns c0 .. c0+3 (those < D are live)
    // Row pieces of the chunk's examples in EXAMPLE order (thread t takes examples t, t + 512, ..): issued first, so
    // that they are in flight while the ids are grouped; they are parked in LDS afterwards and every window picks its
    // 8 pieces from there.  d_emb_fm + d_dnn_in is one fp32 add, as autograd's accumulation of the two uses of the
    // embedding.  Unconditional loads from clamped addresses, masked afterwards.
    constexpr int KPT = SC_ROWS / SC_THREADS;                  // examples per thread
    float4 pe[KPT], pd[KPT];
    {
        const bool skip = (dbg & 2) != 0;
#pragma unroll
        for (int n = 0; n < KPT; ++n) { pe[n] = make_float4(0.f, 0.f, 0.f, 0.f); pd[n] = pe[n]; }
        if (lin) {
            if (!skip) {
#pragma unroll
                for (int n = 0; n < KPT; ++n) {
                    const int i = tid + n * SC_THREADS;
                    pe[n].x = d_lin[(long)(b0 + (i < nb ? i : 0)) * ld_lin];
                }
            }
        } else {
            if (d_emb_fm && !skip) {
#pragma unroll
                for (int n = 0; n < KPT; ++n) {
                    const int i = tid + n * SC_THREADS;
                    const float* src = d_emb_fm + ((long)j * Btot + b0 + (i < nb ? i : 0)) * D;
                    if (VEC == 4) pe[n] = *reinterpret_cast<const float4*>(src + c0);
                    else pe[n] = make_float4(src[c0 < D ? c0 : D - 1], src[c0 + 1 < D ? c0 + 1 : D - 1],
                                             src[c0 + 2 < D ? c0 + 2 : D - 1], src[c0 + 3 < D ? c0 + 3 : D - 1]);
                }
            }
            if (d_dnn_in && !skip) {
#pragma unroll
                for (int n = 0; n < KPT; ++n) {
                    const int i = tid + n * SC_THREADS;
                    const float* src = d_dnn_in + (long)(b0 + (i < nb ? i : 0)) * ld_dnn + (long)j * D;
                    if (VEC == 4) { const f4u t = *reinterpret_cast<const f4u*>(src + c0); pd[n] = make_float4(t.x, t.y, t.z, t.w); }   // rows are only 4-byte aligned (ld = m*D + nd)
                    else pd[n] = make_float4(src[c0 < D ? c0 : D - 1], src[c0 + 1 < D ? c0 + 1 : D - 1],
                                             src[c0 + 2 < D ? c0 + 2 : D - 1], src[c0 + 3 < D ? c0 + 3 : D - 1]);
                }
            }
        }
    }

    // ---- G: group the examples by id ---------------------------------------------------------------------------
    {
        const int V = vocab[j];
        const int col = cols[j];
        float xv[KPT];
#pragma unroll
        for (int n = 0; n < KPT; ++n) {
            const int i = tid + n * SC_THREADS;
            xv[n] = (dbg & 16) ? (float)((i * 7) % V) : X[(long)(b0 + (i < nb ? i : 0)) * ldx + col];
        }
        for (int h = tid; h < HS; h += SC_THREADS) { tkey[h] = SC_EMPTY; tcnt[h] = 0; }
        for (int i = nb + tid; i < npad + 8; i += SC_THREADS) { gid[i] = SC_EMPTY; gb[i] = 0; }
        __syncthreads();
        unsigned ids[KPT];
        int slot[KPT];
        const int hshift = 32 - (31 - __builtin_clz(HS));
        unsigned hh[KPT], step[KPT];
        bool todo[KPT];
#pragma unroll
        for (int n = 0; n < KPT; ++n) {
            const int i = tid + n * SC_THREADS;
            long id = (long)xv[n];                                     // truncation as Tensor.long() (basemodel.py:369)
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);                  // the gather raised the error flag for these
            ids[n] = (unsigned)id;
            slot[n] = 0;
            hh[n] = (ids[n] * 2654435761u) >> hshift;
            step[n] = ((ids[n] * 0x85ebca6bu) >> 15) | 1u;             // double hashing: odd step, HS is a power of two
            todo[n] = i < nb && !(dbg & 1);
            if (i < nb && (dbg & 1)) { gid[i] = ids[n]; gb[i] = (unsigned short)i; }
        }
        // A slot is claimed by integer compare-and-swap.  Branch-free rounds: the thread's 8 probes are issued
        // together (8 LDS atomics in flight, one wait); a probe that has found its slot repeats a compare-and-swap that
        // cannot change anything (its slot already holds its id).  Double hashing: 2-3 rounds on average, the wave
        // goes on until its slowest lane is done; at most HS rounds.
        if (!(dbg & 1)) {
            for (;;) {
                unsigned prev[KPT];
#pragma unroll
                for (int n = 0; n < KPT; ++n) prev[n] = atomicCAS(&tkey[hh[n]], SC_EMPTY, ids[n]);
                bool any = false;
#pragma unroll
                for (int n = 0; n < KPT; ++n) {
                    const bool ok = prev[n] == SC_EMPTY || prev[n] == ids[n];
                    hh[n] = ok ? hh[n] : ((hh[n] + step[n]) & (HS - 1));
                    any = any || !ok;
                }
                if (!__any(any)) break;
            }
#pragma unroll
            for (int n = 0; n < KPT; ++n) {
                slot[n] = (int)hh[n];
                atomicAdd(&tcnt[hh[n]], (tid + n * SC_THREADS < nb) ? 1u : 0u);      // an example past the chunk claims nothing
            }
        }
        (void)todo;
        __syncthreads();
        if (!(dbg & 1)) {
            // exclusive scan of the counts over the slots: PER consecutive slots per thread, wave scan, 8 wave totals
            const int PER = HS >= SC_THREADS ? HS / SC_THREADS : 1;
            const int s0 = tid * PER;
            unsigned loc = 0;
            if (s0 < HS)
                for (int q = 0; q < PER; ++q) loc += tcnt[s0 + q];
            unsigned inc = loc;
            const int lane = tid & 63;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            if (lane == 63) wsum[tid >> 6] = inc;
            __syncthreads();
            unsigned basew = 0;
            for (int q = 0; q < (tid >> 6); ++q) basew += wsum[q];
            unsigned run = basew + inc - loc;
            if (s0 < HS)
                for (int q = 0; q < PER; ++q) { const unsigned c = tcnt[s0 + q]; tcnt[s0 + q] = run; run += c; }
            __syncthreads();
            unsigned pos[KPT];
#pragma unroll
            for (int n = 0; n < KPT; ++n) pos[n] = atomicAdd(&tcnt[slot[n]], (tid + n * SC_THREADS < nb) ? 1u : 0u);
#pragma unroll
            for (int n = 0; n < KPT; ++n) {
                const int i = tid + n * SC_THREADS;
                if (i < nb) { gid[pos[n]] = ids[n]; gb[pos[n]] = (unsigned short)i; }
            }
        }
        __syncthreads();
    }

    // ---- this thread's window ----------------------------------------------------------------
    const int w = tid;
    const int base = w * SC_W;
    const bool live = base < nb;
    const int count = live ? (nb - base < SC_W ? nb - base : SC_W) : 0;
    unsigned id[SC_W];
    int bl[SC_W];
    {
        const int bs = live ? base : 0;
        const uint4 i0 = *reinterpret_cast<const uint4*>(gid + bs), i1 = *reinterpret_cast<const uint4*>(gid + bs + 4);
        const uint4 bb = *reinterpret_cast<const uint4*>(gb + bs);
        const unsigned iv[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
        const unsigned bv[8] = {bb.x & 0xffffu, bb.x >> 16, bb.y & 0xffffu, bb.y >> 16, bb.z & 0xffffu, bb.z >> 16, bb.w & 0xffffu, bb.w >> 16};
#pragma unroll
        for (int q = 0; q < SC_W; ++q) {
            id[q] = live ? iv[q] : SC_EMPTY;
            bl[q] = q < count ? (int)bv[q] : 0;
        }
    }
    const bool contL = live && base > 0 && gid[base - 1] == id[0];
    const bool contR = count == SC_W && base + SC_W < nb && gid[base + SC_W] == id[SC_W - 1];
    const bool single = id[0] == id[SC_W - 1];
    const bool owner = contR && !(contL && single);            // a run starts here and goes on into the next window
    // park the row pieces (the loads issued at the top have had the whole grouping phase to land)
#pragma unroll
    for (int n = 0; n < KPT; ++n) {
        const int i = tid + n * SC_THREADS;
        if (i < npad) tile[i] = make_float4(pe[n].x + pd[n].x, pe[n].y + pd[n].y, pe[n].z + pd[n].z, pe[n].w + pd[n].w);
    }
    const bool any_open = __syncthreads_or(contR) != 0;        // also: tile complete; ids read before the slots (same LDS as the hash table) are written

    const long row_base = lin ? (lin_off ? lin_off[j] : -1) : (tab_off ? tab_off[j] : -1);
    const bool vrow = !lin && VEC == 4 && row_base >= 0 && ((row_base & 3) == 0);
    // old values of the rows this window may finish: unconditional loads (every id of a live window is a valid row;
    // dead positions read row 0), issued before the arithmetic so that they are back when the sums are ready
    float4 old[SC_W];
#pragma unroll
    for (int q = 0; q < SC_W; ++q) old[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_base >= 0 && !(dbg & 4)) {
        if (vrow) {
#pragma unroll
            for (int q = 0; q < SC_W; ++q)
                old[q] = *reinterpret_cast<const float4*>(d_flat + row_base + (long)(q < count ? id[q] : 0u) * D + c0);
        } else if (lin) {
#pragma unroll
            for (int q = 0; q < SC_W; ++q) old[q].x = d_flat[row_base + (long)(q < count ? id[q] : 0u)];
        }
    }
    float v[SC_W][SC_NC];
#pragma unroll
    for (int q = 0; q < SC_W; ++q) {
        const float4 t = tile[bl[q]];
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) v[q][c] = (q < count && (lin ? c == 0 : c0 + c < D)) ? tv[c] : 0.f;
    }

    // ---- 1: piece maxima (forward, then backward over the run structure); closed runs are finished -----------
    float pm[SC_W][SC_NC];
#pragma unroll
    for (int q = 0; q < SC_W; ++q)
#pragma unroll
        for (int c = 0; c < SC_NC; ++c) {
            const float a = fabsf(v[q][c]);
            pm[q][c] = (q > 0 && id[q] == id[q - 1]) ? fmaxf(pm[q - 1][c], a) : a;
        }
#pragma unroll
    for (int q = SC_W - 2; q >= 0; --q)
#pragma unroll
        for (int c = 0; c < SC_NC; ++c)
            if (q + 1 < count && id[q] == id[q + 1]) pm[q][c] = pm[q + 1][c];
    float res[SC_W][SC_NC];                     // finished sums, valid where fin[q]
    bool fin[SC_W];
    {
        double acc[SC_NC];
#pragma unroll
        for (int q = 0; q < SC_W; ++q) {
            const bool first = q == 0 || id[q] != id[q - 1];
            const bool last = q == count - 1 || (q < count && id[q] != id[q + 1 < SC_W ? q + 1 : q]);
            const bool pieceL = contL && id[q] == id[0];
            const bool pieceR = contR && id[q] == id[SC_W - 1];
            fin[q] = q < count && last && !pieceL && !pieceR && !(dbg & 4);
#pragma unroll
            for (int c = 0; c < SC_NC; ++c) {
                const int s = sc_scale_exp(pm[q][c]);
                const double f = sc_fix(v[q][c], sc_pow2(s));
                acc[c] = first ? f : acc[c] + f;
                res[q][c] = (float)(acc[c] * sc_pow2(-s));
            }
            if (any_open && q < count && last && (pieceL || pieceR)) {
                // open piece: its maximum goes to its element (a piece open on both sides uses the window's first element)
#pragma unroll
                for (int c = 0; c < SC_NC; ++c) seqM[(size_t)(2 * w + (pieceL ? 0 : 1)) * SC_NC + c] = pm[q][c];
            }
        }
    }
    // the read-modify-writes of the window (their loads were issued above; one writer per row: no hazard)
    if (row_base >= 0) {
#pragma unroll
        for (int q = 0; q < SC_W; ++q) {
            if (fin[q]) {
                if (vrow) {
                    const long e = row_base + (long)id[q] * D + c0;
                    *reinterpret_cast<float4*>(d_flat + e) = make_float4(old[q].x + res[q][0], old[q].y + res[q][1],
                                                                         old[q].z + res[q][2], old[q].w + res[q][3]);
                    if (marks) marks[e >> 2] = 1;
                } else if (lin) {
                    const long e = row_base + (long)id[q];
                    d_flat[e] = old[q].x + res[q][0];
                    if (marks) marks[e >> 2] = 1;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c0 + c < D) {
                            const long e = row_base + (long)id[q] * D + c0 + c;
                            d_flat[e] += res[q][c];
                            if (marks) marks[e >> 2] = 1;
                        }
                }
            }
        }
    }
    if (!any_open || (dbg & 8)) return;         // no run crosses a window: done (uniform over the workgroup)
    // ---- 2-4: runs that cross windows, in parallel over their pieces -----------------------------------------------
    // A run's open pieces form a chain of elements e -> seqN[e]: (2w+1) -> 2(w+1) when the run leaves window w, and
    // 2w -> 2w+1 inside a window that is one single run passing through.  Pointer jumping (log2 of the longest chain
    // rounds, all chains at once) gives every piece the run's maximum -- hence its grid -- and then the head of the
    // chain, which is the window where the run starts, the exact sum of the pieces.  (A serial walk by the owner cost
    // 150 us on a field with 3 ids: runs of 3000 examples = 350 windows.)
    const bool both = contL && contR && single;              // one piece, open on both sides: element 2w, linked to 2w+1
    const bool act = w < nwin;                               // threads past the chunk's windows own no elements
    const int e0 = act ? 2 * w : 0, e1 = act ? 2 * w + 1 : 0;
    if (act) {
        const bool hasL = live && contL, hasR = live && contR && !both;
        if (!hasL) {
#pragma unroll
            for (int c = 0; c < SC_NC; ++c) seqM[(size_t)e0 * SC_NC + c] = 0.f;
        }
        if (!hasR) {
#pragma unroll
            for (int c = 0; c < SC_NC; ++c) seqM[(size_t)e1 * SC_NC + c] = 0.f;
        }
        seqN[e0] = both ? e1 : -1;
        seqN[e1] = (live && contR) ? e1 + 1 : -1;
        // head pointers start at the left neighbour in the chain (or at the element itself: a head)
        seqH[e0] = (live && contL) ? e0 - 1 : e0;
        seqH[e1] = both ? e0 : e1;
    }
    __syncthreads();
    // suffix maxima along the chains + head of every element, by pointer jumping (reads, barrier, writes, barrier);
    // one 16-byte LDS access per element (scalar accesses at this stride are 8-way bank conflicts)
    float4* seqM4 = reinterpret_cast<float4*>(seqM);
    for (;;) {
        const int n0 = act ? seqN[e0] : -1, n1 = act ? seqN[e1] : -1, h0 = seqH[e0], h1 = seqH[e1];
        const int nn0 = n0 >= 0 ? seqN[n0] : -1, nn1 = n1 >= 0 ? seqN[n1] : -1;
        const int hh0 = seqH[h0], hh1 = seqH[h1];
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 m0 = n0 >= 0 ? seqM4[n0] : z4, m1 = n1 >= 0 ? seqM4[n1] : z4;
        const float4 o0 = seqM4[e0], o1 = seqM4[e1];
        const bool more = act && (n0 >= 0 || n1 >= 0 || hh0 != h0 || hh1 != h1);
        if (!__syncthreads_or(more)) break;
        if (act) {
            seqM4[e0] = make_float4(fmaxf(o0.x, m0.x), fmaxf(o0.y, m0.y), fmaxf(o0.z, m0.z), fmaxf(o0.w, m0.w));
            seqM4[e1] = make_float4(fmaxf(o1.x, m1.x), fmaxf(o1.y, m1.y), fmaxf(o1.z, m1.z), fmaxf(o1.w, m1.w));
            seqN[e0] = nn0; seqN[e1] = nn1; seqH[e0] = hh0; seqH[e1] = hh1;
        }
        __syncthreads();
    }
    // every open piece: the run's grid from the maximum at the chain's head; exact partial sum of the piece
    int sL[SC_NC], sR[SC_NC];
    {
        const int h0 = seqH[e0], h1 = seqH[e1];
        double aL[SC_NC], aR[SC_NC];
#pragma unroll
        for (int c = 0; c < SC_NC; ++c) {
            sL[c] = sc_scale_exp(seqM[(size_t)h0 * SC_NC + c]);
            sR[c] = sc_scale_exp(seqM[(size_t)h1 * SC_NC + c]);
            aL[c] = 0.0; aR[c] = 0.0;
        }
#pragma unroll
        for (int q = 0; q < SC_W; ++q) {
            const bool inL = contL && q < count && id[q] == id[0];
            const bool inR = contR && !both && id[q] == id[SC_W - 1];
#pragma unroll
            for (int c = 0; c < SC_NC; ++c) {
                if (inL) aL[c] += sc_fix(v[q][c], sc_pow2(sL[c]));
                if (inR) aR[c] += sc_fix(v[q][c], sc_pow2(sR[c]));
            }
        }
        if (act) {
#pragma unroll
            for (int c = 0; c < SC_NC; ++c) {
                seqS[(size_t)e0 * SC_NC + c] = aL[c];
                seqS[(size_t)e1 * SC_NC + c] = aR[c];
            }
            seqN[e0] = both ? e1 : -1;                       // the chains again (the jumping above consumed them)
            seqN[e1] = (live && contR) ? e1 + 1 : -1;
        }
    }
    __syncthreads();
    double2* seqS2 = reinterpret_cast<double2*>(seqS);
    for (;;) {                                                // suffix sums along the chains: integer-valued doubles, exact
        const int n0 = act ? seqN[e0] : -1, n1 = act ? seqN[e1] : -1;
        const int nn0 = n0 >= 0 ? seqN[n0] : -1, nn1 = n1 >= 0 ? seqN[n1] : -1;
        const double2 z2 = make_double2(0.0, 0.0);
        const double2 a0l = n0 >= 0 ? seqS2[2 * n0] : z2, a0h = n0 >= 0 ? seqS2[2 * n0 + 1] : z2;
        const double2 a1l = n1 >= 0 ? seqS2[2 * n1] : z2, a1h = n1 >= 0 ? seqS2[2 * n1 + 1] : z2;
        const double2 o0l = seqS2[2 * e0], o0h = seqS2[2 * e0 + 1], o1l = seqS2[2 * e1], o1h = seqS2[2 * e1 + 1];
        if (!__syncthreads_or(n0 >= 0 || n1 >= 0)) break;
        if (act) {
            seqS2[2 * e0] = make_double2(o0l.x + a0l.x, o0l.y + a0l.y); seqS2[2 * e0 + 1] = make_double2(o0h.x + a0h.x, o0h.y + a0h.y);
            seqS2[2 * e1] = make_double2(o1l.x + a1l.x, o1l.y + a1l.y); seqS2[2 * e1 + 1] = make_double2(o1h.x + a1h.x, o1h.y + a1h.y);
            seqN[e0] = nn0; seqN[e1] = nn1;
        }
        __syncthreads();
    }
    // the head of a chain is the last piece of the window where the run starts: it writes the row
    if (owner && row_base >= 0) {
        const unsigned rid = id[SC_W - 1];
        float r[SC_NC];
#pragma unroll
        for (int c = 0; c < SC_NC; ++c) r[c] = (float)(seqS[(size_t)e1 * SC_NC + c] * sc_pow2(-sR[c]));
        if (lin) {
            const long e = row_base + (long)rid;
            d_flat[e] += r[0];
            if (marks) marks[e >> 2] = 1;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c0 + c < D) {
                    const long e = row_base + (long)rid * D + c0 + c;
                    d_flat[e] += r[c];
                    if (marks) marks[e >> 2] = 1;
                }
        }
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
    const bool tabs = tab_off != nullptr;
    const bool lins = d_lin && lin_off;
    const bool dense = nd > 0 && d_lin && d_dense_w;
    if (dense) XDFM_REQUIRE(dense_cols, "embed_scatter_bwd: dense_cols missing");
    if (!tabs && !lins && !dense) return XDFM_OK;
    XDFM_REQUIRE(!tabs || d_emb_fm || d_dnn_in, "embed_scatter_bwd: table offsets without row gradients");
    // float4 path: rows of D floats that start on 16-byte boundaries in d_flat and d_emb_fm (the kernel checks the
    // table offsets, which live on the device, lane by lane)
    const bool vec4 = D % 4 == 0 && ((((size_t)d_flat) | ((size_t)d_emb_fm)) & 15) == 0 && !(xdfm_opt(OPT_DBG) & 2048);
    const int nslice = tabs ? ceil_div(D, SC_SW) : 0;
    const int nlin = lins ? m : 0;
    // chunks of SC_ROWS examples, one launch each, in ascending order: a launch owns every row it writes (one
    // workgroup per field and column slice), launches of one stream run in order -- no two writers ever race
    for (long b0 = 0; b0 < B; b0 += SC_ROWS) {
        const int nb = (int)(B - b0 < SC_ROWS ? B - b0 : SC_ROWS);
        int npad = 8;
        while (npad < nb) npad <<= 1;
        const size_t lds = sc_lds_bytes(npad);
        const dim3 grid(m * nslice + nlin + (dense ? nd : 0));
        const long mark_base = marks && dense ? (long)(d_dense_w - d_flat) : 0L;
        const int dbg = (xdfm_opt(OPT_DBG) >> 12) & 31;
        if (vec4)
            hipLaunchKernelGGL((embed_scatter_grouped_kernel<4>), grid, dim3(SC_THREADS), lds, st, X, ldx, (int)b0, nb, (long)B,
                               cols, vocab, m, D, nslice, d_emb_fm, d_dnn_in, ld_dnn, d_lin, ld_lin, d_flat, tab_off, lin_off,
                               marks, npad, nlin, dense_cols, dense ? nd : 0, d_dense_w, mark_base, dbg);
        else
            hipLaunchKernelGGL((embed_scatter_grouped_kernel<1>), grid, dim3(SC_THREADS), lds, st, X, ldx, (int)b0, nb, (long)B,
                               cols, vocab, m, D, nslice, d_emb_fm, d_dnn_in, ld_dnn, d_lin, ld_lin, d_flat, tab_off, lin_off,
                               marks, npad, nlin, dense_cols, dense ? nd : 0, d_dense_w, mark_base, dbg);
        const int rc = xdfm_check_launch("embed_scatter_bwd");
        if (rc) return rc;
    }
    return XDFM_OK;
}

}  // extern "C"
