// K6: multi-tensor L2 regulariser.
//
// replaces the Python loop of deepctr/models/basemodel.py:412-428, which issues square / mul /
// sum / add per regularised tensor (~58 tensors incl. every embedding table in full: >200 launches
// forward and more in backward) with one reduction launch + one gradient launch.
//
//   value  = sum_t coeff[t] * sum_i w_t[i]^2                (forward)
//   g_t[i] = 2 * coeff[t] * gscale * w_t[i]                 (backward, gscale = d loss / d value)
//
// Deterministic: every tensor is reduced by a fixed grid of REG_BLOCKS blocks into partials that a
// single block sums in a fixed order.
#include "xdfm_internal.h"

#define REG_BLOCKS 32
#define REG_THREADS 256

__global__ __launch_bounds__(REG_THREADS) void l2_sumsq_kernel(const float* const* __restrict__ ptrs,
                                                               const long* __restrict__ numel,
                                                               float* __restrict__ partials) {
    const int t = blockIdx.y;
    const float* __restrict__ w = ptrs[t];
    const long n = numel[t];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const long stride = (long)REG_BLOCKS * REG_THREADS;
    const long tid = (long)blockIdx.x * REG_THREADS + threadIdx.x;
    const bool vec = (((size_t)w) & 15) == 0;
    const long n4 = vec ? n / 4 : 0;
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(w);
    for (long i = tid; i < n4; i += stride) {
        const float4 v = w4[i];
        a0 = fmaf(v.x, v.x, a0); a1 = fmaf(v.y, v.y, a1); a2 = fmaf(v.z, v.z, a2); a3 = fmaf(v.w, v.w, a3);
    }
    for (long i = 4 * n4 + tid; i < n; i += stride) { const float v = w[i]; a0 = fmaf(v, v, a0); }
    float part = (a0 + a1) + (a2 + a3);
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    __shared__ float wsum[REG_THREADS / 64];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) partials[t * REG_BLOCKS + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(REG_THREADS) void l2_finish_kernel(const float* __restrict__ partials,
                                                                const float* __restrict__ coeff, int T,
                                                                float* __restrict__ out) {
    // thread k sums tensor k's partials in index order; then a fixed-order tree over tensors
    __shared__ float acc[REG_THREADS];
    float v = 0.f;
    for (int t = threadIdx.x; t < T; t += REG_THREADS) {
        float s = 0.f;
        for (int b = 0; b < REG_BLOCKS; ++b) s += partials[t * REG_BLOCKS + b];
        v += coeff[t] * s;
    }
    acc[threadIdx.x] = v;
    __syncthreads();
    for (int o = REG_THREADS / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) acc[threadIdx.x] += acc[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = acc[0];
}

__global__ __launch_bounds__(REG_THREADS) void l2_grad_kernel(const float* const* __restrict__ ptrs,
                                                              const long* __restrict__ numel,
                                                              const float* __restrict__ coeff,
                                                              const float* __restrict__ gscale,
                                                              float* __restrict__ gflat,
                                                              const long* __restrict__ goff, int accumulate) {
    const int t = blockIdx.y;
    const float* __restrict__ w = ptrs[t];
    float* __restrict__ g = gflat + goff[t];
    const long n = numel[t];
    const float sc = 2.f * coeff[t] * gscale[0];
    const long tid = (long)blockIdx.x * REG_THREADS + threadIdx.x;
    const long nthr = (long)gridDim.x * REG_THREADS;
    // 16-byte path when both streams are aligned (embedding tables are); scalar tail / fallback
    const bool vec = ((((size_t)w) | ((size_t)g)) & 15) == 0;
    const long n4 = vec ? n / 4 : 0;
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(w);
    float4* __restrict__ g4 = reinterpret_cast<float4*>(g);
    for (long i = tid; i < n4; i += nthr) {
        const float4 v = w4[i];
        float4 r;
        if (accumulate) {
            r = g4[i];
            r.x = fmaf(sc, v.x, r.x); r.y = fmaf(sc, v.y, r.y); r.z = fmaf(sc, v.z, r.z); r.w = fmaf(sc, v.w, r.w);
        } else {
            r = make_float4(sc * v.x, sc * v.y, sc * v.z, sc * v.w);
        }
        g4[i] = r;
    }
    for (long i = 4 * n4 + tid; i < n; i += nthr) g[i] = accumulate ? fmaf(sc, w[i], g[i]) : sc * w[i];
}

extern "C" {

int xdfm_l2_reg_fwd(const float* const* ptrs, const long* numel, const float* coeff, int T, float* partials,
                    float* out, void* stream) {
    XDFM_REQUIRE(ptrs && numel && coeff && partials && out, "l2_reg_fwd: null pointer");
    XDFM_REQUIRE(T > 0 && T <= 65535, "l2_reg_fwd: T=%d", T);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(l2_sumsq_kernel, dim3(REG_BLOCKS, T), dim3(REG_THREADS), 0, st, ptrs, numel, partials);
    int rc = xdfm_check_launch("l2_reg_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(l2_finish_kernel, dim3(1), dim3(REG_THREADS), 0, st, partials, coeff, T, out);
    return xdfm_check_launch("l2_reg_fwd finish");
}

int xdfm_l2_reg_bwd(const float* const* ptrs, const long* numel, const float* coeff, int T, const float* gscale,
                    float* gflat, const long* goff, int accumulate, void* stream) {
    XDFM_REQUIRE(ptrs && numel && coeff && gscale && gflat && goff, "l2_reg_bwd: null pointer");
    XDFM_REQUIRE(T > 0 && T <= 65535, "l2_reg_bwd: T=%d", T);
    hipLaunchKernelGGL(l2_grad_kernel, dim3(4 * REG_BLOCKS, T), dim3(REG_THREADS), 0, (hipStream_t)stream, ptrs,
                       numel, coeff, gscale, gflat, goff, accumulate);
    return xdfm_check_launch("l2_reg_bwd");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Column sums of a row-major [rows][cols] matrix: the bias gradient of a dense layer
// (autograd of deepctr/layers/core.py:120-134: grad_bias = grad_output.sum(0)).  ATen's multi-block
// reduction zeroes a semaphore buffer with hipMemsetAsync first; a memset node inside a captured HIP
// graph is not reliably ordered against its neighbouring kernel nodes on this stack
// (tools/graph_memset_probe.py: 3 of 4 replays wrong), so the train step uses this two-launch,
// atomics-free, fixed-order version instead.
#define CS_ROWBLK 64
// the CS_ROWBLK partials of every column, summed in the order of colsum_finish_kernel (4 groups of 16, then the groups),
// by the block that finished last (xdfm_last_block_done): same bits, no launch of its own
__device__ __forceinline__ void colsum_finish_all(const float* __restrict__ part, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * 64 + threadIdx.x;                      // the 64 columns of this column block
    if (threadIdx.x < 64 && c < cols) {
        float s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s[q] = 0.f;
#pragma unroll
            for (int k = 0; k < CS_ROWBLK / 4; ++k) s[q] += xdfm_peer(part + (long)(q * (CS_ROWBLK / 4) + k) * cols + c);
        }
        out[c] = (s[0] + s[1]) + (s[2] + s[3]);
    }
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ g, long rows, int cols, long ld,
                                                            float* __restrict__ part, float* __restrict__ out,
                                                            unsigned* __restrict__ ticket) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;                         // 4 row groups per block
    const long per = (rows + CS_ROWBLK - 1) / CS_ROWBLK;
    const long r0 = (long)blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    float a = 0.f;
    if (c < cols)
        for (long r = r0 + rg; r < r1; r += 16) {            // 4 rows per step, loads issued together
            float gv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) gv[t] = g[(r + 4 * t < r1 ? r + 4 * t : r) * ld + c];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (r + 4 * t < r1) a += gv[t];
        }
    __shared__ float red[4][64];
    red[rg][threadIdx.x & 63] = a;
    __syncthreads();
    if (rg == 0 && c < cols)
        xdfm_publish(&part[(long)blockIdx.y * cols + c], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
    if (ticket && xdfm_last_block_done(ticket + blockIdx.x, gridDim.y)) colsum_finish_all(part, cols, out);      // one ticket per column block
}

// same walk with the ReLU mask applied on the way: gz = (y > 0) ? g : 0 is written out (the operand of the two
// backward GEMMs of a dense layer) and summed (its bias gradient) -- threshold_backward + column sum in one pass
__global__ __launch_bounds__(256) void relu_bwd_colsum_partial_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                                     long rows, int cols, long ldg, long ldy,
                                                                     float* __restrict__ gz, float* __restrict__ part,
                                                                     float* __restrict__ out, unsigned* __restrict__ ticket) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    const long per = (rows + CS_ROWBLK - 1) / CS_ROWBLK;
    const long r0 = (long)blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    float a = 0.f;
    if (c < cols) {
        // 4 rows per step, both operands loaded unconditionally (a load behind the y > 0 test would wait for it)
        for (long r = r0 + rg; r < r1; r += 16) {
            float gv[4], yv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const long rr = r + 4 * t < r1 ? r + 4 * t : r;
                gv[t] = g[rr * ldg + c];
                yv[t] = y[rr * ldy + c];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (r + 4 * t >= r1) continue;
                const float v = yv[t] > 0.f ? gv[t] : 0.f;
                gz[(r + 4 * t) * cols + c] = v;
                a += v;
            }
        }
    }
    __shared__ float red[4][64];
    red[rg][threadIdx.x & 63] = a;
    __syncthreads();
    if (rg == 0 && c < cols)
        xdfm_publish(&part[(long)blockIdx.y * cols + c], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
    if (ticket && xdfm_last_block_done(ticket + blockIdx.x, gridDim.y)) colsum_finish_all(part, cols, out);      // one ticket per column block
}

// 64 columns per block, the CS_ROWBLK partials of a column summed by 4 threads in a fixed order
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int q = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols) {
#pragma unroll
        for (int k = 0; k < CS_ROWBLK / 4; ++k) s += part[(long)(q * (CS_ROWBLK / 4) + k) * cols + c];
    }
    __shared__ float red[4][64];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < cols) out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" {

size_t xdfm_colsum_ws_elems(int cols) { return cols > 0 ? (size_t)CS_ROWBLK * cols : 0; }

int xdfm_colsum(const float* g, long rows, int cols, long ld, float* ws, float* out, void* stream) {
    XDFM_REQUIRE(g && ws && out, "colsum: null pointer");
    XDFM_REQUIRE(rows > 0 && cols > 0 && ld >= cols, "colsum: bad shape rows=%ld cols=%d ld=%ld", rows, cols, ld);
    hipStream_t st = (hipStream_t)stream;
    unsigned* ticket = ceil_div(cols, 64) <= TK_ROWS ? xdfm_ticket(TK_ROW0) : nullptr;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(cols, 64), CS_ROWBLK), dim3(256), 0, st, g, rows, cols, ld, ws, out, ticket);
    if (!ticket) hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0, st, ws, cols, out);
    return xdfm_check_launch("colsum");
}

int xdfm_relu_bwd_colsum(const float* g, const float* y, long rows, int cols, long ldg, long ldy, float* ws, float* gz,
                         float* out, void* stream) {
    XDFM_REQUIRE(g && y && ws && gz && out, "relu_bwd_colsum: null pointer");
    XDFM_REQUIRE(rows > 0 && cols > 0 && ldg >= cols && ldy >= cols, "relu_bwd_colsum: bad shape rows=%ld cols=%d", rows, cols);
    hipStream_t st = (hipStream_t)stream;
    unsigned* ticket = ceil_div(cols, 64) <= TK_ROWS ? xdfm_ticket(TK_ROW0) : nullptr;
    hipLaunchKernelGGL(relu_bwd_colsum_partial_kernel, dim3(ceil_div(cols, 64), CS_ROWBLK), dim3(256), 0, st, g, y, rows, cols,
                       ldg, ldy, gz, ws, out, ticket);
    if (!ticket) hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0, st, ws, cols, out);
    return xdfm_check_launch("relu_bwd_colsum");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Output head of the binary task:  z_b = lin_b + <u_b, wu> + <v_b, wv> + bias,  p = sigmoid(z),  loss = sum_b BCE(p_b, y_b)
// replaces cin_linear / dnn_linear (two [B,K]x[K,1] products, deepctr/models/xdeepfm.py:95-105), the logit sum,
// PredictionLayer (deepctr/layers/core.py:150-160) and F.binary_cross_entropy(reduction='sum') (basemodel.py:254)
// with their autograd -- two GEMV + ~10 elementwise / reduction launches forward and four skinny GEMMs + ~6
// launches backward, each a few microseconds of launch floor -- by two launches each way.
// BCE as ATen evaluates it: log terms clamped at -100; backward (p - y) * p(1-p) / max(p(1-p), 1e-12).
// All sums in a fixed order (per-block partials + a finish kernel): deterministic.
#define HEAD_BLOCKS 128
#define HEAD_THREADS 256

__device__ __forceinline__ float head_row_dot(const float* __restrict__ x, const float* __restrict__ w, int K, int lane) {
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(x[k], w[k], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// 16 lanes per row, 4 rows per wave at a time, float4 loads issued back to back (K % 4 == 0, 16-byte aligned rows)
template <int MAXQ>
__device__ __forceinline__ float head_row_dot_vec(const float* __restrict__ x, const float* __restrict__ w, int K4, int sub) {
    float4 xv[MAXQ], wv[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int k = sub + 16 * q;
        const int kc = k < K4 ? k : 0;
        xv[q] = reinterpret_cast<const float4*>(x)[kc];
        wv[q] = reinterpret_cast<const float4*>(w)[kc];
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q)
        if (sub + 16 * q < K4) s += (xv[q].x * wv[q].x + xv[q].y * wv[q].y) + (xv[q].z * wv[q].z + xv[q].w * wv[q].w);
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// rows strided over all 16-lane groups of the grid; per-block partial loss -> part[blockIdx.x]
template <bool VEC>
__global__ __launch_bounds__(HEAD_THREADS) void head_fwd_kernel(
    const float* __restrict__ lin, const float* __restrict__ u, const float* __restrict__ wu, int Ku,
    const float* __restrict__ v, const float* __restrict__ wv, int Kv, const float* __restrict__ bias,
    const float* __restrict__ y, int B, float* __restrict__ pred, float* __restrict__ part, float* __restrict__ loss,
    unsigned* __restrict__ ticket) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float bv = bias ? bias[0] : 0.f;
    float acc = 0.f;
    if constexpr (VEC) {
        const int sub = lane & 15, grp = lane >> 4;
        for (int b = (blockIdx.x * 4 + wave) * 4 + grp; b < B; b += HEAD_BLOCKS * 16) {
            float z = (lin ? lin[b] : 0.f) + bv;
            if (u) z += head_row_dot_vec<8>(u + (long)b * Ku, wu, Ku >> 2, sub);       // K <= 512
            if (v) z += head_row_dot_vec<8>(v + (long)b * Kv, wv, Kv >> 2, sub);
            const float p = 1.f / (1.f + expf(-z));
            const float t = y[b];
            if (sub == 0) {
                pred[b] = p;
                acc += -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f));
            }
        }
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
    } else {
        for (int b = blockIdx.x * 4 + wave; b < B; b += HEAD_BLOCKS * 4) {
            float z = (lin ? lin[b] : 0.f) + bv;
            if (u) z += head_row_dot(u + (long)b * Ku, wu, Ku, lane);
            if (v) z += head_row_dot(v + (long)b * Kv, wv, Kv, lane);
            const float p = 1.f / (1.f + expf(-z));
            const float t = y[b];
            if (lane == 0) {
                pred[b] = p;
                acc += -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f));
            }
        }
    }
    __shared__ float red[4];
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) xdfm_publish(&part[blockIdx.x], (red[0] + red[1]) + (red[2] + red[3]));
    if (ticket && xdfm_last_block_done(ticket, gridDim.x)) {         // the tree of head_fwd_finish_kernel, by the last block
        __shared__ float tree[HEAD_BLOCKS];
        if (threadIdx.x < HEAD_BLOCKS) tree[threadIdx.x] = xdfm_peer(part + threadIdx.x);
        __syncthreads();
        for (int o = HEAD_BLOCKS / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) tree[threadIdx.x] += tree[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss[0] = tree[0];
    }
}

__global__ __launch_bounds__(HEAD_BLOCKS) void head_fwd_finish_kernel(const float* __restrict__ part, float* __restrict__ loss) {
    __shared__ float red[HEAD_BLOCKS];
    red[threadIdx.x] = part[threadIdx.x];
    __syncthreads();
    for (int o = HEAD_BLOCKS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0];
}

// g_b = gloss * dBCE/dz;  dlin_b = g_b;  du[b][k] = g_b wu[k];  dv likewise;
// per-block partials of dwu[k] = sum_b g_b u[b][k], dwv[k], dbias = sum_b g_b  -> part[blk][Ku + Kv + 1]
__global__ __launch_bounds__(HEAD_THREADS) void head_bwd_kernel(
    const float* __restrict__ pred, const float* __restrict__ y, const float* __restrict__ gloss,
    const float* __restrict__ u, const float* __restrict__ wu, int Ku, const float* __restrict__ v,
    const float* __restrict__ wv, int Kv, int B, float* __restrict__ dlin, float* __restrict__ du,
    float* __restrict__ dv, float* __restrict__ part, float* __restrict__ grads, unsigned* __restrict__ ticket) {
    extern __shared__ float sm[];                 // [4 waves][Ku + Kv + 1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KT = Ku + Kv + 1;
    float* mine = sm + wave * KT;
    for (int k = lane; k < KT; k += 64) mine[k] = 0.f;
    const float gl = gloss[0];
    float gsum = 0.f;
    for (int b = blockIdx.x * 4 + wave; b < B; b += HEAD_BLOCKS * 4) {
        const float p = pred[b], t = y[b];
        const float pq = (1.f - p) * p;
        const float g = gl * (p - t) / fmaxf(pq, 1e-12f) * pq;
        if (lane == 0 && dlin) dlin[b] = g;
        gsum += g;
        if (u) {
            const float* ur = u + (long)b * Ku;
            float* dr = du + (long)b * Ku;
            for (int k = lane; k < Ku; k += 64) { dr[k] = g * wu[k]; mine[k] = fmaf(g, ur[k], mine[k]); }
        }
        if (v) {
            const float* vr = v + (long)b * Kv;
            float* dr = dv + (long)b * Kv;
            for (int k = lane; k < Kv; k += 64) { dr[k] = g * wv[k]; mine[Ku + k] = fmaf(g, vr[k], mine[Ku + k]); }
        }
    }
    if (lane == 0) mine[Ku + Kv] = gsum;
    __syncthreads();
    for (int k = threadIdx.x; k < KT; k += HEAD_THREADS)
        xdfm_publish(&part[(long)blockIdx.x * KT + k], (sm[k] + sm[KT + k]) + (sm[2 * KT + k] + sm[3 * KT + k]));
    if (ticket && xdfm_last_block_done(ticket, gridDim.x))          // head_bwd_finish_kernel's sums, by the last block
        for (int k = threadIdx.x; k < KT; k += HEAD_THREADS) {
            float s = 0.f;
            for (int b = 0; b < HEAD_BLOCKS; ++b) s += xdfm_peer(part + (long)b * KT + k);
            grads[k] = s;
        }
}

// vectorised variant (K % 4 == 0, K <= 512, 16-byte aligned): 16 lanes per row, 4 rows per wave at a time, float4
// loads / stores issued back to back; every lane keeps the partial column sums of its own columns in registers
// (reduced over the wave's 4 row groups by shuffles at the end, over the 4 waves through LDS) -- the scalar
// kernel above walks its rows one after the other with an LDS read-modify-write per element
__global__ __launch_bounds__(HEAD_THREADS) void head_bwd_vec_kernel(
    const float* __restrict__ pred, const float* __restrict__ y, const float* __restrict__ gloss,
    const float* __restrict__ u, const float* __restrict__ wu, int Ku, const float* __restrict__ v,
    const float* __restrict__ wv, int Kv, int B, float* __restrict__ dlin, float* __restrict__ du,
    float* __restrict__ dv, float* __restrict__ part, float* __restrict__ grads, unsigned* __restrict__ ticket) {
    extern __shared__ float sm[];                 // [4 waves][Ku + Kv + 1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, grp = lane >> 4;
    const int KT = Ku + Kv + 1, Ku4 = Ku >> 2, Kv4 = Kv >> 2;
    const float gl = gloss[0];
    float4 wu4[8], wv4[8], au[8], av[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = sub + 16 * q;
        wu4[q] = (u && k < Ku4) ? reinterpret_cast<const float4*>(wu)[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        wv4[q] = (v && k < Kv4) ? reinterpret_cast<const float4*>(wv)[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        au[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float gsum = 0.f;
    for (int b = (blockIdx.x * 4 + wave) * 4 + grp; b < B; b += HEAD_BLOCKS * 16) {
        const float p = pred[b], t = y[b];
        const float pq = (1.f - p) * p;
        const float g = gl * (p - t) / fmaxf(pq, 1e-12f) * pq;
        if (sub == 0) { if (dlin) dlin[b] = g; gsum += g; }
        float4 xu[8], xv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = sub + 16 * q;
            xu[q] = reinterpret_cast<const float4*>(u ? u + (long)b * Ku : pred)[(u && k < Ku4) ? k : 0];
            xv[q] = reinterpret_cast<const float4*>(v ? v + (long)b * Kv : pred)[(v && k < Kv4) ? k : 0];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = sub + 16 * q;
            if (u && k < Ku4) {
                reinterpret_cast<float4*>(du + (long)b * Ku)[k] = make_float4(g * wu4[q].x, g * wu4[q].y, g * wu4[q].z, g * wu4[q].w);
                au[q].x = fmaf(g, xu[q].x, au[q].x); au[q].y = fmaf(g, xu[q].y, au[q].y);
                au[q].z = fmaf(g, xu[q].z, au[q].z); au[q].w = fmaf(g, xu[q].w, au[q].w);
            }
            if (v && k < Kv4) {
                reinterpret_cast<float4*>(dv + (long)b * Kv)[k] = make_float4(g * wv4[q].x, g * wv4[q].y, g * wv4[q].z, g * wv4[q].w);
                av[q].x = fmaf(g, xv[q].x, av[q].x); av[q].y = fmaf(g, xv[q].y, av[q].y);
                av[q].z = fmaf(g, xv[q].z, av[q].z); av[q].w = fmaf(g, xv[q].w, av[q].w);
            }
        }
    }
    // sum over the wave's 4 row groups (lanes sub, sub+16, sub+32, sub+48 hold the same columns)
    float* mine = sm + wave * KT;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float c[8] = {au[q].x, au[q].y, au[q].z, au[q].w, av[q].x, av[q].y, av[q].z, av[q].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) { c[e] += __shfl_xor(c[e], 16); c[e] += __shfl_xor(c[e], 32); }
        const int k = sub + 16 * q;
        if (grp == 0) {
            if (u && k < Ku4) { mine[4 * k] = c[0]; mine[4 * k + 1] = c[1]; mine[4 * k + 2] = c[2]; mine[4 * k + 3] = c[3]; }
            if (v && k < Kv4) { mine[Ku + 4 * k] = c[4]; mine[Ku + 4 * k + 1] = c[5]; mine[Ku + 4 * k + 2] = c[6]; mine[Ku + 4 * k + 3] = c[7]; }
        }
    }
    gsum += __shfl_xor(gsum, 16);
    gsum += __shfl_xor(gsum, 32);
    if (lane == 0) mine[Ku + Kv] = gsum;
    __syncthreads();
    for (int k = threadIdx.x; k < KT; k += HEAD_THREADS)
        xdfm_publish(&part[(long)blockIdx.x * KT + k], (sm[k] + sm[KT + k]) + (sm[2 * KT + k] + sm[3 * KT + k]));
    if (ticket && xdfm_last_block_done(ticket, gridDim.x))          // head_bwd_finish_kernel's sums, by the last block
        for (int k = threadIdx.x; k < KT; k += HEAD_THREADS) {
            float s = 0.f;
            for (int b = 0; b < HEAD_BLOCKS; ++b) s += xdfm_peer(part + (long)b * KT + k);
            grads[k] = s;
        }
}

__global__ void head_bwd_finish_kernel(const float* __restrict__ part, int KT, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= KT) return;
    float s = 0.f;
    for (int b = 0; b < HEAD_BLOCKS; ++b) s += part[(long)b * KT + k];
    out[k] = s;
}

extern "C" {

size_t xdfm_head_ws_elems(int Ku, int Kv) { return (size_t)HEAD_BLOCKS * (size_t)(Ku + Kv + 1) + HEAD_BLOCKS; }

int xdfm_head_fwd(const float* lin, const float* u, const float* wu, int Ku, const float* v, const float* wv, int Kv,
                  const float* bias, const float* y, int B, float* pred, float* loss, float* ws, void* stream) {
    XDFM_REQUIRE(y && pred && loss && ws && B > 0, "head_fwd: bad arguments");
    XDFM_REQUIRE((!u || (wu && Ku > 0)) && (!v || (wv && Kv > 0)) && Ku >= 0 && Kv >= 0, "head_fwd: bad operand shapes");
    hipStream_t st = (hipStream_t)stream;
    const int ku = u ? Ku : 0, kv = v ? Kv : 0;
    const bool vec = ku % 4 == 0 && kv % 4 == 0 && ku <= 512 && kv <= 512 &&
                     ((((size_t)u) | ((size_t)v) | ((size_t)wu) | ((size_t)wv)) & 15) == 0;
    unsigned* ticket = xdfm_ticket(TK_HEAD_FWD);
    if (vec)
        hipLaunchKernelGGL(head_fwd_kernel<true>, dim3(HEAD_BLOCKS), dim3(HEAD_THREADS), 0, st, lin, u, wu, ku, v, wv, kv,
                           bias, y, B, pred, ws, loss, ticket);
    else
        hipLaunchKernelGGL(head_fwd_kernel<false>, dim3(HEAD_BLOCKS), dim3(HEAD_THREADS), 0, st, lin, u, wu, ku, v, wv, kv,
                           bias, y, B, pred, ws, loss, ticket);
    if (!ticket) hipLaunchKernelGGL(head_fwd_finish_kernel, dim3(1), dim3(HEAD_BLOCKS), 0, st, ws, loss);
    return xdfm_check_launch("head_fwd");
}

/* grads: [Ku + Kv + 1] = dwu | dwv | dbias */
int xdfm_head_bwd(const float* pred, const float* y, const float* gloss, const float* u, const float* wu, int Ku,
                  const float* v, const float* wv, int Kv, int B, float* dlin, float* du, float* dv, float* grads,
                  float* ws, void* stream) {
    XDFM_REQUIRE(pred && y && gloss && grads && ws && B > 0, "head_bwd: bad arguments");
    XDFM_REQUIRE((!u || (wu && du && Ku > 0)) && (!v || (wv && dv && Kv > 0)), "head_bwd: bad operand shapes");
    const int ku = u ? Ku : 0, kv = v ? Kv : 0, KT = ku + kv + 1;
    const size_t lds = (size_t)4 * KT * sizeof(float);
    XDFM_REQUIRE(lds <= 64 * 1024, "head_bwd: Ku + Kv = %d too large", ku + kv);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = ku % 4 == 0 && kv % 4 == 0 && ku <= 512 && kv <= 512 &&
                     ((((size_t)u) | ((size_t)v) | ((size_t)wu) | ((size_t)wv) | ((size_t)du) | ((size_t)dv)) & 15) == 0;
    unsigned* ticket = xdfm_ticket(TK_HEAD_BWD);
    if (vec)
        hipLaunchKernelGGL(head_bwd_vec_kernel, dim3(HEAD_BLOCKS), dim3(HEAD_THREADS), lds, st, pred, y, gloss, u, wu, ku, v,
                           wv, kv, B, dlin, du, dv, ws, grads, ticket);
    else
        hipLaunchKernelGGL(head_bwd_kernel, dim3(HEAD_BLOCKS), dim3(HEAD_THREADS), lds, st, pred, y, gloss, u, wu, ku, v, wv,
                           kv, B, dlin, du, dv, ws, grads, ticket);
    if (!ticket) hipLaunchKernelGGL(head_bwd_finish_kernel, dim3(ceil_div(KT, 256)), dim3(256), 0, st, ws, KT, grads);
    return xdfm_check_launch("head_bwd");
}

}  // extern "C"
