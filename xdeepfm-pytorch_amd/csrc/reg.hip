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
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ g, long rows, int cols, long ld,
                                                            float* __restrict__ part) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;                         // 4 row groups per block
    const long per = (rows + CS_ROWBLK - 1) / CS_ROWBLK;
    const long r0 = (long)blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    float a = 0.f;
    if (c < cols)
        for (long r = r0 + rg; r < r1; r += 4) a += g[r * ld + c];
    __shared__ float red[4][64];
    red[rg][threadIdx.x & 63] = a;
    __syncthreads();
    if (rg == 0 && c < cols)
        part[(long)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
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
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(cols, 64), CS_ROWBLK), dim3(256), 0, st, g, rows, cols, ld, ws);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0, st, ws, cols, out);
    return xdfm_check_launch("colsum");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Output head of the binary task: p = sigmoid(a + b + c + bias), loss = sum_b BCE(p_b, y_b)
// (deepctr/models/xdeepfm.py:100-107 logit sum, deepctr/layers/core.py:150-160 PredictionLayer,
// basemodel.py:254 F.binary_cross_entropy(reduction='sum')) in one single-block launch each way instead of
// ~12 elementwise / reduction launches of a few microseconds each.  BCE as ATen evaluates it: log terms
// clamped at -100; backward (p - y) * p(1-p) / max(p(1-p), 1e-12).  Fixed summation order.
__global__ __launch_bounds__(1024) void head_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ c, const float* __restrict__ bias,
                                                       const float* __restrict__ y, int B, float* __restrict__ pred,
                                                       float* __restrict__ loss) {
    const float bv = bias ? bias[0] : 0.f;
    float part = 0.f;
    for (int i = threadIdx.x; i < B; i += 1024) {
        float z = a[i];
        if (b) z += b[i];
        if (c) z += c[i];
        z += bv;
        const float p = 1.f / (1.f + expf(-z));
        pred[i] = p;
        const float t = y[i];
        part += -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f));
    }
    __shared__ float red[1024];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0];
}

__global__ __launch_bounds__(1024) void head_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                       const float* __restrict__ gloss, int B, float* __restrict__ dlogit,
                                                       float* __restrict__ dbias) {
    const float gl = gloss[0];
    float part = 0.f;
    for (int i = threadIdx.x; i < B; i += 1024) {
        const float p = pred[i], t = y[i];
        const float pq = (1.f - p) * p;
        const float g = gl * (p - t) / fmaxf(pq, 1e-12f) * pq;
        dlogit[i] = g;
        part += g;
    }
    __shared__ float red[1024];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && dbias) dbias[0] = red[0];
}

extern "C" {

int xdfm_head_fwd(const float* a, const float* b, const float* c, const float* bias, const float* y, int B, float* pred,
                  float* loss, void* stream) {
    XDFM_REQUIRE(a && y && pred && loss && B > 0, "head_fwd: bad arguments");
    hipLaunchKernelGGL(head_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, c, bias, y, B, pred, loss);
    return xdfm_check_launch("head_fwd");
}

int xdfm_head_bwd(const float* pred, const float* y, const float* gloss, int B, float* dlogit, float* dbias, void* stream) {
    XDFM_REQUIRE(pred && y && gloss && dlogit && B > 0, "head_bwd: bad arguments");
    hipLaunchKernelGGL(head_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, y, gloss, B, dlogit, dbias);
    return xdfm_check_launch("head_bwd");
}

}  // extern "C"
