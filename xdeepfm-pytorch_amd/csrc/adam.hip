// K7: Adam step over all parameters (one launch per 40 tensors; the embedding tables are 98 % of the bytes).
//
// replaces torch.optim.Adam.step() for the [vocab, D] / [vocab, 1] tables (deepctr/models/basemodel.py:452
// builds torch.optim.Adam over every parameter; the reference's tables carry dense gradients,
// deepctr/inputs.py:168 sparse=False, so every row is updated every step).  44 M parameters at config 2 =
// 1.2 GB of traffic per step (p, m, v read + written, g read): pure HBM streaming, 16-byte accesses, 8 loads
// in flight per thread.  Arithmetic follows ATen's fused kernel (fused_adam_utils.cuh), in fp32:
//   m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Tensor descriptors travel BY VALUE in the kernel arguments (no device-side pointer tables to build or
// refresh, nothing to upload when a gradient moves); the step counters are the device-resident fp32 scalars
// torch keeps per parameter (already incremented by the caller).
//
// Optional L2 term (deepctr/models/basemodel.py:412-428, l2 * sum(w^2) added to the loss): with l2[t] given, the
// kernel adds its gradient 2 l2[t] w to g on the fly (w is being read anyway) and returns the term's VALUE of
// the pre-update weights (per-block partials summed in a fixed order by adam_l2_finish_kernel).  That removes
// two table-sized passes per step (sum of squares; gradient initialisation) from the train step.
#include "xdfm_internal.h"

#include <algorithm>
#include <vector>

#define ADAM_THREADS 256
#define ADAM_BX 128

struct AdamCoef { float w1, b2, w2, lr, eps; };

// The fusions are spelled out and the compiler's own contraction is off: left to itself it fuses differently in
// the marked and the dense loop below, and the two must give the same bits.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float bc2_sqrt, const AdamCoef& c) {
#pragma clang fp contract(off)
    m = fmaf(c.w1, g - m, m);
    v = fmaf(c.w2 * g, g, c.b2 * v);
    const float denom = sqrtf(v) / bc2_sqrt + c.eps;
    p -= step_size * m / denom;
}

#define ADAM_CHUNK 40
struct AdamBatch { xdfm_adam_tensor t[ADAM_CHUNK]; };

__global__ __launch_bounds__(ADAM_THREADS) void adam_step_kernel(
    const AdamBatch batch, int t0, double lr_arg, const double* __restrict__ lr_dev, double beta1, double beta2, double eps,
    float* __restrict__ l2_part) {
    // the learning rate as a kernel argument, or read from device memory (a captured HIP graph then follows a
    // learning-rate schedule without being captured again)
    const double lr = lr_dev ? *lr_dev : lr_arg;
    // gridDim.x blocks per tensor (ADAM_BX by default; fewer = a small footprint that can share the chip with an
    // MFMA-bound kernel on another stream); the L2 partials keep their ADAM_BX slots per tensor
    const xdfm_adam_tensor& d = batch.t[blockIdx.y];
    const int t = t0 + blockIdx.y;
    float* __restrict__ p = d.param;
    float* __restrict__ m = d.exp_avg;
    float* __restrict__ v = d.exp_avg_sq;
    float* __restrict__ g = d.grad;
    unsigned char* __restrict__ marks = d.grad_marks;
    const long n = d.numel;
    const float* l2 = &d.l2;
    // bias corrections in double as ATen does, once per block (two double pow() per THREAD cost more than
    // the whole update of a small tensor)
    __shared__ float bcs[2];
    if (threadIdx.x == 0) {
        const double step = (double)*d.step;
        bcs[0] = (float)(lr / (1.0 - pow(beta1, step)));
        bcs[1] = (float)sqrt(1.0 - pow(beta2, step));
    }
    __syncthreads();
    const float step_size = bcs[0], bc2_sqrt = bcs[1];
    const AdamCoef c = {(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)lr, (float)eps};
    const float l2c = *l2;
    const float g2 = 2.f * l2c;                        // d(l2 * w^2)/dw = 2 l2 w
    float sq = 0.f;
    const long tid = (long)blockIdx.x * ADAM_THREADS + threadIdx.x;
    const long stride = (long)gridDim.x * ADAM_THREADS;
    const bool vec = ((((size_t)p) | ((size_t)m) | ((size_t)v) | ((size_t)g)) & 15) == 0;
    const long n4 = vec ? n / 4 : 0;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    float4* g4 = reinterpret_cast<float4*>(g);
    long i = tid;
    if (marks && (d.flags & XDFM_ADAM_LAZY)) {
        // opt-in row-sparse update: chunks the batch did not touch are skipped altogether (see xdfm.h)
        float zf;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zf));
        const float4 zero4 = make_float4(zf, zf, zf, zf);
        for (; i + 3 * stride < n4; i += 4 * stride) {            // 4 mark bytes in flight per thread
            unsigned char k[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) k[q] = marks[i + q * stride];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (k[q]) {
                    const long e = i + q * stride;
                    float4 pa = p4[e], ma = m4[e], va = v4[e], ga = g4[e];
                    g4[e] = zero4; marks[e] = 0;
                    sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
                    ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
                    adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
                    adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
                    p4[e] = pa; m4[e] = ma; v4[e] = va;
                }
            }
        }
        for (; i < n4; i += stride) {
            if (marks[i]) {
                float4 pa = p4[i], ma = m4[i], va = v4[i], ga = g4[i];
                g4[i] = zero4; marks[i] = 0;
                sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
                ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
                adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
                adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
                p4[i] = pa; m4[i] = ma; v4[i] = va;
            }
        }
    } else if (marks) {
        // sparse gradient: a chunk's mark says whether the scatter touched it; unmarked chunks are zeros by
        // construction and are not read, marked ones are read, re-zeroed and unmarked (each lane owns its chunk's
        // mark, so nothing races).  The mark bytes are loaded first, the gradient loads sit behind them.
        // The zero is opaque to the compiler: with a literal 0 it folds fmaf(2*l2, w, 0) into a product and then
        // contracts that product into the next subtraction, which rounds differently from the dense path -- with
        // the same instruction sequence both paths give the same bits.
        float zf;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zf));
        const float4 zero4 = make_float4(zf, zf, zf, zf);
        for (; i + stride < n4; i += 2 * stride) {
            const unsigned char ka = marks[i], kb = marks[i + stride];
            float4 pa = p4[i], ma = m4[i], va = v4[i];
            float4 pb = p4[i + stride], mb = m4[i + stride], vb = v4[i + stride];
            float4 ga = zero4, gb = zero4;
            if (ka) { ga = g4[i]; g4[i] = zero4; marks[i] = 0; }
            if (kb) { gb = g4[i + stride]; g4[i + stride] = zero4; marks[i + stride] = 0; }
            sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w) + (pb.x * pb.x + pb.y * pb.y) + (pb.z * pb.z + pb.w * pb.w);
            ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
            gb.x = fmaf(g2, pb.x, gb.x); gb.y = fmaf(g2, pb.y, gb.y); gb.z = fmaf(g2, pb.z, gb.z); gb.w = fmaf(g2, pb.w, gb.w);
            adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
            adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
            adam_one(pb.x, gb.x, mb.x, vb.x, step_size, bc2_sqrt, c); adam_one(pb.y, gb.y, mb.y, vb.y, step_size, bc2_sqrt, c);
            adam_one(pb.z, gb.z, mb.z, vb.z, step_size, bc2_sqrt, c); adam_one(pb.w, gb.w, mb.w, vb.w, step_size, bc2_sqrt, c);
            p4[i] = pa; m4[i] = ma; v4[i] = va;
            p4[i + stride] = pb; m4[i + stride] = mb; v4[i + stride] = vb;
        }
        for (; i < n4; i += stride) {
            const unsigned char ka = marks[i];
            float4 pa = p4[i], ma = m4[i], va = v4[i];
            float4 ga = zero4;
            if (ka) { ga = g4[i]; g4[i] = zero4; marks[i] = 0; }
            sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
            ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
            adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
            adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
            p4[i] = pa; m4[i] = ma; v4[i] = va;
        }
    }
    for (; i + stride < n4; i += 2 * stride) {            // two float4 per array in flight
        float4 pa = p4[i], ga = g4[i], ma = m4[i], va = v4[i];
        float4 pb = p4[i + stride], gb = g4[i + stride], mb = m4[i + stride], vb = v4[i + stride];
        sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w) + (pb.x * pb.x + pb.y * pb.y) + (pb.z * pb.z + pb.w * pb.w);
        ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
        gb.x = fmaf(g2, pb.x, gb.x); gb.y = fmaf(g2, pb.y, gb.y); gb.z = fmaf(g2, pb.z, gb.z); gb.w = fmaf(g2, pb.w, gb.w);
        adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
        adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
        adam_one(pb.x, gb.x, mb.x, vb.x, step_size, bc2_sqrt, c); adam_one(pb.y, gb.y, mb.y, vb.y, step_size, bc2_sqrt, c);
        adam_one(pb.z, gb.z, mb.z, vb.z, step_size, bc2_sqrt, c); adam_one(pb.w, gb.w, mb.w, vb.w, step_size, bc2_sqrt, c);
        p4[i] = pa; m4[i] = ma; v4[i] = va;
        p4[i + stride] = pb; m4[i + stride] = mb; v4[i + stride] = vb;
    }
    for (; i < n4; i += stride) {
        float4 pa = p4[i], ga = g4[i], ma = m4[i], va = v4[i];
        sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
        ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
        adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
        adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
        p4[i] = pa; m4[i] = ma; v4[i] = va;
    }
    for (long k = 4 * n4 + tid; k < n; k += stride) {
        float pa = p[k], ma = m[k], va = v[k];
        sq = fmaf(pa, pa, sq);
        adam_one(pa, fmaf(g2, pa, g[k]), ma, va, step_size, bc2_sqrt, c);
        p[k] = pa; m[k] = ma; v[k] = va;
        if (marks) { g[k] = 0.f; marks[k >> 2] = 0; }
    }
    if (l2_part) {                                     // fixed-order block reduction of the squares
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        __shared__ float wsum[ADAM_THREADS / 64];
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) l2_part[(long)t * ADAM_BX + blockIdx.x] = l2c * ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
        if (threadIdx.x < ADAM_BX - gridDim.x && blockIdx.x == 0) l2_part[(long)t * ADAM_BX + gridDim.x + threadIdx.x] = 0.f;
    }
}

__global__ __launch_bounds__(1024) void adam_l2_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float acc[1024];
    float v = 0.f;
    for (int k = threadIdx.x; k < n; k += 1024) v += part[k];
    acc[threadIdx.x] = v;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) acc[threadIdx.x] += acc[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = acc[0];
}

extern "C" {

size_t xdfm_adam_step_ws_elems(int T) { return T > 0 ? (size_t)T * ADAM_BX : 0; }

int xdfm_adam_step(const xdfm_adam_tensor* tensors, int T, double lr, double beta1, double beta2, double eps,
                   float* l2_ws, float* l2_value, void* stream) {
    return xdfm_adam_step_lr(tensors, T, lr, nullptr, beta1, beta2, eps, l2_ws, l2_value, stream);
}

int xdfm_adam_step_lr(const xdfm_adam_tensor* tensors, int T, double lr, const double* lr_dev, double beta1, double beta2,
                      double eps, float* l2_ws, float* l2_value, void* stream) {
    XDFM_REQUIRE(tensors, "adam_step: null pointer");
    XDFM_REQUIRE(T > 0 && T <= 65535, "adam_step: bad tensor count %d", T);
    XDFM_REQUIRE(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, "adam_step: bad hyper-parameters");
    XDFM_REQUIRE(!l2_value || l2_ws, "adam_step: l2_value needs l2_ws");
    for (int t = 0; t < T; ++t)
        XDFM_REQUIRE(tensors[t].param && tensors[t].grad && tensors[t].exp_avg && tensors[t].exp_avg_sq && tensors[t].step &&
                         tensors[t].numel >= 0, "adam_step: tensor %d has a null pointer", t);
    for (int t = 0; t < T; ++t)
        XDFM_REQUIRE(!(tensors[t].flags & XDFM_ADAM_LAZY) || tensors[t].grad_marks, "adam_step: tensor %d is lazy but has no grad_marks", t);
    for (int t = 0; t < T; ++t)
        XDFM_REQUIRE(!tensors[t].grad_marks || ((((size_t)tensors[t].param) | ((size_t)tensors[t].grad) | ((size_t)tensors[t].exp_avg) |
                                                  ((size_t)tensors[t].exp_avg_sq)) & 15) == 0,
                     "adam_step: tensor %d has grad_marks but a pointer that is not 16-byte aligned", t);
    hipStream_t st = (hipStream_t)stream;
    // Launch composition: tensors sorted by size and dealt round-robin to the launches, so that every launch streams
    // its share of the big tables and the small tensors' latency-bound blocks run underneath (a launch of small
    // tensors alone took 25 us for 30 MB).  The order is a pure function of the sizes: deterministic.
    const int nlaunch = ceil_div(T, ADAM_CHUNK);
    std::vector<int> order(T);
    for (int t = 0; t < T; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tensors[a].numel > tensors[b].numel; });
    int slot0 = 0;
    for (int l = 0; l < nlaunch; ++l) {
        AdamBatch batch;
        int cnt = 0;
        for (int k = l; k < T; k += nlaunch) batch.t[cnt++] = tensors[order[k]];
        for (int k = cnt; k < ADAM_CHUNK; ++k) batch.t[k] = batch.t[0];
        int bx = xdfm_opt(OPT_ADAM_BX);
        if (bx <= 0 || bx > ADAM_BX) bx = ADAM_BX;
        hipLaunchKernelGGL(adam_step_kernel, dim3(bx, cnt), dim3(ADAM_THREADS), 0, st, batch, slot0, lr, lr_dev, beta1, beta2, eps,
                           l2_value ? l2_ws : nullptr);
        slot0 += cnt;
    }
    if (l2_value) hipLaunchKernelGGL(adam_l2_finish_kernel, dim3(1), dim3(1024), 0, st, l2_ws, T * ADAM_BX, l2_value);
    return xdfm_check_launch("adam_step");
}

}  // extern "C"
