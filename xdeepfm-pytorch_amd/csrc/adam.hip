// K7: Adam step over all parameters (one launch per 40 tensors, a 1-D grid shared out by tensor size; the embedding
// tables are 98 % of the bytes).
//
// replaces torch.optim.Adam.step() for the [vocab, D] / [vocab, 1] tables (deepctr/models/basemodel.py:452
// builds torch.optim.Adam over every parameter; the reference's tables carry dense gradients,
// deepctr/inputs.py:168 sparse=False, so every row is updated every step).  44 M parameters at config 2 =
// 1.2 GB of traffic per step (p, m, v read + written, g read): pure HBM streaming, 16-byte accesses, 12-16 loads
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
#define ADAM_BX 512             // most blocks one tensor gets (tools/adam_probe.py: 128 .. 4096 are within 5 % of each other, 512 best)
#define ADAM_BLOCK_ELEMS 8192   // a tensor gets one block per this many elements: 8 float4 per thread and array

// Streaming accesses: p, m, v (and the touched gradients) are read once and written once per step, 14 GB of them at
// Criteo-card vocabularies -- non-temporal, four chunks per array and thread in flight.  tools/ubench/stream.hip is the
// bare read-modify-write stream of the same shape: 6.4-7.3 TB/s while the three arrays total 1.9 GB, 4.4-5.5 TB/s at
// the 7.7 GB this step sweeps -- K7 runs at 5.0-5.5 TB/s (tools/adam_probe.py), the rate the memory system gives
// this footprint.
typedef float adam_v4f __attribute__((ext_vector_type(4)));
// wave-uniform chunk pointer + this lane's byte offset (32-bit: the access becomes scalar base + vector offset)
__device__ __forceinline__ float4* at4(float4* base, unsigned byte_off) {
    return reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off);
}
template <bool NT>
__device__ __forceinline__ float4 adam_ld(const float4* a) {
    if constexpr (!NT) return *a;
    const adam_v4f t = __builtin_nontemporal_load(reinterpret_cast<const adam_v4f*>(a));
    return make_float4(t.x, t.y, t.z, t.w);
}
template <bool NT>
__device__ __forceinline__ void adam_st(float4* a, const float4& x) {
    if constexpr (!NT) { *a = x; return; }
    const adam_v4f t = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(t, reinterpret_cast<adam_v4f*>(a));
}

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
// first[k] = first block of tensor k in the launch's 1-D grid (first[cnt] = grid size): a tensor's share of the grid
// follows its size, so a launch that holds four 10 M-row tables and thirty small tensors is 16 000 blocks of table
// sweep, not 128 per tensor
struct AdamBatch { xdfm_adam_tensor t[ADAM_CHUNK]; int first[ADAM_CHUNK + 1]; };

template <bool NT>
__global__ __launch_bounds__(ADAM_THREADS, 3) void adam_step_kernel(
    const AdamBatch batch, int cnt, int slot0, double lr_arg, const double* __restrict__ lr_dev, double beta1, double beta2, double eps,
    float* __restrict__ l2_part) {
    // the learning rate as a kernel argument, or read from device memory (a captured HIP graph then follows a
    // learning-rate schedule without being captured again)
    const double lr = lr_dev ? *lr_dev : lr_arg;
    int ti = 0;                                        // wave-uniform search over <= 40 entries
    for (int k = 1; k < cnt; ++k) ti += (int)blockIdx.x >= batch.first[k] ? 1 : 0;
    const int lb = (int)blockIdx.x - batch.first[ti];  // this block among the tensor's nb blocks
    const int nb = batch.first[ti + 1] - batch.first[ti];
    const xdfm_adam_tensor& d = batch.t[ti];
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
    const long tid = (long)lb * ADAM_THREADS + threadIdx.x;
    const long stride = (long)nb * ADAM_THREADS;
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
        // mark, so nothing races).  Four chunks per array and thread in flight; the mark bytes are loaded first and
        // the (rare) gradient reads sit behind one branch, so the common iteration is a straight run of 16 loads.
        // The zero is opaque to the compiler: with a literal 0 it folds fmaf(2*l2, w, 0) into a product and then
        // contracts that product into the next subtraction, which rounds differently from the dense path -- with
        // the same instruction sequence both paths give the same bits.
        float zf;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zf));
        const float4 zero4 = make_float4(zf, zf, zf, zf);
        // whole-block iterations: the chunk index is (wave-uniform base) + threadIdx.x, so every access is a scalar
        // base plus one 32-bit lane offset (16 per-lane 64-bit addresses would not fit the register budget)
        long iu = (long)lb * ADAM_THREADS;
        const unsigned tx = threadIdx.x, tx16 = tx * 16u;
        for (; iu + 3 * stride + ADAM_THREADS <= n4; iu += 4 * stride) {
            unsigned char k[4];
            float4 P[4], M[4], V[4], G[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) k[q] = (marks + (iu + q * stride))[tx];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                P[q] = adam_ld<NT>(at4(p4 + (iu + q * stride), tx16)); M[q] = adam_ld<NT>(at4(m4 + (iu + q * stride), tx16));
                V[q] = adam_ld<NT>(at4(v4 + (iu + q * stride), tx16));
                G[q] = zero4;
            }
            if (k[0] | k[1] | k[2] | k[3]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k[q]) { G[q] = (*at4(g4 + (iu + q * stride), tx16)); (*at4(g4 + (iu + q * stride), tx16)) = zero4; (marks + (iu + q * stride))[tx] = 0; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 pa = P[q], ma = M[q], va = V[q], ga = G[q];
                sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
                ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
                adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
                adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
                adam_st<NT>(at4(p4 + (iu + q * stride), tx16), pa); adam_st<NT>(at4(m4 + (iu + q * stride), tx16), ma);
                adam_st<NT>(at4(v4 + (iu + q * stride), tx16), va);
                __builtin_amdgcn_sched_barrier(0);       // one chunk's arithmetic at a time: its temporaries die before the next
            }
        }
        i = iu + tx;
        for (; i < n4; i += stride) {
            const unsigned char ka = marks[i];
            float4 pa = adam_ld<NT>(p4 + i), ma = adam_ld<NT>(m4 + i), va = adam_ld<NT>(v4 + i);
            float4 ga = zero4;
            if (ka) { ga = g4[i]; g4[i] = zero4; marks[i] = 0; }
            sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
            ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
            adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
            adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
            adam_st<NT>(p4 + i, pa); adam_st<NT>(m4 + i, ma); adam_st<NT>(v4 + i, va);
        }
    }
    long iu = i - threadIdx.x;                            // dense gradient: four float4 per array in flight
    const unsigned tx = threadIdx.x, tx16 = tx * 16u;
    for (; iu + 3 * stride + ADAM_THREADS <= n4; iu += 4 * stride) {
        float4 P[4], M[4], V[4], G[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            P[q] = adam_ld<NT>(at4(p4 + (iu + q * stride), tx16)); G[q] = adam_ld<NT>(at4(g4 + (iu + q * stride), tx16));
            M[q] = adam_ld<NT>(at4(m4 + (iu + q * stride), tx16)); V[q] = adam_ld<NT>(at4(v4 + (iu + q * stride), tx16));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 pa = P[q], ma = M[q], va = V[q], ga = G[q];
            sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
            ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
            adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
            adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
            adam_st<NT>(at4(p4 + (iu + q * stride), tx16), pa); adam_st<NT>(at4(m4 + (iu + q * stride), tx16), ma);
            adam_st<NT>(at4(v4 + (iu + q * stride), tx16), va);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    i = iu + tx;
    for (; i < n4; i += stride) {
        float4 pa = adam_ld<NT>(p4 + i), ga = adam_ld<NT>(g4 + i), ma = adam_ld<NT>(m4 + i), va = adam_ld<NT>(v4 + i);
        sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
        ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
        adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
        adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
        adam_st<NT>(p4 + i, pa); adam_st<NT>(m4 + i, ma); adam_st<NT>(v4 + i, va);
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
        // one slot per block of the step, in launch order: the finish kernel adds them in that fixed order
        if (threadIdx.x == 0) l2_part[slot0 + blockIdx.x] = l2c * ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
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
        // blocks per tensor by size (option "adam_bx" caps them: a small footprint for experiments)
        int cap = xdfm_opt(OPT_ADAM_BX);
        if (cap <= 0 || cap > ADAM_BX) cap = ADAM_BX;
        batch.first[0] = 0;
        for (int k = 0; k < ADAM_CHUNK; ++k) {
            long nb = k < cnt ? ceil_div(batch.t[k].numel, (long)ADAM_BLOCK_ELEMS) : 0;
            if (k < cnt && nb < 1) nb = 1;
            if (nb > cap) nb = cap;
            batch.first[k + 1] = batch.first[k] + (int)nb;
        }
        if (xdfm_opt(OPT_DBG) & (1 << 17))              // experiment: ordinary (cached) loads and stores
            hipLaunchKernelGGL(adam_step_kernel<false>, dim3(batch.first[cnt]), dim3(ADAM_THREADS), 0, st, batch, cnt, slot0, lr,
                               lr_dev, beta1, beta2, eps, l2_value ? l2_ws : nullptr);
        else
            hipLaunchKernelGGL(adam_step_kernel<true>, dim3(batch.first[cnt]), dim3(ADAM_THREADS), 0, st, batch, cnt, slot0, lr,
                               lr_dev, beta1, beta2, eps, l2_value ? l2_ws : nullptr);
        slot0 += batch.first[cnt];
    }
    if (l2_value) hipLaunchKernelGGL(adam_l2_finish_kernel, dim3(1), dim3(1024), 0, st, l2_ws, slot0, l2_value);
    return xdfm_check_launch("adam_step");
}

}  // extern "C"
