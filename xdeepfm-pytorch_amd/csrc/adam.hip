// K7: Adam step over all parameters (one launch per 52 tensors, a 1-D grid shared out by tensor size; the embedding
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
#include "adam_math.h"

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

#define ADAM_FIX 1099511627776.0          // 2^40: fixed-point scale of the L2 backlog (integer adds: order-independent)

#define ADAM_CHUNK 64
// first[k] = first block of tensor k in the launch's 1-D grid (first[cnt] = grid size): a tensor's share of the grid
// follows its size, so a launch that holds four 10 M-row tables and thirty small tensors is 16 000 blocks of table
// sweep, not 128 per tensor
// the descriptor as the kernels see it: 72 bytes, so that 52 of them (the criteo-card step holds 51 tensors outside the
// by-rows tables: one launch instead of two 35-us launches of small, latency-bound tensors) and the other arguments stay
// inside the 4 KB a kernel's argument block may take
struct AdamDev {
    float* param; float* grad; float* exp_avg; float* exp_avg_sq; const float* step; unsigned char* grad_marks; unsigned char* last;
    long numel; float l2; int flags;
};
static inline AdamDev adam_dev(const xdfm_adam_tensor& t) {
    return AdamDev{t.param, t.grad, t.exp_avg, t.exp_avg_sq, t.step, t.grad_marks, t.last, t.numel, t.l2, t.flags};
}
struct AdamBatch { AdamDev t[ADAM_CHUNK]; int first[ADAM_CHUNK + 1]; };
static_assert(sizeof(AdamBatch) + 128 <= 8192, "the Adam kernels' argument block");

template <bool NT>
__global__ __launch_bounds__(ADAM_THREADS, 3) void adam_step_kernel(
    const AdamBatch batch, int cnt, int slot0, double lr_arg, const double* __restrict__ lr_dev, double beta1, double beta2, double eps,
    float* __restrict__ l2_part, const int* __restrict__ clock, const float* __restrict__ consts, float* __restrict__ l2_out,
    int l2_total, unsigned* __restrict__ ticket) {
    // the learning rate as a kernel argument, or read from device memory (a captured HIP graph then follows a
    // learning-rate schedule without being captured again)
    const double lr = lr_dev ? *lr_dev : lr_arg;
    int ti = 0;                                        // wave-uniform search over <= 40 entries
    for (int k = 1; k < cnt; ++k) ti += (int)blockIdx.x >= batch.first[k] ? 1 : 0;
    const int lb = (int)blockIdx.x - batch.first[ti];  // this block among the tensor's nb blocks
    const int nb = batch.first[ti + 1] - batch.first[ti];
    const AdamDev& d = batch.t[ti];
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
        // with a clock, adam_tick has just written this step's two constants (same formula, same doubles): take them from
        // its table when the tensor's own step counter agrees with the clock -- two double pow() cost microseconds
        const int t = clock ? clock[0] : 0;
        if (clock && step == (double)(clock[1] + t)) {
            bcs[0] = consts[ADAM_CONSTS_PER_STEP * t];
            bcs[1] = consts[ADAM_CONSTS_PER_STEP * t + 1];
        } else {
            bcs[0] = (float)(lr / (1.0 - pow(beta1, step)));
            bcs[1] = (float)sqrt(1.0 - pow(beta2, step));
        }
    }
    __syncthreads();
    const float step_size = bcs[0], bc2_sqrt = bcs[1];
    const AdamCoef c = {(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)lr, (float)eps};
    const float l2c = *l2;
    const float g2 = 2.f * l2c;                        // d(l2 * w^2)/dw = 2 l2 w
    float sq = 0.f;
    adam_f2 sq2 = {0.f, 0.f};                          // squares of the replayed steps (deferred tensors)
    const bool eps_ok = c.eps >= ADAM_EPS_MIN && c.eps <= 1.f;
    const long tid = (long)lb * ADAM_THREADS + threadIdx.x;
    const long stride = (long)nb * ADAM_THREADS;
    const bool vec = ((((size_t)p) | ((size_t)m) | ((size_t)v) | ((size_t)g)) & 15) == 0;
    const long n4 = vec ? n / 4 : 0;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    float4* g4 = reinterpret_cast<float4*>(g);
    long i = tid;
    if (marks && (d.flags & XDFM_ADAM_DEFERRED) && clock) {
        // deferred (exact) update, see xdfm.h: only chunks with a gradient are touched; a chunk that still misses earlier
        // steps (its rows were not brought up to date by xdfm_adam_catchup_rows) replays them first
        const int t = clock[0];
        unsigned char* __restrict__ last = d.last;
        float zf;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zf));
        const float4 zero4 = make_float4(zf, zf, zf, zf);
        auto process = [&](long e) {
            float4 pa = p4[e], ma = m4[e], va = v4[e], ga = g4[e];
            g4[e] = zero4; marks[e] = 0;
            for (int s = (int)last[e] + 1; s < t; ++s) adam_replay4(pa, ma, va, adam_step_const(consts, s), g2, zf, c, sq2, eps_ok);
            sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
            ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
            adam_one(pa.x, ga.x, ma.x, va.x, step_size, bc2_sqrt, c); adam_one(pa.y, ga.y, ma.y, va.y, step_size, bc2_sqrt, c);
            adam_one(pa.z, ga.z, ma.z, va.z, step_size, bc2_sqrt, c); adam_one(pa.w, ga.w, ma.w, va.w, step_size, bc2_sqrt, c);
            p4[e] = pa; m4[e] = ma; v4[e] = va;
            last[e] = (unsigned char)t;
        };
        // The scan reads the mark bytes 16 at a time (one uint4 per lane and load: the sweep over 144 M marks at Criteo-card
        // vocabularies is what this step costs); the chunks in front of the first 16-byte boundary of the marks array and
        // behind the last whole group go one by one.
        const long head = ((16 - (long)((size_t)marks & 15)) & 15) < n4 ? ((16 - (long)((size_t)marks & 15)) & 15) : n4;
        const long groups = (n4 - head) / 16;
        const uint4* __restrict__ m16 = reinterpret_cast<const uint4*>(marks + head);
        long g = tid;
        for (; g + stride < groups; g += 2 * stride) {                       // two groups per thread in flight
            const uint4 wa = m16[g], wb = m16[g + stride];
            if (wa.x | wa.y | wa.z | wa.w) {
                const unsigned w[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
                for (int b = 0; b < 16; ++b)
                    if ((w[b >> 2] >> ((b & 3) * 8)) & 255u) process(head + g * 16 + b);
            }
            if (wb.x | wb.y | wb.z | wb.w) {
                const unsigned w[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                for (int b = 0; b < 16; ++b)
                    if ((w[b >> 2] >> ((b & 3) * 8)) & 255u) process(head + (g + stride) * 16 + b);
            }
        }
        if (g < groups) {                               // this thread's last, unpaired group
            const uint4 wa = m16[g];
            if (wa.x | wa.y | wa.z | wa.w) {
                const unsigned w[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
                for (int b = 0; b < 16; ++b)
                    if ((w[b >> 2] >> ((b & 3) * 8)) & 255u) process(head + g * 16 + b);
            }
        }
        for (long e = tid; e < head; e += stride)
            if (marks[e]) process(e);
        for (long e = head + groups * 16 + tid; e < n4; e += stride)
            if (marks[e]) process(e);
        i = n4 + tid;                                   // nothing left for the dense loops below
    } else if (marks && (d.flags & XDFM_ADAM_LAZY)) {
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
        sq += sq2.x + sq2.y;
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        __shared__ float wsum[ADAM_THREADS / 64];
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sq;
        __syncthreads();
        // one slot per block of the step, in launch order: the finish kernel adds them in that fixed order
        if (threadIdx.x == 0) xdfm_publish(&l2_part[slot0 + blockIdx.x], l2c * ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])));
        // the step's last launch: its last block adds up the partials of ALL launches (the earlier ones are complete: stream
        // order), in slot order -- what adam_l2_finish_kernel does in a launch of its own
        if (ticket && xdfm_last_block_done(ticket, gridDim.x)) {
            __shared__ float acc[ADAM_THREADS];
            float v = 0.f;
            for (int k = threadIdx.x; k < l2_total; k += ADAM_THREADS) v += xdfm_peer(l2_part + k);
            acc[threadIdx.x] = v;
            __syncthreads();
            for (int o = ADAM_THREADS / 2; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) acc[threadIdx.x] += acc[threadIdx.x + o];
                __syncthreads();
            }
            if (threadIdx.x == 0) l2_out[0] = acc[0];
        }
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

// ---------------------------------------------------------------------------------------------
// deferred Adam: clock, catch-up of the rows a batch gathers, flush (xdfm.h, "K7d")
// ---------------------------------------------------------------------------------------------
__global__ void adam_tick_kernel(int* __restrict__ clock, float* __restrict__ consts, int cap, double lr_arg,
                                 const double* __restrict__ lr_dev, double beta1, double beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double lr = lr_dev ? *lr_dev : lr_arg;
    int t = clock[0] + 1;
    if (t >= cap) t = cap - 1;                          // the host flushes long before (defensive)
    clock[0] = t;
    const double step = (double)(clock[1] + t);         // the value the per-parameter step counters hold at this step
    const float ss = (float)(lr / (1.0 - pow(beta1, step)));
    const float bc = (float)sqrt(1.0 - pow(beta2, step));
    // the replay's short forms (adam_math.h) are proven for constants in these ranges; outside them rc2 = 0 sends the
    // step through the reference spelling
    const bool ok = bc >= 0.00390625f && bc <= 1.f && ss >= 9.313225746154785e-10f && ss <= 1024.f;    // 2^-8, 2^-30, 2^10
    float* k = consts + ADAM_CONSTS_PER_STEP * t;
    k[0] = ss;
    k[1] = bc;
    k[2] = bc * 4294967296.f;                           // 2^32
    k[3] = ok ? adam_exact_rcp(bc) * 2.3283064365386963e-10f : 0.f;     // 2^-32
}

__global__ void adam_clock_reset_kernel(int* __restrict__ clock) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { clock[1] += clock[0]; clock[0] = 0; }
}

// block sum of l2-weighted squares in fixed point (integer adds are exact: the total does not depend on which thread
// replayed which chunk, nor on the order of the blocks), one atomic per block
__device__ __forceinline__ void adam_backlog_add(float v, unsigned long long* __restrict__ backlog) {
    long long f = (long long)((double)v * ADAM_FIX);
    for (int o = 32; o > 0; o >>= 1) f += __shfl_xor(f, o);
    __shared__ long long part[ADAM_THREADS / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tot = 0;
        for (int k = 0; k < ADAM_THREADS / 64; ++k) tot += part[k];
        if (tot) atomicAdd(backlog, (unsigned long long)tot);
    }
}

struct AdamRowsDev {
    float* const* p; float* const* m; float* const* v; unsigned char* const* last; const float* l2;
    float* const* g; unsigned char* const* marks;
};
// the embedding or the linear tables' pointer arrays, picked field by field (a reference chosen at run time between the
// two kernel-argument structs makes hipcc copy both to scratch and index them there: 120 bytes of private memory per lane)
__device__ __forceinline__ AdamRowsDev adam_rows_pick(bool is_lin, const AdamRowsDev& emb, const AdamRowsDev& lin) {
    AdamRowsDev R;
    R.p = is_lin ? lin.p : emb.p; R.m = is_lin ? lin.m : emb.m; R.v = is_lin ? lin.v : emb.v;
    R.last = is_lin ? lin.last : emb.last; R.l2 = is_lin ? lin.l2 : emb.l2;
    R.g = is_lin ? lin.g : emb.g; R.marks = is_lin ? lin.marks : emb.marks;
    return R;
}

// The chunk of thread idx -- (example, field, chunk of the row) -- and its claim: `old` >= 0 when this thread is the first
// to reach the chunk in this launch (CAS on the word that holds its `last` byte) and has to bring it from step `old` to t;
// duplicates of an id lose the claim and skip.
struct AdamClaim { int f; long cc; int old; bool is_lin; };
__device__ __forceinline__ AdamClaim adam_claim_chunk(long idx, const float* __restrict__ X, long ldx, const int* __restrict__ cols,
                                                      const int* __restrict__ vocab, int m, int D, int QE, int QT,
                                                      const AdamRowsDev& emb, const AdamRowsDev& lin, int t, bool need_table) {
    AdamClaim k;
    const int q = (int)(idx % QT);
    const long r = idx / QT;
    k.f = (int)(r % m);
    const long b = r / m;
    const int V = vocab[k.f];
    long id = (long)X[b * ldx + cols[k.f]];             // as the gather (embed.hip): truncation, clamped
    if (id < 0 || id >= V) id = id < 0 ? 0 : V - 1;
    k.is_lin = q >= QE;
    const AdamRowsDev R = adam_rows_pick(k.is_lin, emb, lin);
    const long w = k.is_lin ? 1 : D;
    const long c0 = id * w / 4, c1 = (id * w + w - 1) / 4;
    k.cc = c0 + (k.is_lin ? 0 : q);
    const long n4 = (long)V * w / 4;                    // whole chunks; the tail elements are updated densely every step
    k.old = -1;
    if (k.cc <= c1 && k.cc < n4 && (!need_table || R.p[k.f] != nullptr)) {      // a null table: left to the step's mark scan (small tables)
        unsigned char* last = R.last[k.f];
        unsigned* word = reinterpret_cast<unsigned*>(last + (k.cc & ~3L));
        const int sh = (int)(k.cc & 3) * 8;
        unsigned seen = *word;          // a plain (cached) read: stale at worst, and then the CAS below returns the current word
        while (true) {
            const int ob = (int)((seen >> sh) & 255u);
            if (ob >= t) break;
            const unsigned want = (seen & ~(255u << sh)) | ((unsigned)t << sh);
            const unsigned got = atomicCAS(word, seen, want);
            if (got == seen) { k.old = ob; break; }
            seen = got;
        }
    }
    return k;
}

// One thread per (example, field, chunk of the row): the first thread to reach a chunk replays the steps it misses.  The
// gather runs in a later launch.
__global__ __launch_bounds__(ADAM_THREADS) void adam_catchup_rows_kernel(
    const float* __restrict__ X, long ldx, int B, const int* __restrict__ cols, const int* __restrict__ vocab, int m, int D,
    AdamRowsDev emb, AdamRowsDev lin, int has_lin, const int* __restrict__ clock, const float* __restrict__ consts,
    double beta1, double beta2, double eps, unsigned long long* __restrict__ backlog) {
    const int t = clock[0];
    const int QE = (D + 3) / 4 + ((D & 3) ? 1 : 0);     // chunks a row of D floats can straddle
    const int QT = QE + (has_lin ? 1 : 0);
    const long idx = (long)blockIdx.x * ADAM_THREADS + threadIdx.x;
    const AdamCoef c = {(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 0.f, (float)eps};
    const bool eps_ok = c.eps >= ADAM_EPS_MIN && c.eps <= 1.f;
    float zf;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zf));
    AdamClaim k;
    k.old = -1; k.f = 0; k.cc = 0; k.is_lin = false;
    if (t > 0 && idx < (long)B * m * QT) k = adam_claim_chunk(idx, X, ldx, cols, vocab, m, D, QE, QT, emb, lin, t, false);
    const bool act = k.old >= 0;
    const AdamRowsDev R = adam_rows_pick(k.is_lin, emb, lin);
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), ma = pa, va = pa;
    float4 *p4 = nullptr, *m4 = nullptr, *v4 = nullptr;
    float l2c = 0.f;
    if (act) {
        p4 = reinterpret_cast<float4*>(R.p[k.f]) + k.cc;
        m4 = reinterpret_cast<float4*>(R.m[k.f]) + k.cc;
        v4 = reinterpret_cast<float4*>(R.v[k.f]) + k.cc;
        pa = *p4; ma = *m4; va = *v4;
        l2c = R.l2[k.f];
    }
    adam_f2 sq2 = {0.f, 0.f};
    adam_replay_span(pa, ma, va, act, k.old, t, consts, 2.f * l2c, zf, c, sq2, eps_ok);
    if (act) { *p4 = pa; *m4 = ma; *v4 = va; }
    adam_backlog_add(l2c * (sq2.x + sq2.y), backlog);
}

// The step's update of the deferred tables, by the batch's rows (single process: every marked chunk belongs to a row
// of X).  Threads [0, B*m*QT): one per (example, field, chunk of the row), first come first served through `last` as in
// the catch-up; then 4 * m threads per table kind for the numel % 4 tail elements, which no chunk covers.
__global__ __launch_bounds__(ADAM_THREADS) void adam_apply_rows_kernel(
    const float* __restrict__ X, long ldx, int B, const int* __restrict__ cols, const int* __restrict__ vocab, int m, int D,
    AdamRowsDev emb, AdamRowsDev lin, int has_lin, const int* __restrict__ clock, const float* __restrict__ consts,
    double beta1, double beta2, double eps, unsigned long long* __restrict__ cell, float* __restrict__ l2_value,
    unsigned* __restrict__ ticket) {
    const int t = clock[0];
    const int QE = (D + 3) / 4 + ((D & 3) ? 1 : 0);
    const int QT = QE + (has_lin ? 1 : 0);
    const long idx = (long)blockIdx.x * ADAM_THREADS + threadIdx.x;
    const long nrow = (long)B * m * QT;
    const float ss = consts[ADAM_CONSTS_PER_STEP * t], bc = consts[ADAM_CONSTS_PER_STEP * t + 1];
    const AdamCoef c = {(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 0.f, (float)eps};
    const bool eps_ok = c.eps >= ADAM_EPS_MIN && c.eps <= 1.f;
    float zf;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zf));
    float sqv = 0.f;
    AdamClaim k;
    k.old = -1; k.f = 0; k.cc = 0; k.is_lin = false;
    if (idx < nrow) k = adam_claim_chunk(idx, X, ldx, cols, vocab, m, D, QE, QT, emb, lin, t, true);
    const bool act = k.old >= 0;
    {
        const AdamRowsDev R = adam_rows_pick(k.is_lin, emb, lin);
        float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), ma = pa, va = pa, ga = pa;
        float4 *p4 = nullptr, *m4 = nullptr, *v4 = nullptr;
        float l2c = 0.f;
        if (act) {
            p4 = reinterpret_cast<float4*>(R.p[k.f]) + k.cc;
            m4 = reinterpret_cast<float4*>(R.m[k.f]) + k.cc;
            v4 = reinterpret_cast<float4*>(R.v[k.f]) + k.cc;
            float4* g4 = reinterpret_cast<float4*>(R.g[k.f]) + k.cc;
            pa = *p4; ma = *m4; va = *v4; ga = *g4;
            *g4 = make_float4(zf, zf, zf, zf);
            R.marks[k.f][k.cc] = 0;
            l2c = R.l2[k.f];
        }
        const float g2 = 2.f * l2c;
        adam_f2 sq2 = {0.f, 0.f};
        adam_replay_span(pa, ma, va, act, k.old, t - 1, consts, g2, zf, c, sq2, eps_ok);     // the steps the chunk still misses
        if (act) {                                                                           // then this step, with its gradient
            float sq = sq2.x + sq2.y;
            sq += (pa.x * pa.x + pa.y * pa.y) + (pa.z * pa.z + pa.w * pa.w);
            ga.x = fmaf(g2, pa.x, ga.x); ga.y = fmaf(g2, pa.y, ga.y); ga.z = fmaf(g2, pa.z, ga.z); ga.w = fmaf(g2, pa.w, ga.w);
            adam_one(pa.x, ga.x, ma.x, va.x, ss, bc, c); adam_one(pa.y, ga.y, ma.y, va.y, ss, bc, c);
            adam_one(pa.z, ga.z, ma.z, va.z, ss, bc, c); adam_one(pa.w, ga.w, ma.w, va.w, ss, bc, c);
            *p4 = pa; *m4 = ma; *v4 = va;
            sqv = l2c * sq;
        }
    }
    if (idx >= nrow && idx < nrow + 8L * m) {
        // tail elements: (table kind, field, element k < 4) -- updated every step, like the sweep does
        const long u = idx - nrow;
        const int kk = (int)(u & 3);
        const int f = (int)((u >> 2) % m);
        const bool is_lin = (u >> 2) >= m;
        if (!is_lin || has_lin) {
            const AdamRowsDev R = adam_rows_pick(is_lin, emb, lin);
            const long numel = (long)vocab[f] * (is_lin ? 1 : D);
            const long e = numel / 4 * 4 + kk;
            if (e < numel && R.p[f] != nullptr) {
                float* p = R.p[f]; float* mm = R.m[f]; float* vv = R.v[f]; float* g = R.g[f];
                const float l2c = R.l2[f];
                float pa = p[e], ma = mm[e], va = vv[e];
                sqv = l2c * (pa * pa);
                adam_one(pa, fmaf(2.f * l2c, pa, g[e]), ma, va, ss, bc, c);
                p[e] = pa; mm[e] = ma; vv[e] = va;
                g[e] = 0.f;
                R.marks[f][e >> 2] = 0;
            }
        }
    }
    adam_backlog_add(sqv, cell);
    // adam_rows_finish_kernel's job, by the block that finishes last (every block's integer add is in `cell` by then)
    if (ticket && xdfm_last_block_done(ticket, gridDim.x) && threadIdx.x == 0) {
        const unsigned long long tot = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (l2_value) l2_value[0] += (float)((double)(long long)tot / ADAM_FIX);
        __hip_atomic_store(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void adam_rows_finish_kernel(unsigned long long* __restrict__ cell, float* __restrict__ l2_value) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (l2_value) l2_value[0] += (float)((double)(long long)cell[0] / ADAM_FIX);
        cell[0] = 0ull;
    }
}

// Every chunk of the deferred tensors up to the clock; `last` back to 0.  Same 1-D grid as the step.
__global__ __launch_bounds__(ADAM_THREADS) void adam_flush_kernel(const AdamBatch batch, int cnt, const int* __restrict__ clock,
                                                                  const float* __restrict__ consts, double beta1, double beta2,
                                                                  double eps, unsigned long long* __restrict__ backlog) {
    int ti = 0;
    for (int k = 1; k < cnt; ++k) ti += (int)blockIdx.x >= batch.first[k] ? 1 : 0;
    const int lb = (int)blockIdx.x - batch.first[ti];
    const int nb = batch.first[ti + 1] - batch.first[ti];
    const AdamDev& d = batch.t[ti];
    const int t = clock[0];
    float4* p4 = reinterpret_cast<float4*>(d.param);
    float4* m4 = reinterpret_cast<float4*>(d.exp_avg);
    float4* v4 = reinterpret_cast<float4*>(d.exp_avg_sq);
    unsigned char* __restrict__ last = d.last;
    const long n4 = d.numel / 4;
    const float l2c = d.l2;
    const float g2 = 2.f * l2c;
    float zf;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zf));
    const AdamCoef c = {(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 0.f, (float)eps};
    adam_f2 sq2 = {0.f, 0.f};
    const bool eps_ok = c.eps >= ADAM_EPS_MIN && c.eps <= 1.f;
    // whole-wave iterations (the replay lines the lanes up on the step number): the tail lanes are masked, not absent
    for (long base = (long)lb * ADAM_THREADS; base < n4; base += (long)nb * ADAM_THREADS) {
        const long i = base + threadIdx.x;
        const bool in = i < n4;
        const int old = in ? (int)last[i] : 0;
        const bool act = in && old < t;
        float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), ma = pa, va = pa;
        if (act) { pa = adam_ld<true>(p4 + i); ma = adam_ld<true>(m4 + i); va = adam_ld<true>(v4 + i); }
        adam_replay_span(pa, ma, va, act, old, t, consts, g2, zf, c, sq2, eps_ok);
        if (act) { adam_st<true>(p4 + i, pa); adam_st<true>(m4 + i, ma); adam_st<true>(v4 + i, va); }
        if (in && old) last[i] = 0;
    }
    adam_backlog_add(l2c * (sq2.x + sq2.y), backlog);
}

// ---------------------------------------------------------------------------------------------
// Self-test of the replay's short forms against the reference spellings (adam_math.h), on the device, for the tests.
// out[0] = cases compared, out[1] = mismatches, out[2] = an encoding of one mismatching case.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long adam_mix(unsigned long long z) {       // splitmix64
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float adam_rand_float(unsigned long long h, int emin, int emax, bool sign) {     // 2^[emin, emax) x [1, 2)
    const int e = emin + (int)((h >> 32) % (unsigned)(emax - emin));
    const unsigned bits = ((unsigned)(e + 127) << 23) | ((unsigned)h & 0x7fffffu) | ((sign && ((h >> 60) & 1)) ? 0x80000000u : 0u);
    return __uint_as_float(bits);
}
__device__ __forceinline__ AdamStepConst adam_const_for(double lr, double beta1, double beta2, double step) {
    AdamStepConst k;
    k.ss = (float)(lr / (1.0 - pow(beta1, step)));
    k.bc = (float)sqrt(1.0 - pow(beta2, step));
    k.c2 = k.bc * 4294967296.f;
    k.rc2 = adam_exact_rcp(k.bc) * 2.3283064365386963e-10f;
    return k;
}
__global__ __launch_bounds__(256) void adam_selftest_kernel(int mode, unsigned long long base, unsigned long long n,
                                                            unsigned long long seed, double lr, double beta1, double beta2, double eps,
                                                            unsigned long long* __restrict__ out) {
#pragma clang fp contract(off)
    const unsigned long long idx = base + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0, done = 0, what = 0;
    if (idx < n) {
        if (mode == 0) {                    // square root: every bit pattern; compared where the replay uses it (0 < x < 2^62)
            const float x = __uint_as_float((unsigned)idx);
            if (x > 0.f && x < ADAM_V_MAX) {
                const float x2 = x * 18446744073709551616.f;
                const float fast = adam_sqrt_scaled2(adam_f2{x2, x2}).x, ref = sqrtf(x) * 4294967296.f;
                done = 1; bad = __float_as_uint(fast) != __float_as_uint(ref); what = idx;
            }
        } else if (mode == 1) {             // division by the step's constant: every numerator of two binades x steps 1..
            const unsigned mant = (unsigned)(idx & 0xffffffull);
            const unsigned long long tt = (idx >> 24) + 1;
            // beyond the first 256 steps: random constants in [2^-8, 1)
            const float bc = tt <= 256 ? (float)sqrt(1.0 - pow(beta2, (double)tt)) : adam_rand_float(adam_mix(seed + tt), -8, 0, false);
            const float S = __uint_as_float(0x3f800000u + mant);
            const float s2 = S * 4294967296.f;
            const float fast = adam_div_const2(adam_f2{s2, s2}, bc * 4294967296.f, adam_exact_rcp(bc) * 2.3283064365386963e-10f).y;
            const float ref = S / bc;
            done = 1; bad = __float_as_uint(fast) != __float_as_uint(ref); what = idx;
        } else if (mode == 2) {             // general division inside the guard: d in [2^-40, 2^40), |n| in [2^-80, 2^40) or 0
            const unsigned long long h0 = adam_mix(seed + 2 * idx), h1 = adam_mix(seed + 2 * idx + 1);
            const float d = adam_rand_float(h0, -40, 40, false);
            float nn = adam_rand_float(h1, -80, 40, true);
            if ((h1 >> 40) % 1021 == 0) nn = 0.f;
            if ((h1 >> 40) % 1021 == 1) nn = __uint_as_float(((h0 >> 61) & 1) ? 0x17800000u : 0x537fffffu);    // 2^-80, just under 2^40
            const float fast = adam_div2(adam_f2{nn, nn}, adam_f2{d, d}).x, ref = nn / d;
            done = 1; bad = __float_as_uint(fast) != __float_as_uint(ref) && !(fast == 0.f && ref == 0.f);
            what = ((unsigned long long)__float_as_uint(nn) << 32) | __float_as_uint(d);
        } else {                            // whole replayed steps against adam_one, states incl. zeros / denormals / huge values
            const AdamCoef c = {(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 0.f, (float)eps};
            const bool eps_ok = c.eps >= ADAM_EPS_MIN && c.eps <= 1.f;
            float zf;
            asm volatile("v_mov_b32 %0, 0" : "=v"(zf));
            float pv[4], mv[4], vv[4];
            for (int e = 0; e < 4; ++e) {
                const unsigned long long h = adam_mix(seed + 16 * idx + e), h2 = adam_mix(h), h3 = adam_mix(h2);
                pv[e] = adam_rand_float(h, -30, 8, true);
                mv[e] = adam_rand_float(h2, -60, 4, true);
                vv[e] = adam_rand_float(h3, -100, 10, false);
                const unsigned sel = (unsigned)((h >> 48) % 97);
                if (sel == 0) pv[e] = 0.f;
                if (sel == 1) { mv[e] = 0.f; vv[e] = 0.f; }
                if (sel == 2) pv[e] = __uint_as_float(0x00000fffu);                   // denormal weight
                if (sel == 3) { pv[e] = 3e30f; vv[e] = 1e35f; }
                if (sel == 4) mv[e] = __uint_as_float(0x00400000u);                   // denormal first moment
                if (sel == 5) vv[e] = __uint_as_float(0x00000001u);                   // smallest second moment
            }
            const float g2 = (adam_mix(seed + idx) & 7) == 0 ? 0.f : 2e-5f;
            float4 pf = make_float4(pv[0], pv[1], pv[2], pv[3]), mf = make_float4(mv[0], mv[1], mv[2], mv[3]),
                   vf = make_float4(vv[0], vv[1], vv[2], vv[3]);
            for (int st = 1; st <= 6 && !bad; ++st) {
                const AdamStepConst k = adam_const_for(lr, beta1, beta2, (double)(st + (idx % 50) * 7));
                float4 pr = pf, mr = mf, vr = vf;
                adam_f2 sq2 = {0.f, 0.f};
                adam_replay4(pf, mf, vf, k, g2, zf, c, sq2, eps_ok);
                float* a[3] = {&pr.x, &mr.x, &vr.x};
                for (int e = 0; e < 4; ++e) adam_one(a[0][e], fmaf(g2, a[0][e], zf), a[1][e], a[2][e], k.ss, k.bc, c);
                const float* f[3] = {&pf.x, &mf.x, &vf.x};
                for (int q = 0; q < 3; ++q)
                    for (int e = 0; e < 4; ++e) {
                        const float x = f[q][e], y = a[q][e];
                        if (__float_as_uint(x) != __float_as_uint(y) && !(x != x && y != y) && !(x == 0.f && y == 0.f)) bad = 1;
                    }
                done += 1;
                what = idx * 8 + st;
            }
        }
    }
    if (done) atomicAdd(&out[0], done);
    if (bad) { atomicAdd(&out[1], 1ull); atomicMax(&out[2], what); }
}

extern "C" {

size_t xdfm_adam_step_ws_elems(int T) { return T > 0 ? (size_t)T * ADAM_BX : 0; }

int xdfm_adam_step(const xdfm_adam_tensor* tensors, int T, double lr, double beta1, double beta2, double eps,
                   float* l2_ws, float* l2_value, void* stream) {
    return xdfm_adam_step_lr(tensors, T, lr, nullptr, beta1, beta2, eps, l2_ws, l2_value, stream);
}

static int adam_step_impl(const xdfm_adam_tensor* tensors, int T, const xdfm_adam_clock* clk, double lr, const double* lr_dev,
                          double beta1, double beta2, double eps, float* l2_ws, float* l2_value, void* stream);

int xdfm_adam_step_lr(const xdfm_adam_tensor* tensors, int T, double lr, const double* lr_dev, double beta1, double beta2,
                      double eps, float* l2_ws, float* l2_value, void* stream) {
    return adam_step_impl(tensors, T, nullptr, lr, lr_dev, beta1, beta2, eps, l2_ws, l2_value, stream);
}

int xdfm_adam_step_deferred(const xdfm_adam_tensor* tensors, int T, const xdfm_adam_clock* clk, double lr,
                            const double* lr_dev, double beta1, double beta2, double eps, float* l2_ws, float* l2_value,
                            void* stream) {
    XDFM_REQUIRE(clk && clk->clock && clk->consts && clk->cap > 2, "adam_step_deferred: bad clock");
    return adam_step_impl(tensors, T, clk, lr, lr_dev, beta1, beta2, eps, l2_ws, l2_value, stream);
}

static int adam_step_impl(const xdfm_adam_tensor* tensors, int T, const xdfm_adam_clock* clk, double lr, const double* lr_dev,
                          double beta1, double beta2, double eps, float* l2_ws, float* l2_value, void* stream) {
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
        XDFM_REQUIRE(!(tensors[t].flags & XDFM_ADAM_DEFERRED) || (clk && tensors[t].grad_marks && tensors[t].last),
                     "adam_step: tensor %d is deferred but has no clock / grad_marks / last", t);
    for (int t = 0; t < T; ++t)
        XDFM_REQUIRE(!tensors[t].grad_marks || ((((size_t)tensors[t].param) | ((size_t)tensors[t].grad) | ((size_t)tensors[t].exp_avg) |
                                                  ((size_t)tensors[t].exp_avg_sq)) & 15) == 0,
                     "adam_step: tensor %d has grad_marks but a pointer that is not 16-byte aligned", t);
    hipStream_t st = (hipStream_t)stream;
    const int* clock = clk ? clk->clock : nullptr;
    const float* consts = clk ? clk->consts : nullptr;
    if (clk) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, st, clk->clock, clk->consts, clk->cap, lr, lr_dev, beta1, beta2);
    // Launch composition: tensors sorted by size and dealt round-robin to the launches, so that every launch streams
    // its share of the big tables and the small tensors' latency-bound blocks run underneath (a launch of small
    // tensors alone took 25 us for 30 MB).  The order is a pure function of the sizes: deterministic.
    const int nlaunch = ceil_div(T, ADAM_CHUNK);
    unsigned* ticket = xdfm_ticket(TK_ADAM_L2);
    std::vector<int> order(T);
    for (int t = 0; t < T; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tensors[a].numel > tensors[b].numel; });
    int slot0 = 0;
    for (int l = 0; l < nlaunch; ++l) {
        AdamBatch batch;
        int cnt = 0;
        for (int k = l; k < T; k += nlaunch) batch.t[cnt++] = adam_dev(tensors[order[k]]);
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
        unsigned* tk = (l2_value && l == nlaunch - 1) ? ticket : nullptr;
        const int l2_total = slot0 + batch.first[cnt];
        if (xdfm_opt(OPT_DBG) & (1 << 17))              // experiment: ordinary (cached) loads and stores
            hipLaunchKernelGGL(adam_step_kernel<false>, dim3(batch.first[cnt]), dim3(ADAM_THREADS), 0, st, batch, cnt, slot0, lr,
                               lr_dev, beta1, beta2, eps, l2_value ? l2_ws : nullptr, clock, consts, l2_value, l2_total, tk);
        else
            hipLaunchKernelGGL(adam_step_kernel<true>, dim3(batch.first[cnt]), dim3(ADAM_THREADS), 0, st, batch, cnt, slot0, lr,
                               lr_dev, beta1, beta2, eps, l2_value ? l2_ws : nullptr, clock, consts, l2_value, l2_total, tk);
        slot0 += batch.first[cnt];
    }
    if (l2_value && !ticket) hipLaunchKernelGGL(adam_l2_finish_kernel, dim3(1), dim3(1024), 0, st, l2_ws, slot0, l2_value);
    return xdfm_check_launch("adam_step");
}

int xdfm_adam_catchup_rows(const float* X, long ldx, int B, const int* cols, const int* vocab, int m, int D,
                           const xdfm_adam_rows* emb, const xdfm_adam_rows* lin, const xdfm_adam_clock* clk,
                           double beta1, double beta2, double eps, float* backlog, void* stream) {
    XDFM_REQUIRE(X && cols && vocab && emb && clk && backlog, "adam_catchup_rows: null pointer");
    XDFM_REQUIRE(B > 0 && m > 0 && D > 0, "adam_catchup_rows: bad shape B=%d m=%d D=%d", B, m, D);
    XDFM_REQUIRE((((size_t)backlog) & 7) == 0, "adam_catchup_rows: backlog must be 8-byte aligned");
    const AdamRowsDev e = {emb->param, emb->exp_avg, emb->exp_avg_sq, emb->last, emb->l2, nullptr, nullptr};
    const AdamRowsDev l = lin ? AdamRowsDev{lin->param, lin->exp_avg, lin->exp_avg_sq, lin->last, lin->l2, nullptr, nullptr} : e;
    const int QT = (D + 3) / 4 + ((D & 3) ? 1 : 0) + (lin ? 1 : 0);
    const long threads = (long)B * m * QT;
    hipLaunchKernelGGL(adam_catchup_rows_kernel, dim3((unsigned)ceil_div(threads, (long)ADAM_THREADS)), dim3(ADAM_THREADS), 0,
                       (hipStream_t)stream, X, ldx, B, cols, vocab, m, D, e, l, lin ? 1 : 0, clk->clock, clk->consts, beta1, beta2,
                       eps, reinterpret_cast<unsigned long long*>(backlog));
    return xdfm_check_launch("adam_catchup_rows");
}

int xdfm_adam_apply_rows(const float* X, long ldx, int B, const int* cols, const int* vocab, int m, int D,
                         const xdfm_adam_rows* emb, const xdfm_adam_rows* lin, const xdfm_adam_clock* clk,
                         double beta1, double beta2, double eps, float* l2_cell, float* l2_value, void* stream) {
    XDFM_REQUIRE(X && cols && vocab && emb && clk && l2_cell, "adam_apply_rows: null pointer");
    XDFM_REQUIRE(emb->grad && emb->marks && (!lin || (lin->grad && lin->marks)), "adam_apply_rows: gradient / mark tables missing");
    XDFM_REQUIRE(B > 0 && m > 0 && D > 0, "adam_apply_rows: bad shape B=%d m=%d D=%d", B, m, D);
    XDFM_REQUIRE((((size_t)l2_cell) & 7) == 0, "adam_apply_rows: l2_cell must be 8-byte aligned");
    const AdamRowsDev e = {emb->param, emb->exp_avg, emb->exp_avg_sq, emb->last, emb->l2, emb->grad, emb->marks};
    const AdamRowsDev l = lin ? AdamRowsDev{lin->param, lin->exp_avg, lin->exp_avg_sq, lin->last, lin->l2, lin->grad, lin->marks} : e;
    const int QT = (D + 3) / 4 + ((D & 3) ? 1 : 0) + (lin ? 1 : 0);
    const long threads = (long)B * m * QT + 8L * m;
    hipStream_t st = (hipStream_t)stream;
    unsigned* ticket = nullptr;      // thousands of blocks: the finish stays a launch of its own (tickets on one address serialise)
    hipLaunchKernelGGL(adam_apply_rows_kernel, dim3((unsigned)ceil_div(threads, (long)ADAM_THREADS)), dim3(ADAM_THREADS), 0, st, X, ldx,
                       B, cols, vocab, m, D, e, l, lin ? 1 : 0, clk->clock, clk->consts, beta1, beta2, eps,
                       reinterpret_cast<unsigned long long*>(l2_cell), l2_value, ticket);
    if (!ticket)
        hipLaunchKernelGGL(adam_rows_finish_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned long long*>(l2_cell), l2_value);
    return xdfm_check_launch("adam_apply_rows");
}

int xdfm_adam_flush(const xdfm_adam_tensor* tensors, int T, const xdfm_adam_clock* clk, double beta1, double beta2,
                    double eps, float* backlog, void* stream) {
    XDFM_REQUIRE(tensors && clk && clk->clock && clk->consts && backlog, "adam_flush: null pointer");
    XDFM_REQUIRE((((size_t)backlog) & 7) == 0, "adam_flush: backlog must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    std::vector<int> order;
    for (int t = 0; t < T; ++t)
        if (tensors[t].flags & XDFM_ADAM_DEFERRED) {
            XDFM_REQUIRE(tensors[t].last && tensors[t].param && tensors[t].exp_avg && tensors[t].exp_avg_sq,
                         "adam_flush: tensor %d has a null pointer", t);
            order.push_back(t);
        }
    const int n = (int)order.size();
    for (int l0 = 0; l0 < n; l0 += ADAM_CHUNK) {
        AdamBatch batch;
        const int cnt = n - l0 < ADAM_CHUNK ? n - l0 : ADAM_CHUNK;
        for (int k = 0; k < ADAM_CHUNK; ++k) batch.t[k] = adam_dev(tensors[order[l0 + (k < cnt ? k : 0)]]);
        batch.first[0] = 0;
        for (int k = 0; k < ADAM_CHUNK; ++k) {
            long nb = k < cnt ? ceil_div(batch.t[k].numel, (long)ADAM_BLOCK_ELEMS) : 0;
            if (k < cnt && nb < 1) nb = 1;
            if (nb > 4096) nb = 4096;
            batch.first[k + 1] = batch.first[k] + (int)nb;
        }
        hipLaunchKernelGGL(adam_flush_kernel, dim3(batch.first[cnt]), dim3(ADAM_THREADS), 0, st, batch, cnt, clk->clock, clk->consts,
                           beta1, beta2, eps, reinterpret_cast<unsigned long long*>(backlog));
    }
    hipLaunchKernelGGL(adam_clock_reset_kernel, dim3(1), dim3(64), 0, st, clk->clock);
    return xdfm_check_launch("adam_flush");
}

int xdfm_adam_selftest(int mode, unsigned long long n, unsigned long long seed, double lr, double beta1, double beta2, double eps,
                       unsigned long long* out, void* stream) {
    XDFM_REQUIRE(out && mode >= 0 && mode <= 3 && n > 0 && n <= (1ull << 34), "adam_selftest: bad arguments");
    const unsigned long long per = 1ull << 30;           // a launch holds fewer than 2^32 threads
    for (unsigned long long base = 0; base < n; base += per) {
        const unsigned long long cnt = n - base < per ? n - base : per;
        hipLaunchKernelGGL(adam_selftest_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mode, base, n,
                           seed, lr, beta1, beta2, eps, out);
    }
    return xdfm_check_launch("adam_selftest");
}

}  // extern "C"
