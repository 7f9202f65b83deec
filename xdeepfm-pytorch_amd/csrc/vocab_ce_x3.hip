// Vocabulary-wide softmax cross-entropy of the SFG heads on the f16x3 matrix pipe, without the logits in HBM
// (deepctr/xdeepfm_pro/sfg_decoder.py:146-149: nn.Linear(K, V) per sparse field, :277-283: F.cross_entropy; K = 32 or 64).
//
//   ce[r] = logsumexp_v(h_r . W_v + b_v) - (h_r . W_t + b_t),  t = target[r]
//
// The tiled path (vocab_ce.hip + library GEMMs) writes each [rows, tile] block of logits and reads it two or three times;
// here a 32 x 32 block of logits exists only in the accumulators of one wave:
//
//   vx_hs_kernel<KT, 0>  "rows stationary": a workgroup of 8 waves keeps 16 row tiles (32 rows each) of the hidden layer as
//                        MFMA B operands in registers and streams the head's weights through LDS, 256 vocabulary rows at a
//                        time (each wave converts one 32-row block to fp16 hi / lo halves with the block's own power-of-two
//                        scale).  z^T = W_blk . H_tile^T has lane = example, so the online log-sum-exp (base 2) is lane
//                        local: one running (max, sum) pair per lane and tile.  Partials per vocabulary range, merged by
//                        vx_lse_merge_kernel together with the target logit.  Also leaves max|W| (the backward's scale).
//   vx_hs_kernel<KT, 1>  same skeleton, backward w.r.t. the hidden layer: P^T = g * softmax in the accumulator layout IS a
//                        B operand (k = vocabulary row in accumulator order), A = W^T fragments the producer wave scatters
//                        into LDS: dH^T += W_blk^T . P^T.  Partial slabs per vocabulary range, summed in a fixed order by
//                        vx_dh_merge_kernel (no atomics), which also subtracts g * W[target].
//   vx_ws_kernel<KT>     "weights stationary": a wave keeps its 32 vocabulary rows as B operand and the [K x 32] gradient
//                        of those rows as accumulators and walks ALL row tiles (fragments of H and H^T, packed once per
//                        step, through a double-buffered LDS ring): z = H_tile . W_blk^T (lane = vocabulary row, registers
//                        = examples), P as B operand again, dW^T += H_tile^T . P; db = column sums.  Every row of dW is
//                        written once, complete.
//
// Arithmetic: every product is hi*hi + hi*lo + lo*hi on fp16 halves of power-of-two scaled operands (as the CIN's
// f16x3 kernels), fp32 accumulation; exp / log in base 2 (v_exp_f32 / v_log_f32) on fp32.
#include "cin_x3_fwd.h"

#define VX_LOG2E 1.4426950408889634f
#define VX_LN2 0.6931471805599453f
#define VX_WAVES 8                      // waves of the rows-stationary kernels = 32-row weight blocks per LDS stage
#define VX_SB (32 * VX_WAVES)           // vocabulary rows per stage
#define VX_RG (64 * VX_WAVES)           // rows per row group: two tiles per wave
#define VX_NEG (-1.0e38f)

__device__ __forceinline__ float vx_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float vx_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// hi / lo fp16 halves of two fp32 values with the residual z - hi formed by one v_fma_mix_f32 per element (the fp16 half is
// read in place): 4 vector instructions per pair instead of 6 -- the vector issue port is what bounds these kernels
__device__ __forceinline__ void vx_split2(float z0, float z1, h2& hi, h2& lo) {
    const f2 z = {z0, z1};
    hi = __builtin_convertvector(z, h2);
    const unsigned hb = __builtin_bit_cast(unsigned, hi);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(z0), "v"(hb));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(z1), "v"(hb));
    const f2 r = {r0, r1};
    lo = __builtin_convertvector(r, h2);
}

// words[i * stride] = 0 (a kernel, not a memset node: captured graphs with memset nodes are not replayable on this stack)
__global__ void vx_zero_words_kernel(unsigned* __restrict__ w, int n, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[(long)i * stride] = 0u;
}

// out[y] = max(out[y], bits(max |x[y * stride + i]|, i < n)) -- non-negative floats order like their bit patterns
__global__ __launch_bounds__(256) void vx_absmax_kernel(const float* __restrict__ x, long n, long stride, unsigned* __restrict__ out, int out_stride) {
    const float* __restrict__ p = x + (long)blockIdx.y * stride;
    float mx = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) mx = fmaxf(mx, fabsf(p[i]));
    mx = vx_wave_max(mx);
    if ((threadIdx.x & 63) == 0) atomicMax(out + (long)blockIdx.y * out_stride, __float_as_uint(mx));
}

// One wave per row tile: fragments of H (lane (b, hh) holds H[32 t + b][16 s + 8 hh + 0..8): A operand of z = H W^T and B
// operand of z^T = W H^T) and of H^T (lane (kl, hh) holds H[32 t + frag_row(8 s' + j, hh)][32 kt + kl], j < 8: A operand of
// dW^T = H^T P with the examples in accumulator order), hi and lo halves, scaled by one power of two for the whole matrix.
template <int KT>
__global__ __launch_bounds__(64) void vx_pack_hidden_kernel(const float* __restrict__ H, long ldh, int R, const unsigned* __restrict__ hmax,
                                                            h8* __restrict__ Hf, h8* __restrict__ HTf, float* __restrict__ hscale) {
    constexpr int KS = 2 * KT;
    const int t = blockIdx.x, lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
    const float sH = x3_pow2_scale(__uint_as_float(*hmax), 12);
    if (t == 0 && lane == 0) { hscale[0] = sH; hscale[1] = 1.f / sH; }
    const int row = 32 * t + c;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = row < R ? H[(long)row * ldh + 16 * s + 8 * hh + e] * sH : 0.f;
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            h2 a, b;
            x3_split2(x[e], x[e + 1], a, b);
            hi[e] = a.x; hi[e + 1] = a.y; lo[e] = b.x; lo[e + 1] = b.y;
        }
        Hf[(((long)t * KS + s) * 2 + 0) * 64 + lane] = hi;
        Hf[(((long)t * KS + s) * 2 + 1) * 64 + lane] = lo;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 32 * t + frag_row(8 * sp + j, hh);
                x[j] = r < R ? H[(long)r * ldh + 32 * kt + c] * sH : 0.f;
            }
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                h2 a, b;
                x3_split2(x[e], x[e + 1], a, b);
                hi[e] = a.x; hi[e + 1] = a.y; lo[e] = b.x; lo[e + 1] = b.y;
            }
            HTf[((((long)t * KT + kt) * 2 + sp) * 2 + 0) * 64 + lane] = hi;
            HTf[((((long)t * KT + kt) * 2 + sp) * 2 + 1) * 64 + lane] = lo;
        }
}

// Upstream gradients g [F][R] -> gpack = [F][4] headers {sP, 1/sP, bits of max|g|, 0} then [F][Rpad] rows gs = g * sP (0 for
// the padding rows); sP = the power of two that puts the field's max|g| just below 2^15 (P = g * softmax <= max|g| in fp16).
__global__ __launch_bounds__(256) void vx_pack_g_kernel(const float* __restrict__ g, int R, int Rpad, int F, float* __restrict__ gpack) {
    const int f = blockIdx.y;
    float* __restrict__ hdr = gpack + 4 * f;
    const float sP = x3_pow2_scale(__uint_as_float(reinterpret_cast<const unsigned*>(hdr)[2]), 15);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { hdr[0] = sP; hdr[1] = 1.f / sP; }
    if (i < Rpad) gpack[4L * F + (long)f * Rpad + i] = i < R ? g[(long)f * R + i] * sP : 0.f;
}

typedef xdfm_vce_field VxField;
typedef xdfm_vce_item VxItem;

template <int KT, int MODE>
struct VxLds {
    static constexpr int KS = 2 * KT;
    static constexpr size_t a_bytes = (size_t)VX_WAVES * KS * 2 * 64 * sizeof(h8);
    static constexpr size_t wt_bytes = MODE == 1 ? (size_t)VX_WAVES * KT * 2 * 2 * 64 * sizeof(h8) : 0;
    static constexpr size_t bytes = a_bytes + wt_bytes + VX_WAVES * 32 * sizeof(float) + VX_WAVES * sizeof(float);
};

template <int KT, int MODE>
__global__ __launch_bounds__(64 * VX_WAVES) void vx_hs_kernel(
    const h8* __restrict__ Hf, const float* __restrict__ hscale, int ntiles, const VxField* __restrict__ fields,
    const VxItem* __restrict__ items, int F, unsigned* __restrict__ wmax_all, const float* __restrict__ lse2_all,
    const float* __restrict__ gpack, float* __restrict__ ws, long Rpad) {
    constexpr int KS = 2 * KT, K = 32 * KT;
    const VxItem item = items[blockIdx.x];                 // (field, vocabulary stages [sb0, sb1), index of the range)
    const VxField fd = fields[item.field];
    const float* __restrict__ W = fd.W;
    const float* __restrict__ bias = fd.bias;
    const int V = fd.V;
    unsigned* __restrict__ wmax = wmax_all + item.field;
    const float* __restrict__ lse2 = lse2_all + (long)item.field * Rpad;
    const float* __restrict__ gs = gpack + 4L * F + (long)item.field * Rpad;
    const float* __restrict__ gscale = gpack + 4 * item.field;
    float* __restrict__ out = ws + fd.ws_off;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h8* const La = reinterpret_cast<h8*>(smem);                                             // [wave][s][hi|lo][lane]
    _Float16* const Lwt = reinterpret_cast<_Float16*>(smem + VxLds<KT, MODE>::a_bytes);     // [wave][kt][s'][hi|lo][lane][8]
    float* const Lbias = reinterpret_cast<float*>(smem + VxLds<KT, MODE>::a_bytes + VxLds<KT, MODE>::wt_bytes);   // [wave][32], * log2(e)
    float* const Linv = Lbias + VX_WAVES * 32;                                              // [wave] 1 / block scale

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, hh = lane >> 5;
    const float invH = hscale[1];
    float sWg = 1.f;
    if constexpr (MODE == 1) sWg = x3_pow2_scale(__uint_as_float(*wmax), 12);

    // the wave's two row tiles: B operands for the whole kernel
    h8 bh[2][KS], bl[2][KS];
    int tile[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        tile[t] = blockIdx.y * (2 * VX_WAVES) + w + VX_WAVES * t;      // a group's first 8 tiles: one per wave
        const int ti = tile[t] < ntiles ? tile[t] : 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bh[t][s] = Hf[(((long)ti * KS + s) * 2 + 0) * 64 + lane];
            bl[t][s] = Hf[(((long)ti * KS + s) * 2 + 1) * 64 + lane];
            if (tile[t] >= ntiles) { bh[t][s] = (h8)(_Float16)0; bl[t][s] = (h8)(_Float16)0; }
        }
    }
    // a row group with at most 8 tiles (the tail of R just above a multiple of 512, or a small batch) has no second tile in
    // any wave: its stages run the one-tile instance of the consume loop (workgroup-uniform)
    const bool two = ntiles - (int)blockIdx.y * (2 * VX_WAVES) > VX_WAVES;
    float st_m[2] = {VX_NEG, VX_NEG}, st_s[2] = {0.f, 0.f};        // MODE 0: running base-2 maximum and sum per lane
    float r_lse[2] = {0.f, 0.f}, r_gs[2] = {0.f, 0.f};             // MODE 1: the lane's example
    f32x16 dacc[2][KT];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long row = 32L * tile[t] + c;
            if (tile[t] < ntiles) { r_lse[t] = lse2[row]; r_gs[t] = gs[row]; }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[t][kt][r] = 0.f;
        }
    }

    const int sb0 = item.sb0, sb1 = item.sb1;
    float4 raw[KS][2];
    float rawb = 0.f, wseen = 0.f;
    auto load_w = [&](int sb) {
        const long v = (long)sb * VX_SB + 32 * w + c;
        const long vc = v < V ? v : V - 1;
        const float4* __restrict__ src = reinterpret_cast<const float4*>(W + vc * K + 8 * hh);
#pragma unroll
        for (int s = 0; s < KS; ++s) { raw[s][0] = src[4 * s]; raw[s][1] = src[4 * s + 1]; }
        rawb = bias[vc];
    };
    auto publish = [&](int sb) {
        const long v = (long)sb * VX_SB + 32 * w + c;
        const bool live = v < V;
        float x[KS][8];
        float amax = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            x[s][0] = raw[s][0].x; x[s][1] = raw[s][0].y; x[s][2] = raw[s][0].z; x[s][3] = raw[s][0].w;
            x[s][4] = raw[s][1].x; x[s][5] = raw[s][1].y; x[s][6] = raw[s][1].z; x[s][7] = raw[s][1].w;
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[s][e] = live ? x[s][e] : 0.f; amax = fmaxf(amax, fabsf(x[s][e])); }
        }
        float sW = sWg;
        if constexpr (MODE == 0) {
            amax = vx_wave_max(amax);
            wseen = fmaxf(wseen, amax);
            sW = x3_pow2_scale(amax, 12);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                h2 a, b;
                x3_split2(x[s][e] * sW, x[s][e + 1] * sW, a, b);
                hi[e] = a.x; hi[e + 1] = a.y; lo[e] = b.x; lo[e + 1] = b.y;
            }
            La[((w * KS + s) * 2 + 0) * 64 + lane] = hi;
            La[((w * KS + s) * 2 + 1) * 64 + lane] = lo;
            if constexpr (MODE == 1) {
                // the same values as W^T fragments: element (v = c, kappa = 16 s + 8 hh + e) -> fragment (kt, s'), lane
                // (kappa & 31, hh'), slot j with frag_row(8 s' + j, hh') == c
                const int sp = c >> 4, hd = (c >> 2) & 1, j = (c & 3) + 4 * ((c >> 3) & 1);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int kappa = 16 * s + 8 * hh + e;
                    const int base = ((((w * KT + (kappa >> 5)) * 2 + sp) * 2) * 64 + (kappa & 31) + 32 * hd) * 8 + j;
                    Lwt[base] = hi[e];
                    Lwt[base + 64 * 8] = lo[e];
                }
            }
        }
        if (hh == 0) Lbias[w * 32 + c] = live ? rawb * VX_LOG2E : -INFINITY;
        if (lane == 0) Linv[w] = 1.f / sW;
    };

    if (sb0 < sb1) load_w(sb0);
    for (int sb = sb0; sb < sb1; ++sb) {
        publish(sb);
        __syncthreads();
        if (sb + 1 < sb1) load_w(sb + 1);               // in flight while this stage is consumed
        auto zmma = [&](auto ntc, int wb, f32x16 (&acc)[2], const float (&c0)[2]) {  // c0 + z^T of block wb against the wave's NT tiles
            constexpr int NT = decltype(ntc)::value;
            h8 ah[KS], al[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ah[s] = La[((wb * KS + s) * 2 + 0) * 64 + lane];
                al[s] = La[((wb * KS + s) * 2 + 1) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = c0[t];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = x3_mfma<3>(al[s], bh[t][s], acc[t]);
                    acc[t] = x3_mfma<3>(ah[s], bl[t][s], acc[t]);
                    acc[t] = x3_mfma<3>(ah[s], bh[t][s], acc[t]);
                }
        };
        auto bias_of = [&](int wb, float (&b2)[16]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(Lbias + wb * 32 + 8 * q + 4 * hh);
                b2[4 * q] = bq.x; b2[4 * q + 1] = bq.y; b2[4 * q + 2] = bq.z; b2[4 * q + 3] = bq.w;
            }
        };
        if constexpr (MODE == 0) {
            // A wave issues in order: behind a chain of dependent MFMAs its own VALU instructions wait, and the two waves
            // of a SIMD run the same phase after every barrier.  So the log-sum-exp of block wb is dealt, by hand, between
            // the MFMAs of block wb + 1: 12 slots of [one MFMA per tile, a slice of the epilogue], fenced so that hipcc
            // keeps the order.  The operands of block wb + 2 and the bias of block wb + 1 are read from LDS into the
            // registers the slots have just released.
            auto consume = [&](auto ntc) {
            constexpr int NT = decltype(ntc)::value;
            f32x16 acc[2][2];
            h8 ah[KS], al[KS];
            float b2[16], cw;
            auto load_frag = [&](int wb, int s) {
                ah[s] = La[((wb * KS + s) * 2 + 0) * 64 + lane];
                al[s] = La[((wb * KS + s) * 2 + 1) * 64 + lane];
            };
            auto load_bias = [&](int wb) { bias_of(wb, b2); cw = VX_LOG2E * invH * Linv[wb]; };
            auto mma = [&](f32x16 (&an)[2], int k) {              // slot k: term k % 3 of k-step k / 3, both tiles
                const int ks = k / 3, term = k % 3;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    an[t] = term == 0 ? x3_mfma<3>(al[ks], bh[t][ks], an[t])
                          : term == 1 ? x3_mfma<3>(ah[ks], bl[t][ks], an[t]) : x3_mfma<3>(ah[ks], bh[t][ks], an[t]);
            };
            float z[2][16], mx[2], mn[2], ex[2], sum[2];
            auto slice = [&](const f32x16 (&ac)[2], int k) {      // slice k of the epilogue of the block in `ac`
                if (k < 4) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (k == 0) mx[t] = VX_NEG;
#pragma unroll
                        for (int r = 4 * k; r < 4 * k + 4; ++r) { z[t][r] = fmaf(ac[t][r], cw, b2[r]); mx[t] = fmaxf(mx[t], z[t][r]); }
                        if (k == 3) { mn[t] = fmaxf(st_m[t], mx[t]); ex[t] = vx_exp2(st_m[t] - mn[t]); sum[t] = 0.f; }
                    }
                } else {
                    const int t = (k - 4) >> 2, r0 = 4 * ((k - 4) & 3);
                    if (t < NT) {
#pragma unroll
                        for (int r = r0; r < r0 + 4; ++r) sum[t] += vx_exp2(z[t][r] - mn[t]);
                    }
                    if (k == 11) {
#pragma unroll
                        for (int u = 0; u < NT; ++u) { st_s[u] = fmaf(st_s[u], ex[u], sum[u]); st_m[u] = mn[u]; }
                    }
                }
            };
            auto zero = [&](f32x16 (&a)[2]) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[t][r] = 0.f;
            };
            // block 0: nothing to run under its MFMAs
#pragma unroll
            for (int q = 0; q < KS; ++q) load_frag(0, q);
            zero(acc[0]);
#pragma unroll
            for (int k = 0; k < 3 * KS; ++k) {
                mma(acc[0], k);
                if (k % 3 == 2) load_frag(1, k / 3);
            }
            load_bias(0);
            __builtin_amdgcn_sched_barrier(0);
            auto step = [&](f32x16 (&ac)[2], f32x16 (&an)[2], int wb) {      // MFMAs of block wb + 1 over the epilogue of block wb
                const int nn = wb + 2 < VX_WAVES ? wb + 2 : VX_WAVES - 1;    // operands to fetch next (clamped: the last fetch is idle)
                zero(an);
#pragma unroll
                for (int k = 0; k < 3 * KS; ++k) {
                    mma(an, k);
                    if (k < 12) slice(ac, k);
                    if (k % 3 == 2) load_frag(nn, k / 3);
                    if (k == 3) load_bias(wb + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (3 * KS < 12) {
#pragma unroll
                    for (int k = 3 * KS; k < 12; ++k) slice(ac, k);
                }
            };
#pragma unroll 1
            for (int wb = 0; wb < VX_WAVES - 2; wb += 2) {
                step(acc[0], acc[1], wb);
                step(acc[1], acc[0], wb + 1);
            }
            step(acc[0], acc[1], VX_WAVES - 2);
#pragma unroll
            for (int k = 0; k < 12; ++k) slice(acc[1], k);        // the stage's last block: its bias was fetched in the step above
            };
            if (two) consume(std::integral_constant<int, 2>{}); else consume(std::integral_constant<int, 1>{});
        } else {
            auto consume = [&](auto ntc) {
            constexpr int NT = decltype(ntc)::value;
            const float icw = sWg / (VX_LOG2E * invH);           // 1 / cw (one scale for the whole head in this mode)
            const float c0[2] = {-r_lse[0] * icw, -r_lse[1] * icw};
#pragma unroll 1
            for (int wb = 0; wb < VX_WAVES; ++wb) {
                const float cw = VX_LOG2E * invH * Linv[wb];
                float b2[16];
                bias_of(wb, b2);
                f32x16 acc[2];
                zmma(ntc, wb, acc, c0);                 // accumulators start at -lse / cw: cw * acc + bias is logit - lse (base 2)
                h8 ph[2][2], pl[2][2];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const int r = 8 * sp + e;
                            const float p0 = vx_exp2(fmaf(acc[t][r], cw, b2[r])) * r_gs[t];
                            const float p1 = vx_exp2(fmaf(acc[t][r + 1], cw, b2[r + 1])) * r_gs[t];
                            h2 a, b;
                            vx_split2(p0, p1, a, b);
                            ph[t][sp][e] = a.x; ph[t][sp][e + 1] = a.y; pl[t][sp][e] = b.x; pl[t][sp][e + 1] = b.y;
                        }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp) {
                        const h8* f = reinterpret_cast<const h8*>(Lwt) + (((wb * KT + kt) * 2 + sp) * 2) * 64 + lane;
                        const h8 wh = f[0], wl = f[64];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            dacc[t][kt] = x3_mfma<3>(wl, ph[t][sp], dacc[t][kt]);
                            dacc[t][kt] = x3_mfma<3>(wh, pl[t][sp], dacc[t][kt]);
                            dacc[t][kt] = x3_mfma<3>(wh, ph[t][sp], dacc[t][kt]);
                        }
                    }
            }
            };
            if (two) consume(std::integral_constant<int, 2>{}); else consume(std::integral_constant<int, 1>{});
        }
        __syncthreads();
    }

    if constexpr (MODE == 0) {
        wseen = vx_wave_max(wseen);
        if (lane == 0 && blockIdx.y == 0) atomicMax(wmax, __float_as_uint(wseen));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float mo = __shfl_xor(st_m[t], 32), so = __shfl_xor(st_s[t], 32);
            const float m0 = hh == 0 ? st_m[t] : mo, s0 = hh == 0 ? st_s[t] : so;      // lower half first, in both lanes
            const float m1 = hh == 0 ? mo : st_m[t], s1 = hh == 0 ? so : st_s[t];
            const float M = fmaxf(m0, m1);
            const float S = fmaf(s0, vx_exp2(m0 - M), s1 * vx_exp2(m1 - M));
            if (hh == 0 && tile[t] < ntiles)
                reinterpret_cast<float2*>(out)[(long)item.range * Rpad + 32L * tile[t] + c] = make_float2(M, S);
        }
    } else {
        const float inv = gscale[1] / sWg;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (tile[t] >= ntiles) continue;
            float* __restrict__ dst = out + ((long)item.range * Rpad + 32L * tile[t] + c) * K;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(dst + 32 * kt + 8 * q + 4 * hh) =
                        make_float4(dacc[t][kt][4 * q] * inv, dacc[t][kt][4 * q + 1] * inv, dacc[t][kt][4 * q + 2] * inv,
                                    dacc[t][kt][4 * q + 3] * inv);
        }
    }
}

// One thread per (row, field): merge the ranges' (max, sum) pairs in range order, take the target logit with plain fp32
// fmas, write ce = lse - z_target and the base-2 log-sum-exp the backward reuses.
__global__ __launch_bounds__(256) void vx_lse_merge_kernel(const float* __restrict__ ws, const VxField* __restrict__ fields, long Rpad, int R,
                                                           const float* __restrict__ H, long ldh, const long* __restrict__ tgt_all, int K,
                                                           float* __restrict__ ce, float* __restrict__ lse2) {
    const int r = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (r >= R) return;
    const VxField fd = fields[f];
    const float2* __restrict__ part = reinterpret_cast<const float2*>(ws + fd.ws_off);
    float M = VX_NEG;
    for (int i = 0; i < fd.vr; ++i) M = fmaxf(M, part[(long)i * Rpad + r].x);
    float S = 0.f;
    for (int i = 0; i < fd.vr; ++i) { const float2 p = part[(long)i * Rpad + r]; S = fmaf(p.y, vx_exp2(p.x - M), S); }
    const float l2 = M + __builtin_amdgcn_logf(S);             // v_log_f32: log2
    lse2[(long)f * Rpad + r] = l2;
    long t = tgt_all[(long)f * R + r];
    t = t < 0 ? 0 : (t >= fd.V ? fd.V - 1 : t);
    const float* __restrict__ wr = fd.W + t * K;
    const float* __restrict__ hr = H + (long)r * ldh;
    float z = 0.f;
    for (int k = 0; k < K; ++k) z = fmaf(hr[k], wr[k], z);
    ce[(long)f * R + r] = l2 * VX_LN2 - (z + fd.bias[t]);
}

// dh[r][k..k+4) = sum over fields (in order) of [sum over the field's ranges (in order) of its slabs - g[f][r] * W_f[target]]
__global__ __launch_bounds__(256) void vx_dh_merge_kernel(const float* __restrict__ ws, const VxField* __restrict__ fields, int F, long Rpad,
                                                          int R, int K, const float* __restrict__ g, const long* __restrict__ tgt_all,
                                                          float* __restrict__ dh, long lddh) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int k4 = K / 4;
    if (i >= (long)R * k4) return;
    const int r = (int)(i / k4), k = 4 * (int)(i % k4);
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int f = 0; f < F; ++f) {
        const VxField fd = fields[f];
        const float* __restrict__ slab = ws + fd.ws_off;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int v = 0; v < fd.vr; ++v) {
            const float4 p = *reinterpret_cast<const float4*>(slab + ((long)v * Rpad + r) * K + k);
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        long t = tgt_all[(long)f * R + r];
        t = t < 0 ? 0 : (t >= fd.V ? fd.V - 1 : t);
        const float4 wv = *reinterpret_cast<const float4*>(fd.W + t * K + k);
        const float gr = g[(long)f * R + r];
        tot.x += s.x - gr * wv.x; tot.y += s.y - gr * wv.y; tot.z += s.z - gr * wv.z; tot.w += s.w - gr * wv.w;
    }
    *reinterpret_cast<float4*>(dh + (long)r * lddh + k) = tot;
}

#define VX_WS_WAVES 8
#define VX_WS_TP 2                       // row tiles per ring slot: one barrier and one fetch / park round per slot
template <int KT>
struct VxSlotLds {                       // VX_WS_TP row tiles as the weights-stationary kernel consumes them
    static constexpr int KS = 2 * KT;
    h8 hf[VX_WS_TP][KS * 2 * 64];
    h8 htf[VX_WS_TP][KT * 2 * 2 * 64];
    float lse[VX_WS_TP][32], gs[VX_WS_TP][32];
    int tgt[VX_WS_TP][32];
};

// Measured with parts of the kernel switched off (tools/vocab_ce_dbg.py): its phases ADD -- MFMAs 1.0 ms, exponentials and
// splits 0.7, tile ring and barriers 0.75, the rest 0.6 of 2.95 -- two waves per SIMD hide nothing of each other.  Hence 8
// waves share one ring whose slots hold two tiles: half the barriers, fetches and LDS writes per tile.
template <int KT>
__global__ __launch_bounds__(64 * VX_WS_WAVES, 1) void vx_ws_kernel(
    const h8* __restrict__ Hf, const h8* __restrict__ HTf, const float* __restrict__ hscale, int ntiles,
    const VxField* __restrict__ fields, int F, int nblk_all, const unsigned* __restrict__ wmax_all, const float* __restrict__ lse2_all,
    const float* __restrict__ gpack, const long* __restrict__ tgt_all, int R, long Rpad, int dbg) {
    // dbg (option "dbg", bits 20..23; timing experiments, results become wrong): 1 = no exponentials / target test / split,
    // 2 = no dW MFMAs, 4 = no z MFMAs, 8 = no tile ring traffic and no barriers
    constexpr int KS = 2 * KT, K = 32 * KT;
    constexpr int NHF = KS * 2 * 64, NHT = KT * 2 * 2 * 64, NTL = NHF + NHT;      // h8 per tile
    constexpr int NTH = 64 * VX_WS_WAVES;
    constexpr int PER = VX_WS_TP * NTL / NTH;                                     // h8 per thread and slot
    static_assert(VX_WS_TP * NTL == NTH * PER && NTL % 64 == 0, "a slot is dealt evenly to the threads");
    extern __shared__ __attribute__((aligned(16))) char ws_smem[];
    VxSlotLds<KT>* const L = reinterpret_cast<VxSlotLds<KT>*>(ws_smem);            // ring of 3 slots
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, hh = lane >> 5;
    const int nslots = (ntiles + VX_WS_TP - 1) / VX_WS_TP;

    for (int gb = blockIdx.x; gb < nblk_all; gb += gridDim.x) {
        int f = 0;
        while (f + 1 < F && fields[f + 1].blk0 <= gb) ++f;              // the field this block of 256 vocabulary rows belongs to
        const VxField fd = fields[f];
        const float* __restrict__ W = fd.W;
        const float* __restrict__ bias = fd.bias;
        float* __restrict__ dW = fd.dW;
        float* __restrict__ db = fd.db;
        const int V = fd.V, blk = gb - fd.blk0;
        const float* __restrict__ lse2 = lse2_all + (long)f * Rpad;
        const float* __restrict__ gs = gpack + 4L * F + (long)f * Rpad;
        const long* __restrict__ tgt = tgt_all + (long)f * R;
        const float sW = x3_pow2_scale(__uint_as_float(wmax_all[f]), 12);
        const float cw = VX_LOG2E * hscale[1] / sW;
        const float icw = -1.f / cw;                           // the tiles' lse values are parked as -lse / cw: z starts there
        const float inv_b = gpack[4 * f + 1], inv_w = inv_b * hscale[1];
        const long v = (long)blk * (32 * VX_WS_WAVES) + 32 * w + c;
        const bool live = v < V;
        const long vc = live ? v : V - 1;
        h8 bh[KS], bl[KS];
        {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(W + vc * K + 8 * hh);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 a = src[4 * s], b = src[4 * s + 1];
                const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    h2 p, q;
                    x3_split2(live ? x[e] * sW : 0.f, live ? x[e + 1] * sW : 0.f, p, q);
                    bh[s][e] = p.x; bh[s][e + 1] = p.y; bl[s][e] = q.x; bl[s][e + 1] = q.y;
                }
            }
        }
        const float b2 = live ? bias[vc] * VX_LOG2E : -INFINITY;
        const int vi = live ? (int)v : -2;
        f32x16 acc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
        float dbacc = 0.f;

        h8 stage[PER];
        unsigned stage_l = 0u;                              // bits of -lse2 / cw or gs, or a target id
        auto fetch = [&](int sl) {                          // slot sl = tiles VX_WS_TP * sl ..; a tile past the last one repeats it with
#pragma unroll                                              // zero gradients (gs is zero beyond R, the targets are -1)
            for (int i = 0; i < PER; ++i) {
                const int e = threadIdx.x + i * NTH;
                const int u = e / NTL, r = e - u * NTL;
                const int t = min(VX_WS_TP * sl + u, ntiles - 1);
                stage[i] = r < NHF ? Hf[(long)t * NHF + r] : HTf[(long)t * NHT + (r - NHF)];
            }
            const int q = threadIdx.x >> 6, i = threadIdx.x & 63;          // q: 0 lse, 1 gs, 2 targets; i: row within the slot
            const long row = 32L * VX_WS_TP * sl + i;
            if (q == 0) stage_l = __float_as_uint(lse2[row] * icw);
            else if (q == 1) stage_l = __float_as_uint(gs[row]);
            else if (q == 2) {
                const long tv = row < R ? tgt[row] : -1;
                stage_l = (unsigned)(row < R ? (int)(tv < 0 ? 0 : (tv >= V ? V - 1 : tv)) : -1);
            }
        };
        auto park = [&](int buf) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int e = threadIdx.x + i * NTH;
                const int u = e / NTL, r = e - u * NTL;
                if (r < NHF) L[buf].hf[u][r] = stage[i]; else L[buf].htf[u][r - NHF] = stage[i];
            }
            const int q = threadIdx.x >> 6, i = threadIdx.x & 63;
            if (q == 0) L[buf].lse[i >> 5][i & 31] = __uint_as_float(stage_l);
            else if (q == 1) L[buf].gs[i >> 5][i & 31] = __uint_as_float(stage_l);
            else if (q == 2) L[buf].tgt[i >> 5][i & 31] = (int)stage_l;
        };
        __syncthreads();                                   // the previous block's last slot is consumed
        fetch(0);
        park(0);
        if (nslots > 1) { fetch(1); park(1); }
        __syncthreads();
        auto zmma = [&](int buf, int u, f32x16& z) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {                       // accumulators start at -lse[row] / cw
                const float4 lq = *reinterpret_cast<const float4*>(L[buf].lse[u] + 8 * q + 4 * hh);
                z[4 * q] = lq.x; z[4 * q + 1] = lq.y; z[4 * q + 2] = lq.z; z[4 * q + 3] = lq.w;
            }
            if (dbg & 4) return;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const h8 ah = L[buf].hf[u][(s * 2 + 0) * 64 + lane], al = L[buf].hf[u][(s * 2 + 1) * 64 + lane];
                z = x3_mfma<3>(al, bh[s], z);
                z = x3_mfma<3>(ah, bl[s], z);
                z = x3_mfma<3>(ah, bh[s], z);
            }
        };
        // one tile: P = g (softmax - onehot(target)) from its z (formed while the previous tile was here), then dW^T += H^T P
        auto pstep = [&](int buf, int u, const f32x16& z) {
            h8 ph[2], pl[2];
            if (dbg & 1) {
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ph[sp][e] = (_Float16)z[8 * sp + e]; pl[sp][e] = (_Float16)0; }
            } else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 gq = *reinterpret_cast<const float4*>(L[buf].gs[u] + 8 * q + 4 * hh);
                const int4 tq = *reinterpret_cast<const int4*>(L[buf].tgt[u] + 8 * q + 4 * hh);
                // the row's own target entry is subtracted here, so dW / db are complete when they are stored
                const float p0 = (vx_exp2(fmaf(z[4 * q], cw, b2)) - (tq.x == vi ? 1.f : 0.f)) * gq.x;
                const float p1 = (vx_exp2(fmaf(z[4 * q + 1], cw, b2)) - (tq.y == vi ? 1.f : 0.f)) * gq.y;
                const float p2 = (vx_exp2(fmaf(z[4 * q + 2], cw, b2)) - (tq.z == vi ? 1.f : 0.f)) * gq.z;
                const float p3 = (vx_exp2(fmaf(z[4 * q + 3], cw, b2)) - (tq.w == vi ? 1.f : 0.f)) * gq.w;
                dbacc += (p0 + p1) + (p2 + p3);
                h2 a, b;
                const int sp = q >> 1, e = 4 * (q & 1);
                vx_split2(p0, p1, a, b);
                ph[sp][e] = a.x; ph[sp][e + 1] = a.y; pl[sp][e] = b.x; pl[sp][e + 1] = b.y;
                vx_split2(p2, p3, a, b);
                ph[sp][e + 2] = a.x; ph[sp][e + 3] = a.y; pl[sp][e + 2] = b.x; pl[sp][e + 3] = b.y;
            }
            if (!(dbg & 2)) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp) {
                        const h8 th = L[buf].htf[u][((kt * 2 + sp) * 2 + 0) * 64 + lane], tl = L[buf].htf[u][((kt * 2 + sp) * 2 + 1) * 64 + lane];
                        acc[kt] = x3_mfma<3>(tl, ph[sp], acc[kt]);
                        acc[kt] = x3_mfma<3>(th, pl[sp], acc[kt]);
                        acc[kt] = x3_mfma<3>(th, ph[sp], acc[kt]);
                    }
            } else { acc[0][0] += (float)ph[0][0] + (float)ph[1][3] + (float)pl[0][1] + (float)pl[1][5]; }
        };
        f32x16 za, zb;
        zmma(0, 0, za);
        int buf = 0;
        for (int sl = 0; sl < nslots; ++sl) {
            const int bufn = buf == 2 ? 0 : buf + 1, bufa = bufn == 2 ? 0 : bufn + 1;
            if (sl + 2 < nslots && !(dbg & 8)) fetch(sl + 2);
            zmma(buf, 1, zb);                                  // the next tile's z MFMAs are issued before this tile's exponentials
            pstep(buf, 0, za);
            if (sl + 1 < nslots) zmma(bufn, 0, za);
            pstep(buf, 1, zb);
            if (!(dbg & 8)) {
                if (sl + 2 < nslots) park(bufa);               // the slot of two rounds ago: every wave left it at the last barrier
                __syncthreads();
            }
            buf = bufn;
        }
        const float dbo = dbacc + __shfl_xor(dbacc, 32);
        if (live) {
            if (hh == 0 && db) db[v] = dbo * inv_b;
            if (dW) {
                float* __restrict__ dst = dW + v * K;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(dst + 32 * kt + 8 * q + 4 * hh) =
                            make_float4(acc[kt][4 * q] * inv_w, acc[kt][4 * q + 1] * inv_w, acc[kt][4 * q + 2] * inv_w, acc[kt][4 * q + 3] * inv_w);
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static inline long vx_rows_padded(int R) { return (long)ceil_div(R, VX_RG) * VX_RG; }
static inline long vx_pack_floats(int R, int K) {           // [16 header floats: sH, 1/sH, -, -, bits of max|H|, ...][Hf][HTf]
    return 16 + (long)ceil_div(R, 32) * K * 64;             // per tile: K*32 values x (hi, lo) x two packings, 2 bytes each
}
static inline bool vx_k_ok(int K) { return K == 32 || K == 64; }

extern "C" {

int xdfm_vocab_ce_x3_supported(int K) { return vx_k_ok(K) && x3_terms() == 3 ? 1 : 0; }

long xdfm_vocab_ce_pack_elems(int R, int K) { return R > 0 && vx_k_ok(K) ? vx_pack_floats(R, K) : 0; }

long xdfm_vocab_ce_rows_padded(int R) { return R > 0 ? vx_rows_padded(R) : 0; }

long xdfm_vocab_ce_plan(int F, const int* V, int R, int K, xdfm_vce_field* fields, xdfm_vce_item* items, long max_items,
                        long* ws_elems, int* n_blk) {
    if (F <= 0 || !V || R <= 0 || !vx_k_ok(K)) { xdfm_fail(XDFM_ERR_INVALID, "vocab_ce_plan: bad arguments F=%d R=%d K=%d", F, R, K); return -1; }
    const int rg = ceil_div(R, VX_RG);
    const long Rpad = vx_rows_padded(R);
    long stages = 0;
    for (int f = 0; f < F; ++f) {
        if (V[f] <= 0) { xdfm_fail(XDFM_ERR_INVALID, "vocab_ce_plan: V[%d]=%d", f, V[f]); return -1; }
        stages += ceil_div(V[f], VX_SB);
    }
    // one workgroup (64-128 KB of LDS) per CU at a time: about four rounds of 256 over all fields and row groups
    long spr = ceil_div(stages * rg, 1024);
    if (spr < 1) spr = 1;
    for (;; ++spr) {                            // the per-field round-up must not spill into a fifth, nearly empty round
        long wgs = 0;
        int longest = 0;
        for (int f = 0; f < F; ++f) {
            const int nsb = ceil_div(V[f], VX_SB);
            wgs += (long)ceil_div(nsb, spr) * rg;
            longest = nsb > longest ? nsb : longest;
        }
        if (wgs <= 1024 || spr >= longest) break;
    }
    long n = 0, ws = 0;
    int blk = 0;
    for (int f = 0; f < F; ++f) {
        const int nsb = ceil_div(V[f], VX_SB);
        const int vr = ceil_div(nsb, spr);
        if (fields) {
            fields[f].V = V[f]; fields[f].vr = vr; fields[f].item0 = (int)n; fields[f].blk0 = blk; fields[f].ws_off = ws;
        }
        for (int i = 0; i < vr; ++i, ++n)
            if (items && n < max_items) {
                items[n].field = f; items[n].sb0 = (int)(i * spr); items[n].sb1 = (int)((i + 1) * spr < nsb ? (i + 1) * spr : nsb);
                items[n].range = i;
            }
        ws += (long)vr * Rpad * K;
        blk += ceil_div(V[f], 32 * VX_WS_WAVES);
    }
    if (ws_elems) *ws_elems = ws;
    if (n_blk) *n_blk = blk;
    return n;
}

int xdfm_vocab_ce_pack_hidden(const float* H, long ldh, int R, int K, float* pack, void* stream) {
    XDFM_REQUIRE(H && pack, "vocab_ce_pack_hidden: null pointer");
    XDFM_REQUIRE(R > 0 && vx_k_ok(K) && ldh == K, "vocab_ce_pack_hidden: bad shape R=%d K=%d ld=%ld (rows must be contiguous)", R, K, ldh);
    hipStream_t st = (hipStream_t)stream;
    unsigned* hmax = reinterpret_cast<unsigned*>(pack) + 4;
    hipLaunchKernelGGL(vx_zero_words_kernel, dim3(1), dim3(64), 0, st, hmax, 1, 1);
    const int tiles = ceil_div(R, 32);
    int gx = ceil_div((long)R * K, 256 * 8);
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    hipLaunchKernelGGL(vx_absmax_kernel, dim3(gx, 1), dim3(256), 0, st, H, (long)R * K, 0L, hmax, 0);
    h8* Hf = reinterpret_cast<h8*>(pack + 16);
    h8* HTf = Hf + (long)tiles * (K / 16) * 2 * 64;
    if (K == 64) hipLaunchKernelGGL((vx_pack_hidden_kernel<2>), dim3(tiles), dim3(64), 0, st, H, ldh, R, hmax, Hf, HTf, pack);
    else hipLaunchKernelGGL((vx_pack_hidden_kernel<1>), dim3(tiles), dim3(64), 0, st, H, ldh, R, hmax, Hf, HTf, pack);
    return xdfm_check_launch("vocab_ce_pack_hidden");
}

int xdfm_vocab_ce_fwd(const float* pack, const float* H, long ldh, int R, int K, const xdfm_vce_field* fields, int F,
                      const xdfm_vce_item* items, long n_items, const long* targets, float* ws, float* ce, float* lse2,
                      unsigned* wmax, void* stream) {
    XDFM_REQUIRE(pack && H && fields && items && targets && ws && ce && lse2 && wmax, "vocab_ce_fwd: null pointer");
    XDFM_REQUIRE(R > 0 && F > 0 && n_items > 0 && vx_k_ok(K) && ldh >= K, "vocab_ce_fwd: bad shape R=%d K=%d F=%d", R, K, F);
    hipStream_t st = (hipStream_t)stream;
    const long Rpad = vx_rows_padded(R);
    const int ntiles = ceil_div(R, 32);
    hipLaunchKernelGGL(vx_zero_words_kernel, dim3(ceil_div(F, 64)), dim3(64), 0, st, wmax, F, 1);
    const h8* Hf = reinterpret_cast<const h8*>(pack + 16);
    const dim3 grid((unsigned)n_items, ceil_div(R, VX_RG)), block(64 * VX_WAVES);
    if (K == 64)
        hipLaunchKernelGGL((vx_hs_kernel<2, 0>), grid, block, (VxLds<2, 0>::bytes), st, Hf, pack, ntiles, fields, items, F, wmax,
                           (const float*)nullptr, (const float*)nullptr, ws, Rpad);
    else
        hipLaunchKernelGGL((vx_hs_kernel<1, 0>), grid, block, (VxLds<1, 0>::bytes), st, Hf, pack, ntiles, fields, items, F, wmax,
                           (const float*)nullptr, (const float*)nullptr, ws, Rpad);
    hipLaunchKernelGGL(vx_lse_merge_kernel, dim3(ceil_div(R, 256), F), dim3(256), 0, st, ws, fields, Rpad, R, H, ldh, targets, K, ce, lse2);
    return xdfm_check_launch("vocab_ce_fwd");
}

int xdfm_vocab_ce_pack_g(const float* g, int F, int R, float* gpack, void* stream) {
    XDFM_REQUIRE(g && gpack && R > 0 && F > 0, "vocab_ce_pack_g: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long Rpad = vx_rows_padded(R);
    hipLaunchKernelGGL(vx_zero_words_kernel, dim3(ceil_div(F, 64)), dim3(64), 0, st, reinterpret_cast<unsigned*>(gpack) + 2, F, 4);
    int gx = ceil_div(R, 2048);
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(vx_absmax_kernel, dim3(gx, F), dim3(256), 0, st, g, (long)R, (long)R, reinterpret_cast<unsigned*>(gpack) + 2, 4);
    hipLaunchKernelGGL(vx_pack_g_kernel, dim3(ceil_div(Rpad, 256), F), dim3(256), 0, st, g, R, (int)Rpad, F, gpack);
    return xdfm_check_launch("vocab_ce_pack_g");
}

int xdfm_vocab_ce_bwd_h(const float* pack, int R, int K, const xdfm_vce_field* fields, int F, const xdfm_vce_item* items, long n_items,
                        const long* targets, const float* g, const float* gpack, const float* lse2, unsigned* wmax, float* ws,
                        float* dh, long lddh, void* stream) {
    XDFM_REQUIRE(pack && fields && items && targets && g && gpack && lse2 && wmax && ws && dh, "vocab_ce_bwd_h: null pointer");
    XDFM_REQUIRE(R > 0 && F > 0 && n_items > 0 && vx_k_ok(K) && lddh >= K && lddh % 4 == 0, "vocab_ce_bwd_h: bad shape R=%d K=%d F=%d", R, K, F);
    hipStream_t st = (hipStream_t)stream;
    const long Rpad = vx_rows_padded(R);
    const int ntiles = ceil_div(R, 32);
    const h8* Hf = reinterpret_cast<const h8*>(pack + 16);
    const dim3 grid((unsigned)n_items, ceil_div(R, VX_RG)), block(64 * VX_WAVES);
    if (K == 64)
        hipLaunchKernelGGL((vx_hs_kernel<2, 1>), grid, block, (VxLds<2, 1>::bytes), st, Hf, pack, ntiles, fields, items, F, wmax, lse2, gpack,
                           ws, Rpad);
    else
        hipLaunchKernelGGL((vx_hs_kernel<1, 1>), grid, block, (VxLds<1, 1>::bytes), st, Hf, pack, ntiles, fields, items, F, wmax, lse2, gpack,
                           ws, Rpad);
    hipLaunchKernelGGL(vx_dh_merge_kernel, dim3(ceil_div((long)R * (K / 4), 256)), dim3(256), 0, st, ws, fields, F, Rpad, R, K, g, targets,
                       dh, lddh);
    return xdfm_check_launch("vocab_ce_bwd_h");
}

int xdfm_vocab_ce_bwd_w(const float* pack, int R, int K, const xdfm_vce_field* fields, int F, int n_blk, const long* targets,
                        const float* gpack, const float* lse2, const unsigned* wmax, void* stream) {
    XDFM_REQUIRE(pack && fields && targets && gpack && lse2 && wmax, "vocab_ce_bwd_w: null pointer");
    XDFM_REQUIRE(R > 0 && F > 0 && n_blk > 0 && vx_k_ok(K), "vocab_ce_bwd_w: bad shape R=%d K=%d F=%d", R, K, F);
    hipStream_t st = (hipStream_t)stream;
    const int tiles = ceil_div(R, 32);
    const long Rpad = vx_rows_padded(R);
    const h8* Hf = reinterpret_cast<const h8*>(pack + 16);
    const h8* HTf = Hf + (long)tiles * (K / 16) * 2 * 64;
    const int grid = n_blk < 1024 ? n_blk : 1024;              // one workgroup of 8 waves per CU at a time (100 KB of LDS)
    const int dbg = (xdfm_opt(OPT_DBG) >> 20) & 15;
    if (K == 64)
        hipLaunchKernelGGL((vx_ws_kernel<2>), dim3(grid), dim3(64 * VX_WS_WAVES), 3 * sizeof(VxSlotLds<2>), st, Hf, HTf, pack, tiles, fields, F,
                           n_blk, wmax, lse2, gpack, targets, R, Rpad, dbg);
    else
        hipLaunchKernelGGL((vx_ws_kernel<1>), dim3(grid), dim3(64 * VX_WS_WAVES), 3 * sizeof(VxSlotLds<1>), st, Hf, HTf, pack, tiles, fields, F,
                           n_blk, wmax, lse2, gpack, targets, R, Rpad, dbg);
    return xdfm_check_launch("vocab_ce_bwd_w");
}

}  // extern "C"
