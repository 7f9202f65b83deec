// K5: attention pooling over the CIN feature maps, one workgroup per example, one thread per token.
//
// replaces deepctr/layers/cin_attention.py:63-97 (MultiHeadSelfAttention), :130-144
// (AttentionPooling) and the tails of CINAttention.forward (:302-316: MHSA -> +residual ->
// LayerNorm -> pooling -> output_proj) and CINAttentionV2.forward (:452-464: N x (MHSA, residual,
// LayerNorm) -> pooling).  The reference materialises the [B, heads, S, S] score tensor (4.3-6.7 GB
// at B=4096, S=256-320) forward and again for backward; here the S tokens of an example live in
// LDS / registers and scores exist only as scalars.
//
// fp32 VALU on purpose: with head_dim 2-8 the contractions are tiny and the fp32 MFMA rate equals the
// fp32 VALU rate on gfx950, so matrix cores would buy nothing; the cost is S^2*(3D FMA + heads exp).
//
// theta (all parameters, packed by the host): per layer  Wq Wk Wv Wo [D][D] each, then (use_ln)
// gamma[D] beta[D];  after the layers  W1 [D][D], b1 [D], w2 [D].  The kernel returns the pooled
// vector [B][D]; CINAttention's output_proj (D -> featuremap_num, cin_attention.py:316) is a plain
// [B,D]x[D,fm] GEMM left to hipBLASLt like the model's other one-row heads.
#include "xdfm_internal.h"

template <int D>
struct RowVec {   // widest aligned vector that divides a row of D floats
    static constexpr int W = (D % 4 == 0) ? 4 : ((D % 2 == 0) ? 2 : 1);
};

template <int D>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&r)[D]) {
    constexpr int W = RowVec<D>::W;
    if constexpr (W == 4) {
#pragma unroll
        for (int c = 0; c < D / 4; ++c) {
            const float4 v = reinterpret_cast<const float4*>(p)[c];
            r[4 * c] = v.x; r[4 * c + 1] = v.y; r[4 * c + 2] = v.z; r[4 * c + 3] = v.w;
        }
    } else if constexpr (W == 2) {
#pragma unroll
        for (int c = 0; c < D / 2; ++c) {
            const float2 v = reinterpret_cast<const float2*>(p)[c];
            r[2 * c] = v.x; r[2 * c + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int c = 0; c < D; ++c) r[c] = p[c];
    }
}

template <int D>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&r)[D]) {
    constexpr int W = RowVec<D>::W;
    if constexpr (W == 4) {
#pragma unroll
        for (int c = 0; c < D / 4; ++c)
            reinterpret_cast<float4*>(p)[c] = make_float4(r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]);
    } else if constexpr (W == 2) {
#pragma unroll
        for (int c = 0; c < D / 2; ++c) reinterpret_cast<float2*>(p)[c] = make_float2(r[2 * c], r[2 * c + 1]);
    } else {
#pragma unroll
        for (int c = 0; c < D; ++c) p[c] = r[c];
    }
}

// y[i] += sum_d MT[d][i] * xrow[d]   (MT = transposed weight in LDS, xrow = this thread's row in LDS).
// The loop over d stays rolled: x is indexed dynamically (from LDS), y statically (registers), and
// only one weight row is live at a time -- a fully unrolled D x D matvec spills.
// the same with the vector's elements `xs` floats apart (the forward keeps its tokens element-major in LDS: [D][threads],
// conflict-free without a padding column -- the 1 KB that padding cost was what kept a third workgroup off the CU)
template <int D>
__device__ __forceinline__ void matvec_acc_s(const float* __restrict__ MT, const float* __restrict__ x, int xs, float (&y)[D]) {
#pragma unroll 2
    for (int d = 0; d < D; ++d) {
        const float xd = x[d * xs];
        float w[D];
        load_row<D>(MT + d * D, w);
#pragma unroll
        for (int i = 0; i < D; ++i) y[i] = fmaf(w[i], xd, y[i]);
    }
}
template <int D>
__device__ __forceinline__ void matvec_acc(const float* __restrict__ MT, const float* __restrict__ xrow, float (&y)[D]) {
#pragma unroll 2
    for (int d = 0; d < D; ++d) {
        const float xd = xrow[d];
        float w[D];
        load_row<D>(MT + d * D, w);
#pragma unroll
        for (int i = 0; i < D; ++i) y[i] = fmaf(w[i], xd, y[i]);
    }
}

// stage `nmat` [D][D] matrices transposed (dst[k][c][r] = src[k][r][c]) followed by `ntail` plain floats
template <int D>
__device__ __forceinline__ void stage_weights_t(float* __restrict__ dst, const float* __restrict__ src, int nmat,
                                                int ntail) {
    for (int i = threadIdx.x; i < nmat * D * D; i += blockDim.x) {
        const int k = i / (D * D), rc = i - k * D * D, r = rc / D, c = rc - r * D;
        dst[k * D * D + c * D + r] = src[i];
    }
    for (int i = threadIdx.x; i < ntail; i += blockDim.x) dst[nmat * D * D + i] = src[nmat * D * D + i];
}

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// block-wide sum / max through a small LDS scratch (red has >= 16 floats); all threads get the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < nw; ++k) t += red[k];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int k = 1; k < nw; ++k) t = fmaxf(t, red[k]);
    return t;
}

// ---------------------------------------------------------------------------------------------
// Attention dropout (cin_attention.py:86: `self.dropout(attn_weights)` between the softmax and P.V).  The keep mask
// is a counter-based hash of (seed, example, layer, head, query, key): nothing is stored, the backward regenerates
// the same bits.  One 32-bit finaliser per element on top of a per-(example, layer, head, query) row key.
struct AttnDrop {
    const unsigned long long* seed;     // device scalar, drawn per forward call from torch's generator
    unsigned thresh;                    // keep iff hash >= thresh   (thresh = p * 2^32)
    float keep_scale;                   // 1 / (1 - p)
};
__device__ __forceinline__ unsigned drop_mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned drop_row_key(unsigned long long seed, int b, int layer, int n_layers, int h, int nh,
                                                 int q) {
    const unsigned w0 = ((unsigned)b * (unsigned)n_layers + (unsigned)layer) * (unsigned)nh + (unsigned)h;
    const unsigned x = drop_mix((unsigned)seed ^ (w0 * 0x9E3779B1U));
    return drop_mix(x + (unsigned)(seed >> 32) + (unsigned)q * 0x85EBCA77U);
}
__device__ __forceinline__ float drop_keep(unsigned row_key, int t, unsigned thresh, float keep_scale) {
    return drop_mix(row_key + (unsigned)t * 0x9E3779B1U) >= thresh ? keep_scale : 0.f;
}

// The softmax runs in base 2 with the score scale folded into the query: q2 = q * (log2(e) / sqrt(head_dim)), so
// exp(score - max) = exp2(q2 . k - max2) is one subtraction and one v_exp_f32 per (query, key, head) -- no scale multiply,
// no log2(e) multiply.  Forward and backward use the same q2 and the same saved statistic lg = max2 + log2(sum), so the
// backward's p = exp2(q2 . k - lg) is the forward's normalised probability without another multiply; the factors the
// gradients owe (scale for dq, 1 / log2(e) for dk, which is accumulated against q2) are applied once per row at the end.
#define ATTN_LOG2E 1.4426950408889634f
#define ATTN_LN2 0.6931471805599453f
__device__ __forceinline__ float attn_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
template <int D, int NH>
__device__ __forceinline__ void head_dots(const float (&q)[D], const float (&k)[D], float (&sc)[NH]) {
    constexpr int HD = D / NH;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < HD; ++e) a = fmaf(q[h * HD + e], k[h * HD + e], a);
        sc[h] = a;
    }
}
// ---------------------------------------------------------------------------------------------
// scores of one query against key row t, all heads
template <int D, int NH>
__device__ __forceinline__ void head_scores(const float (&q)[D], const float (&k)[D], float scale, float (&sc)[NH]) {
    constexpr int HD = D / NH;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < HD; ++e) a = fmaf(q[h * HD + e], k[h * HD + e], a);
        sc[h] = a * scale;
    }
}

// Head-interleaved order of a D-vector: element e of head h at e*NH + h.  A head's dot product over its HD elements is
// a serial chain, and with the natural order [h][e] the four chains of a key row read registers that are HD apart, so
// hipcc leaves them as scalar FMAs; interleaved, step e of all heads reads NH adjacent registers and packs into
// v_pk_fma_f32 (two heads per instruction) -- same operations in the same order per head, so the same bits.  The forward keeps its K rows
// in LDS in this order (forward 0.64 -> 0.595 ms at config 3; the same treatment of the backward's two passes measured
// 4 % SLOWER -- the extra register-pair moves cost more than the packed FMAs save -- and is not in the tree).
template <int D, int NH>
__device__ __forceinline__ void head_interleave(const float (&x)[D], float (&y)[D]) {
    constexpr int HD = D / NH;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < HD; ++e) y[e * NH + h] = x[h * HD + e];
}
// per-head dot products of two head-interleaved vectors
typedef float attn_f2 __attribute__((ext_vector_type(2)));
template <int D, int NH>
__device__ __forceinline__ void head_dots_il(const float (&x)[D], const float (&y)[D], float (&a)[NH]) {
    constexpr int HD = D / NH;
    if constexpr (NH % 2 == 0) {
        // two heads per v_pk_fma_f32, spelled out: the chains end in scalar code (exp), which gives hipcc's SLP
        // vectoriser nothing to start from
        attn_f2 acc[NH / 2];
#pragma unroll
        for (int hp = 0; hp < NH / 2; ++hp) acc[hp] = (attn_f2){0.f, 0.f};
#pragma unroll
        for (int e = 0; e < HD; ++e)
#pragma unroll
            for (int hp = 0; hp < NH / 2; ++hp)
                acc[hp] = __builtin_elementwise_fma((attn_f2){x[e * NH + 2 * hp], x[e * NH + 2 * hp + 1]},
                                                    (attn_f2){y[e * NH + 2 * hp], y[e * NH + 2 * hp + 1]}, acc[hp]);
#pragma unroll
        for (int hp = 0; hp < NH / 2; ++hp) { a[2 * hp] = acc[hp].x; a[2 * hp + 1] = acc[hp].y; }
    } else {
#pragma unroll
        for (int h = 0; h < NH; ++h) a[h] = 0.f;
#pragma unroll
        for (int e = 0; e < HD; ++e)
#pragma unroll
            for (int h = 0; h < NH; ++h) a[h] = fmaf(x[e * NH + h], y[e * NH + h], a[h]);
    }
}
template <int D, int NH>
__device__ __forceinline__ void head_scores_il(const float (&q)[D], const float (&k)[D], float scale, float (&sc)[NH]) {
    head_dots_il<D, NH>(q, k, sc);
#pragma unroll
    for (int h = 0; h < NH; ++h) sc[h] *= scale;
}

// One MHSA layer for the token of this thread.  The token lives in the thread's column of Xs ([D][threads],
// element-major: conflict-free without padding) and is replaced there by the layer's output
// (post residual / LayerNorm), which is also returned in a.  Ks/Vs: [S][D] LDS; WT: this layer's
// weights in LDS, matrices transposed.  mx / ls: softmax statistics of the thread's query row.
template <int D, int NH, bool DROP>
__device__ __forceinline__ void mhsa_layer_fwd(float (&a)[D], float (&on)[D], float (&mx)[NH], float (&ls)[NH],
                                               float* Xs, float* Ks, float* Vs, const float* WT, int S, bool live,
                                               int use_ln, int use_res, const unsigned (&rk)[NH], unsigned thresh,
                                               float keep_scale) {
    constexpr int HD = D / NH;
    const int XS = blockDim.x;                           // Xs is element-major: element d of this thread's token at Xs[d * XS + s]
    const int s = threadIdx.x;
    float* xrow = Xs + s;
    const float scale = 1.0f / sqrtf((float)HD);
    float q[D];
    {
        float kk[D], vv[D];
#pragma unroll
        for (int i = 0; i < D; ++i) { q[i] = 0.f; kk[i] = 0.f; vv[i] = 0.f; }
        matvec_acc_s<D>(WT, xrow, XS, q);
        matvec_acc_s<D>(WT + D * D, xrow, XS, kk);
        matvec_acc_s<D>(WT + 2 * D * D, xrow, XS, vv);
        float ki[D];
        head_interleave<D, NH>(kk, ki);
        if (live) {
            store_row<D>(Ks + s * D, ki);                // K rows head-interleaved (head_interleave above)
            store_row<D>(Vs + s * D, vv);
        }
    }
    float qi[D];
    head_interleave<D, NH>(q, qi);
    const float c2 = scale * ATTN_LOG2E;
#pragma unroll
    for (int i = 0; i < D; ++i) qi[i] *= c2;            // q2: scores come out in base-2 units
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h) { mx[h] = -3.0e38f; ls[h] = 0.f; }
    // One pass over the keys, in blocks of KB: scores of the block, the running maximum moves to the block's maximum (the
    // sums so far are rescaled by exp2(old - new): nothing after the first blocks, where the maximum settles), then the
    // block's exponentials and P.V.  The separate maximum pass this replaces computed every score twice.
    float o[D];
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
    auto block = [&](auto kbc, int t0) {
        constexpr int KB = decltype(kbc)::value;
        float sc[KB][NH], bm[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) bm[h] = -3.0e38f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            float k[D];
            load_row<D>(Ks + (t0 + j) * D, k);
            head_dots_il<D, NH>(qi, k, sc[j]);
#pragma unroll
            for (int h = 0; h < NH; ++h) bm[h] = fmaxf(bm[h], sc[j][h]);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float mn = fmaxf(mx[h], bm[h]);
            const float corr = attn_exp2(mx[h] - mn);    // 0 for the first block (mx = -3e38), 1 once the maximum has settled
            ls[h] *= corr;
#pragma unroll
            for (int e = 0; e < HD; ++e) o[h * HD + e] *= corr;
            mx[h] = mn;
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            float v[D];
            load_row<D>(Vs + (t0 + j) * D, v);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float p = attn_exp2(sc[j][h] - mx[h]);
                ls[h] += p;                              // the softmax normaliser is over ALL keys; the mask comes after
                const float pm = DROP ? p * drop_keep(rk[h], t0 + j, thresh, keep_scale) : p;
#pragma unroll
                for (int e = 0; e < HD; ++e) o[h * HD + e] = fmaf(pm, v[h * HD + e], o[h * HD + e]);
            }
        }
    };
    {
        int t0 = 0;
        for (; t0 + 8 <= S; t0 += 8) block(std::integral_constant<int, 8>{}, t0);
        for (; t0 < S; ++t0) block(std::integral_constant<int, 1>{}, t0);
    }
    // a = (residual x) + W_o (o / l)
#pragma unroll
    for (int d = 0; d < D; ++d) a[d] = use_res ? xrow[d * XS] : 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float inv = 1.0f / ls[h];
#pragma unroll
        for (int e = 0; e < HD; ++e) {
            on[h * HD + e] = o[h * HD + e] * inv;
            xrow[(h * HD + e) * XS] = on[h * HD + e];                       // own column: no barrier needed
        }
    }
    matvec_acc_s<D>(WT + 3 * D * D, xrow, XS, a);
    if (use_ln) {
        const float* g = WT + 4 * D * D;
        float mean = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) mean += a[d];
        mean *= (1.0f / D);
        float var = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) var = fmaf(a[d] - mean, a[d] - mean, var);
        const float rstd = rsqrtf(var * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int d = 0; d < D; ++d) a[d] = (a[d] - mean) * rstd * g[d] + g[D + d];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) xrow[d * XS] = a[d];
}

// TB = compile-time bound of the block size (round_up(S, 64) threads): 512 leaves 256 VGPRs per lane
// (2 waves/SIMD), 1024 only 128 -- the host picks the smallest that covers S.
template <int D, int NH, int TB, bool DROP>
__global__ __launch_bounds__(TB) void attn_pool_fwd_kernel(
    const float* __restrict__ fm, long N, int B, int S, int n_layers, int use_ln, int use_res,
    const float* __restrict__ theta, float* __restrict__ out, float* __restrict__ tok_save,
    float* __restrict__ o_save, float* __restrict__ ml_save, AttnDrop drop) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int XS = blockDim.x;
    float* Ks = smem;                                    // [S][D]
    float* Vs = Ks + S * D;                              // [S][D]
    float* Ws = Vs + S * D;                              // 4*D*D + 2*D floats (weights of the current stage)
    float* red = Ws + 4 * D * D + 2 * D;                 // 16 + (waves of the block) * D floats
    float* Xs = red + 16 + (XS >> 6) * D;                // [D][blockDim] tokens, element-major: one column per thread
    const int b = blockIdx.x;
    const int s = threadIdx.x;
    const bool live = s < S;
    const int lsz = 4 * D * D + (use_ln ? 2 * D : 0);

    float* xrow = Xs + s;
#pragma unroll
    for (int d = 0; d < D; ++d) {                        // unconditional load from a clamped row, masked afterwards:
        const float t = fm[(long)(live ? s : 0) * N + (long)b * D + d];   // behind `live ? load : 0` hipcc issues the loads one by one
        xrow[d * XS] = live ? t : 0.f;
    }

    for (int layer = 0; layer < n_layers; ++layer) {
        __syncthreads();                                 // previous layer's reads of Ks/Vs/Ws are done
        stage_weights_t<D>(Ws, theta + (long)layer * lsz, 4, use_ln ? 2 * D : 0);
        __syncthreads();
        float a[D], on[D], mx[NH], ls[NH];
        unsigned rk[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) rk[h] = DROP ? drop_row_key(*drop.seed, b, layer, n_layers, h, NH, s) : 0u;
        mhsa_layer_fwd<D, NH, DROP>(a, on, mx, ls, Xs, Ks, Vs, Ws, S, live, use_ln, use_res, rk, drop.thresh,
                                    drop.keep_scale);
        if (live) {
            // saved for backward: the layer's output tokens, its attention output (before W_o) and the
            // softmax statistics (max, 1/sum) of every query row
            store_row<D>(tok_save + (((long)layer * B + b) * S + s) * D, a);
            store_row<D>(o_save + (((long)layer * B + b) * S + s) * D, on);
            float* ml = ml_save + ((((long)layer * B + b) * S + s) * NH) * 2;
#pragma unroll
            for (int h = 0; h < NH; ++h) { ml[2 * h] = mx[h] + __builtin_amdgcn_logf(ls[h]); ml[2 * h + 1] = 1.0f / ls[h]; }   // lg (base 2), 1 / sum
        }
    }

    // attention pooling: softmax_s( w2 . tanh(W1 x_s + b1) ) weighted sum of the tokens
    const float* tp = theta + (long)n_layers * lsz;
    __syncthreads();
    stage_weights_t<D>(Ws, tp, 1, 2 * D);                // W1^T, b1, w2
    __syncthreads();
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = xrow[d * XS];
    float e = 0.f;
    {
        float u[D];
#pragma unroll
        for (int i = 0; i < D; ++i) u[i] = Ws[D * D + i];
        matvec_acc_s<D>(Ws, xrow, XS, u);
#pragma unroll
        for (int i = 0; i < D; ++i) e = fmaf(Ws[D * D + D + i], tanhf(u[i]), e);
    }
    const float emax = block_max(live ? e : -3.0e38f, red);
    const float w = live ? __expf(e - emax) : 0.f;
    const float wsum = block_sum(w, red);
    const float alpha = w / wsum;
    // pooled[d] = sum_s alpha_s x_s[d]: wave shuffle reduction, then across waves through LDS
    float* pw = red + 16;
    const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float t = wave_sum(alpha * x[d]);
        if ((threadIdx.x & 63) == 0) pw[wv * D + d] = t;
    }
    __syncthreads();
    float* pooled = Ks;                                  // K/V are free now
    if (threadIdx.x < D) {
        float t = 0.f;
        for (int k = 0; k < nw; ++k) t += pw[k * D + threadIdx.x];
        pooled[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < D) out[(long)b * D + threadIdx.x] = pooled[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
static size_t attn_fwd_lds(int S, int D) {
    const long threads = round_up(S, 64);
    return (size_t)(2 * S * D + 4 * D * D + 2 * D + 16 + (threads / 64) * D + threads * D) * sizeof(float);
}

template <int D, int NH>
static int launch_attn_fwd(const float* fm, int B, int S, int n_layers, int use_ln, int use_res,
                           const float* theta, float* out, float* tok_save, float* o_save, float* ml_save,
                           AttnDrop drop, hipStream_t st) {
    const int threads = (int)round_up(S, 64);
    const size_t lds = attn_fwd_lds(S, D);
#define XDFM_ATTN_FWD(TB, DR)                                                                                       \
    hipLaunchKernelGGL((attn_pool_fwd_kernel<D, NH, TB, DR>), dim3(B), dim3(threads), lds, st, fm, (long)B * D, B, S, \
                       n_layers, use_ln, use_res, theta, out, tok_save, o_save, ml_save, drop)
    if (drop.seed) {
        if (threads <= 512) XDFM_ATTN_FWD(512, true); else XDFM_ATTN_FWD(1024, true);
    } else {
        if (threads <= 512) XDFM_ATTN_FWD(512, false); else XDFM_ATTN_FWD(1024, false);
    }
#undef XDFM_ATTN_FWD
    return xdfm_check_launch("cin_attn_pool_fwd");
}

#define ATTN_DISPATCH(FN, ...)                                                                  \
    switch (D * 16 + nh) {                                                                      \
        case 4 * 16 + 1: return FN<4, 1>(__VA_ARGS__);                                          \
        case 4 * 16 + 2: return FN<4, 2>(__VA_ARGS__);                                          \
        case 4 * 16 + 4: return FN<4, 4>(__VA_ARGS__);                                          \
        case 8 * 16 + 1: return FN<8, 1>(__VA_ARGS__);                                          \
        case 8 * 16 + 2: return FN<8, 2>(__VA_ARGS__);                                          \
        case 8 * 16 + 4: return FN<8, 4>(__VA_ARGS__);                                          \
        case 10 * 16 + 1: return FN<10, 1>(__VA_ARGS__);                                        \
        case 10 * 16 + 2: return FN<10, 2>(__VA_ARGS__);                                        \
        case 16 * 16 + 1: return FN<16, 1>(__VA_ARGS__);                                        \
        case 16 * 16 + 2: return FN<16, 2>(__VA_ARGS__);                                        \
        case 16 * 16 + 4: return FN<16, 4>(__VA_ARGS__);                                        \
        case 16 * 16 + 8: return FN<16, 8>(__VA_ARGS__);                                        \
        case 32 * 16 + 2: return FN<32, 2>(__VA_ARGS__);                                        \
        case 32 * 16 + 4: return FN<32, 4>(__VA_ARGS__);                                        \
        case 32 * 16 + 8: return FN<32, 8>(__VA_ARGS__);                                        \
        default:                                                                                \
            return xdfm_fail(XDFM_ERR_INVALID, "cin_attn_pool: (embedding_dim %d, heads %d) has no kernel instance", D, nh); \
    }

// =============================================================================================
// backward
// =============================================================================================
// dst[i*D + d] += sum_s A[s][i] * G[s*gp + d]   (A: [S][pitch] LDS tile; G: rows in global memory,
// pitch gp -- the layer's input tokens or its saved attention output, L2-resident)
template <int D>
__device__ __forceinline__ void outer_accumulate(float* __restrict__ dst, const float* __restrict__ A, int pitch,
                                                 const float* __restrict__ G, long gp, int S) {
    for (int idx = threadIdx.x; idx < D * D; idx += blockDim.x) {
        const int i = idx / D, d = idx - i * D;
        // 16 rows per batch: the global (L2-resident) loads of a batch are issued back to back -- with 4 the loop
        // paid one L2 round trip per 4 rows, five times per layer and example
        float acc0 = 0.f, acc1 = 0.f;
        int s = 0;
        for (; s + 16 <= S; s += 16) {
            float g[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) g[k] = G[(long)(s + k) * gp + d];
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                acc0 = fmaf(A[(s + k) * pitch + i], g[k], acc0);
                acc1 = fmaf(A[(s + k + 1) * pitch + i], g[k + 1], acc1);
            }
        }
        for (; s < S; ++s) acc0 = fmaf(A[s * pitch + i], G[(long)s * gp + d], acc0);
        dst[idx] += acc0 + acc1;
    }
}

// dst[d] += sum over the block of v[d]: wave shuffles, the waves' sums parked in LDS (scratch: 16 * D floats) and added in
// wave order by one thread per element -- no float atomics, the same bits on every run (an LDS atomic per wave and
// element, as before round 3, added the waves in whatever order they arrived).  Every thread of the block calls it.
template <int D>
__device__ __forceinline__ void vec_accumulate(float* __restrict__ dst, const float (&v)[D], float* __restrict__ scratch) {
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float t = wave_sum(v[d]);
        if ((threadIdx.x & 63) == 0) scratch[w * D + d] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < D) {
        float s = 0.f;
        for (int k = 0; k < nw; ++k) s += scratch[k * D + threadIdx.x];
        dst[threadIdx.x] += s;
    }
    __syncthreads();
}

// LDS carve-up of the backward kernel (floats).  TP = rows of the per-thread tiles = blockDim.
struct AttnBwdLds {
    int kv, z, st, w, acc, red, total;
};
static __host__ __device__ inline AttnBwdLds attn_bwd_layout(int S, int D, int NH, int n_layers, int TP) {
    AttnBwdLds L;
    int off = 0;
    L.kv = off;  off += 2 * S * D;                   // K,V  then Q,dO
    L.z = off;   off += TP * (D + 1);                // the one per-thread tile: dy / du / dq / dk / dv rows
    L.st = off;  off += S * 4 * NH;                  // per query: max, 1/sum, delta, dropout row key  per head
    L.w = off;   off += 8 * D * D + 2 * D;           // transposed + plain weights of the stage
    L.acc = off; off += n_layers * (4 * D * D + 2 * D) + D * D + 2 * D;   // parameter-gradient accumulator
    L.red = off; off += 16 + 16 * D;
    L.total = off;
    return L;
}

template <int D, int NH, int TB, bool DROP>
__global__ __launch_bounds__(TB) void attn_pool_bwd_kernel(
    const float* __restrict__ fm, long N, int B, int S, int n_layers, int use_ln, int use_res,
    const float* __restrict__ theta, const float* __restrict__ tok_save, const float* __restrict__ o_save,
    const float* __restrict__ ml_save, const float* __restrict__ dout, float* __restrict__ dfm,
    float* __restrict__ dtheta, float* __restrict__ part, AttnDrop drop) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HD = D / NH;
    constexpr int XP = D + 1;
    const AttnBwdLds L = attn_bwd_layout(S, D, NH, n_layers, blockDim.x);
    float* Ks = smem + L.kv;
    float* Vs = Ks + S * D;
    float* Zs = smem + L.z;
    float* St = smem + L.st;
    float* WT = smem + L.w;                 // 4 transposed matrices (forward products)
    float* WN = WT + 4 * D * D;             // 4 plain matrices (transposed products), then gamma/beta
    float* Acc = smem + L.acc;
    float* red = smem + L.red;
    const int s = threadIdx.x;
    const bool live = s < S;
    const int lsz = 4 * D * D + (use_ln ? 2 * D : 0);
    const int asz = n_layers * lsz + D * D + 2 * D;
    const float scale = 1.0f / sqrtf((float)HD);
    float* zrow = Zs + s * XP;
    const int sl = live ? s : 0;                // dead threads read row 0 and contribute nothing

    for (int i = threadIdx.x; i < asz; i += blockDim.x) Acc[i] = 0.f;

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // ------------------------------------------------------------------ pooling backward
        const float* tp = theta + (long)n_layers * lsz;
        float* accp = Acc + n_layers * lsz;             // [dW1 (D*D) | db1 (D) | dw2 (D)]
        __syncthreads();
        stage_weights_t<D>(WT, tp, 1, 0);                                   // W1^T
        for (int i = threadIdx.x; i < D * D + 2 * D; i += blockDim.x) WN[i] = tp[i];   // W1, b1, w2
        const float* tok_in = tok_save + ((long)(n_layers - 1) * B + b) * S * D;   // tokens entering the pooling
        const float* xrow = tok_in + (long)sl * D;       // this thread's token (global, L1/L2 resident)
        __syncthreads();
        float da[D];                                     // gradient w.r.t. the tokens entering the pooling
        {
            float u[D], th[D];
#pragma unroll
            for (int i = 0; i < D; ++i) u[i] = WN[D * D + i];
            matvec_acc<D>(WT, xrow, u);
            float e = 0.f;
#pragma unroll
            for (int i = 0; i < D; ++i) { th[i] = tanhf(u[i]); e = fmaf(WN[D * D + D + i], th[i], e); }
            const float emax = block_max(live ? e : -3.0e38f, red);
            const float w = live ? __expf(e - emax) : 0.f;
            const float wsum = block_sum(w, red);
            const float alpha = w / wsum;
            float dpool[D];
#pragma unroll
            for (int d = 0; d < D; ++d) dpool[d] = dout[(long)b * D + d];
            float dalpha = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) dalpha = fmaf(xrow[d], dpool[d], dalpha);
            if (!live) dalpha = 0.f;
            const float c = block_sum(alpha * dalpha, red);
            const float de = alpha * (dalpha - c);       // softmax backward
            float du[D], g2[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                g2[i] = de * th[i];                                         // d w2
                du[i] = de * WN[D * D + D + i] * (1.f - th[i] * th[i]);      // through tanh
                zrow[i] = du[i];
            }
            vec_accumulate<D>(accp + D * D + D, g2, red + 16);
            vec_accumulate<D>(accp + D * D, du, red + 16);
#pragma unroll
            for (int d = 0; d < D; ++d) da[d] = alpha * dpool[d];
            matvec_acc<D>(WN, zrow, da);                 // += W1^T du
            __syncthreads();
            outer_accumulate<D>(accp, Zs, XP, tok_in, D, S);    // dW1[i][d] += sum_s du_s[i] x_s[d]
        }

        // ------------------------------------------------------------------ MHSA layers, last to first
        for (int layer = n_layers - 1; layer >= 0; --layer) {
            const float* th_l = theta + (long)layer * lsz;
            float* accl = Acc + layer * lsz;             // [dWq | dWk | dWv | dWo | dgamma | dbeta]
            __syncthreads();                             // previous stage done with Xs / Zs / weights
            stage_weights_t<D>(WT, th_l, 4, 0);
            for (int i = threadIdx.x; i < lsz; i += blockDim.x) WN[i] = th_l[i];
            // tokens entering this layer and its saved attention output: global rows (pitch xgp / D)
            const float* xg = (layer == 0) ? fm + (long)b * D : tok_save + ((long)(layer - 1) * B + b) * S * D;
            const long xgp = (layer == 0) ? N : D;
            const float* og = o_save + ((long)layer * B + b) * S * D;
            const float* xrow = xg + (long)sl * xgp;
            const float* orow = og + (long)sl * D;
            float lg[NH];                                // base-2 log-sum-exp of the query's scores: p = exp2(q2 . k - lg)
            {
                const float* ml = ml_save + ((((long)layer * B + b) * S + (live ? s : 0)) * NH) * 2;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float t0 = ml[2 * h];                          // unconditional (clamped row), then masked
                    lg[h] = live ? t0 : 1.0e30f;                         // a dead thread's probabilities are 0
                }
            }
            __syncthreads();
            // recompute q, k, v and the attention output o
            float q[D], kk[D], vv[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { q[i] = 0.f; kk[i] = 0.f; vv[i] = 0.f; }
            matvec_acc<D>(WT, xrow, q);
            matvec_acc<D>(WT + D * D, xrow, kk);
            matvec_acc<D>(WT + 2 * D * D, xrow, vv);
            // NH even: every per-head quantity of the two passes below is kept head-interleaved (element e of head h at
            // e*NH + h, as the forward's K rows), so that one v_pk_fma_f32 serves two heads: K / V rows in LDS, q2, dO, and the
            // accumulators dq, dk, dv (de-interleaved once at the end)
            constexpr bool IL = (NH % 2 == 0);
            if (IL) { float t[D]; head_interleave<D, NH>(kk, t);
#pragma unroll
                for (int d = 0; d < D; ++d) kk[d] = t[d];
                head_interleave<D, NH>(vv, t);
#pragma unroll
                for (int d = 0; d < D; ++d) vv[d] = t[d]; }
            if (live) {
                store_row<D>(Ks + s * D, kk);
                store_row<D>(Vs + s * D, vv);
            }
            __syncthreads();
            float o[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float t = orow[d];
                o[d] = live ? t : 0.f;
            }
            // y = W_o o (+x), LayerNorm backward -> dy
            float dy[D];
            {
                float y[D];
#pragma unroll
                for (int d = 0; d < D; ++d) y[d] = use_res ? xrow[d] : 0.f;
                matvec_acc<D>(WT + 3 * D * D, orow, y);
                if (use_ln) {
                    const float* g = WN + 4 * D * D;
                    float mean = 0.f;
#pragma unroll
                    for (int d = 0; d < D; ++d) mean += y[d];
                    mean *= (1.0f / D);
                    float var = 0.f;
#pragma unroll
                    for (int d = 0; d < D; ++d) var = fmaf(y[d] - mean, y[d] - mean, var);
                    const float rstd = rsqrtf(var * (1.0f / D) + 1e-5f);
                    float yh[D], dg[D], m1 = 0.f, m2 = 0.f;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        yh[d] = (y[d] - mean) * rstd;
                        dg[d] = da[d] * yh[d];
                        const float dyh = da[d] * g[d];
                        m1 += dyh;
                        m2 = fmaf(dyh, yh[d], m2);
                    }
                    m1 *= (1.0f / D);
                    m2 *= (1.0f / D);
#pragma unroll
                    for (int d = 0; d < D; ++d) dy[d] = rstd * (da[d] * g[d] - m1 - yh[d] * m2);
                    vec_accumulate<D>(accl + 4 * D * D, dg, red + 16);          // dgamma
                    vec_accumulate<D>(accl + 4 * D * D + D, da, red + 16);      // dbeta
                } else {
#pragma unroll
                    for (int d = 0; d < D; ++d) dy[d] = da[d];
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) zrow[d] = dy[d];
            __syncthreads();
            outer_accumulate<D>(accl + 3 * D * D, Zs, XP, og, D, S);  // dWo[i][d] += sum_s dy_s[i] o_s[d]
            // do = W_o^T dy ; delta_h = do_h . o_h ; dx starts as the residual branch
            float dO[D], delta[NH], dx[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { dO[d] = 0.f; dx[d] = use_res ? dy[d] : 0.f; }
            matvec_acc<D>(WN + 3 * D * D, zrow, dO);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < HD; ++e) a = fmaf(dO[h * HD + e], o[h * HD + e], a);
                delta[h] = a;
            }
            // pass A (thread = query): dq.  Under dropout P' = P.M/(1-p): dP = M/(1-p) . (dO V^T), and
            // delta = sum_t P dP = dO . o still holds because o was built from P'.
            unsigned rk[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) rk[h] = DROP ? drop_row_key(*drop.seed, b, layer, n_layers, h, NH, s) : 0u;
            float dq[D], q2[D];
            const float c2 = scale * ATTN_LOG2E;
#pragma unroll
            for (int d = 0; d < D; ++d) { dq[d] = 0.f; q2[d] = q[d] * c2; }
            if constexpr (IL) {
                float q2i[D], dOi[D];
                head_interleave<D, NH>(q2, q2i);
                head_interleave<D, NH>(dO, dOi);
                attn_f2 dqi[D / 2], lg2[NH / 2], dl2[NH / 2];
#pragma unroll
                for (int i = 0; i < D / 2; ++i) dqi[i] = (attn_f2){0.f, 0.f};
#pragma unroll
                for (int hp = 0; hp < NH / 2; ++hp) { lg2[hp] = (attn_f2){lg[2 * hp], lg[2 * hp + 1]}; dl2[hp] = (attn_f2){delta[2 * hp], delta[2 * hp + 1]}; }
#pragma unroll 2
                for (int t = 0; t < S; ++t) {
                    float k[D], v[D];
                    load_row<D>(Ks + t * D, k);
                    load_row<D>(Vs + t * D, v);
#pragma unroll
                    for (int hp = 0; hp < NH / 2; ++hp) {
                        attn_f2 sc = {0.f, 0.f}, dp = {0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < HD; ++e) {
                            const attn_f2 kp = {k[e * NH + 2 * hp], k[e * NH + 2 * hp + 1]};
                            const attn_f2 vp = {v[e * NH + 2 * hp], v[e * NH + 2 * hp + 1]};
                            sc = __builtin_elementwise_fma((attn_f2){q2i[e * NH + 2 * hp], q2i[e * NH + 2 * hp + 1]}, kp, sc);
                            dp = __builtin_elementwise_fma((attn_f2){dOi[e * NH + 2 * hp], dOi[e * NH + 2 * hp + 1]}, vp, dp);
                        }
                        sc -= lg2[hp];
                        const attn_f2 pr = {attn_exp2(sc.x), attn_exp2(sc.y)};
                        if (DROP) dp *= (attn_f2){drop_keep(rk[2 * hp], t, drop.thresh, drop.keep_scale),
                                                  drop_keep(rk[2 * hp + 1], t, drop.thresh, drop.keep_scale)};
                        const attn_f2 ds = pr * (dp - dl2[hp]);           // * scale: once per row, below
#pragma unroll
                        for (int e = 0; e < HD; ++e)
                            dqi[(e * NH + 2 * hp) / 2] = __builtin_elementwise_fma(ds, (attn_f2){k[e * NH + 2 * hp], k[e * NH + 2 * hp + 1]},
                                                                                   dqi[(e * NH + 2 * hp) / 2]);
                    }
                }
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int e = 0; e < HD; ++e) {
                        const int i = e * NH + h;
                        dq[h * HD + e] = (i & 1) ? dqi[i / 2].y : dqi[i / 2].x;
                    }
            } else {
#pragma unroll 2
            for (int t = 0; t < S; ++t) {
                float k[D], v[D], sc[NH];
                load_row<D>(Ks + t * D, k);
                load_row<D>(Vs + t * D, v);
                head_dots<D, NH>(q2, k, sc);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float p = attn_exp2(sc[h] - lg[h]);
                    float dp = 0.f;
#pragma unroll
                    for (int e = 0; e < HD; ++e) dp = fmaf(dO[h * HD + e], v[h * HD + e], dp);
                    if (DROP) dp *= drop_keep(rk[h], t, drop.thresh, drop.keep_scale);
                    const float ds = p * (dp - delta[h]);               // * scale: once per row, below
#pragma unroll
                    for (int e = 0; e < HD; ++e) dq[h * HD + e] = fmaf(ds, k[h * HD + e], dq[h * HD + e]);
                }
            }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) dq[d] *= scale;
            __syncthreads();                             // all reads of K / V and of Ys / Zs are done
            // stage Q, dO and the row statistics for the key-side pass (over the K / V storage)
            float* Qs = Ks;
            float* dOs = Vs;
            if (live) {
                if constexpr (IL) {
                    float t1[D];
                    head_interleave<D, NH>(q2, t1);
                    store_row<D>(Qs + s * D, t1);
                    head_interleave<D, NH>(dO, t1);
                    store_row<D>(dOs + s * D, t1);
                    // statistics by kind: [lg of the NH heads | delta | dropout row key | -]: a head pair is 8 adjacent bytes
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        St[s * 4 * NH + h] = lg[h];
                        St[s * 4 * NH + NH + h] = delta[h];
                        St[s * 4 * NH + 2 * NH + h] = __uint_as_float(rk[h]);
                    }
                } else {
                store_row<D>(Qs + s * D, q2);
                store_row<D>(dOs + s * D, dO);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    St[(s * NH + h) * 4] = lg[h];
                    St[(s * NH + h) * 4 + 1] = 0.f;
                    St[(s * NH + h) * 4 + 2] = delta[h];
                    St[(s * NH + h) * 4 + 3] = __uint_as_float(rk[h]);
                }
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) zrow[d] = dq[d];
            __syncthreads();
            outer_accumulate<D>(accl, Zs, XP, xg, xgp, S);    // dWq[i][d] += sum_s dq_s[i] x_s[d]
            matvec_acc<D>(WN, zrow, dx);                 // dx += Wq^T dq
            // pass B (thread = key): dk, dv
            float dk[D], dv[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
            if constexpr (IL) {
                attn_f2 dki[D / 2], dvi[D / 2];
#pragma unroll
                for (int i = 0; i < D / 2; ++i) { dki[i] = (attn_f2){0.f, 0.f}; dvi[i] = (attn_f2){0.f, 0.f}; }
#pragma unroll 2
                for (int r = 0; r < S; ++r) {
                    float qs[D], dos[D], stl[NH], std_[NH], strk[NH];
                    load_row<D>(Qs + r * D, qs);
                    load_row<D>(dOs + r * D, dos);
                    load_row<NH>(St + r * 4 * NH, stl);
                    load_row<NH>(St + r * 4 * NH + NH, std_);
                    if (DROP) load_row<NH>(St + r * 4 * NH + 2 * NH, strk);
#pragma unroll
                    for (int hp = 0; hp < NH / 2; ++hp) {
                        attn_f2 sc = {0.f, 0.f}, dp = {0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < HD; ++e) {
                            const int i = e * NH + 2 * hp;
                            sc = __builtin_elementwise_fma((attn_f2){qs[i], qs[i + 1]}, (attn_f2){kk[i], kk[i + 1]}, sc);
                            dp = __builtin_elementwise_fma((attn_f2){dos[i], dos[i + 1]}, (attn_f2){vv[i], vv[i + 1]}, dp);
                        }
                        sc -= (attn_f2){stl[2 * hp], stl[2 * hp + 1]};
                        const attn_f2 pr = {attn_exp2(sc.x), attn_exp2(sc.y)};
                        attn_f2 pv = pr;
                        if (DROP) {
                            const attn_f2 keep = {drop_keep(__float_as_uint(strk[2 * hp]), s, drop.thresh, drop.keep_scale),
                                                  drop_keep(__float_as_uint(strk[2 * hp + 1]), s, drop.thresh, drop.keep_scale)};
                            dp *= keep;
                            pv = pr * keep;
                        }
                        const attn_f2 ds = pr * (dp - (attn_f2){std_[2 * hp], std_[2 * hp + 1]});    // * q2 / log2(e): below
#pragma unroll
                        for (int e = 0; e < HD; ++e) {
                            const int i = e * NH + 2 * hp;
                            dvi[i / 2] = __builtin_elementwise_fma(pv, (attn_f2){dos[i], dos[i + 1]}, dvi[i / 2]);
                            dki[i / 2] = __builtin_elementwise_fma(ds, (attn_f2){qs[i], qs[i + 1]}, dki[i / 2]);
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int e = 0; e < HD; ++e) {
                        const int i = e * NH + h;
                        dk[h * HD + e] = (i & 1) ? dki[i / 2].y : dki[i / 2].x;
                        dv[h * HD + e] = (i & 1) ? dvi[i / 2].y : dvi[i / 2].x;
                    }
            } else {
#pragma unroll 2
            for (int r = 0; r < S; ++r) {
                float qs[D], dos[D], sc[NH];
                load_row<D>(Qs + r * D, qs);
                load_row<D>(dOs + r * D, dos);
                head_dots<D, NH>(qs, kk, sc);                         // qs = the query's q2
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float4 stq = *reinterpret_cast<const float4*>(St + (r * NH + h) * 4);
                    const float p = attn_exp2(sc[h] - stq.x);
                    const float keep = DROP ? drop_keep(__float_as_uint(stq.w), s, drop.thresh, drop.keep_scale) : 1.f;
                    float dp = 0.f;
#pragma unroll
                    for (int e = 0; e < HD; ++e) dp = fmaf(dos[h * HD + e], vv[h * HD + e], dp);
                    if (DROP) dp *= keep;
                    const float ds = p * (dp - stq.z);                  // * scale * q = * q2 / log2(e): once per row, below
                    const float pv = DROP ? p * keep : p;
#pragma unroll
                    for (int e = 0; e < HD; ++e) {
                        dv[h * HD + e] = fmaf(pv, dos[h * HD + e], dv[h * HD + e]);
                        dk[h * HD + e] = fmaf(ds, qs[h * HD + e], dk[h * HD + e]);
                    }
                }
            }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) dk[d] *= ATTN_LN2;
            if (!live) {
#pragma unroll
                for (int d = 0; d < D; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
            }
            __syncthreads();                             // outer_accumulate(dq) finished reading Zs
#pragma unroll
            for (int d = 0; d < D; ++d) zrow[d] = dk[d];
            __syncthreads();
            outer_accumulate<D>(accl + D * D, Zs, XP, xg, xgp, S);     // dWk
            matvec_acc<D>(WN + D * D, zrow, dx);                       // dx += Wk^T dk
            __syncthreads();
#pragma unroll
            for (int d = 0; d < D; ++d) zrow[d] = dv[d];
            __syncthreads();
            outer_accumulate<D>(accl + 2 * D * D, Zs, XP, xg, xgp, S); // dWv
            matvec_acc<D>(WN + 2 * D * D, zrow, dx);                   // dx += Wv^T dv
            // dx is the gradient w.r.t. the tokens entering this layer
            if (layer == 0) {
                if (live) {
#pragma unroll
                    for (int d = 0; d < D; ++d) dfm[(long)s * N + (long)b * D + d] = dx[d];
                }
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) da[d] = dx[d];
            }
        }
    }
    __syncthreads();
    // the workgroup's share of the parameter gradients: its own row of `part`, summed over the workgroups in a fixed order
    // by attn_dtheta_finish_kernel -- or (no workspace: the round-2 entry point) fp32 atomics, whose order varies
    if (part) { for (int i = threadIdx.x; i < asz; i += blockDim.x) part[(long)blockIdx.x * asz + i] = Acc[i]; }
    else { for (int i = threadIdx.x; i < asz; i += blockDim.x) atomicAdd(&dtheta[i], Acc[i]); }
}

// dtheta[i] = sum over the workgroups of part[g][i] in a FIXED tree: 16 contiguous ranges of workgroups, each summed in
// order by one thread (8 loads in flight), the 16 range sums added in range order -- the same bits on every run, and 16
// times the parallelism of one chain per element (a single chain over 2048 partials took 99 us at config 3).
#define ATTN_FIN_R 16
__global__ __launch_bounds__(64 * ATTN_FIN_R) void attn_dtheta_finish_kernel(const float* __restrict__ part, int asz, int groups,
                                                                             float* __restrict__ dtheta) {
    __shared__ float red[ATTN_FIN_R][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + c;
    const int per = (groups + ATTN_FIN_R - 1) / ATTN_FIN_R;
    const int g0 = r * per, g1 = min(groups, g0 + per);
    float s = 0.f;
    if (i < asz) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = part[(long)(g + q) * asz + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += t[q];
        }
        for (; g < g1; ++g) s += part[(long)g * asz + i];
    }
    red[r][c] = s;
    __syncthreads();
    if (r == 0 && i < asz) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < ATTN_FIN_R; ++q) t += red[q][c];
        dtheta[i] = t;
    }
}

static inline int attn_bwd_groups(int B) { return B < 2048 ? B : 2048; }       // workgroups of the backward launch

template <int D, int NH>
static int launch_attn_bwd(const float* fm, int B, int S, int n_layers, int use_ln, int use_res, const float* theta,
                           const float* tok_save, const float* o_save, const float* ml_save, const float* dout,
                           float* dfm, float* dtheta, float* part, AttnDrop drop, hipStream_t st) {
    const int threads = (int)round_up(S, 64);
    const size_t lds = (size_t)attn_bwd_layout(S, D, NH, n_layers, threads).total * sizeof(float);
    if (lds > 160 * 1024) return xdfm_fail(XDFM_ERR_INVALID, "cin_attn_pool_bwd: S=%d D=%d does not fit LDS", S, D);
    const int grid = attn_bwd_groups(B);
#define XDFM_ATTN_BWD(TB, DR)                                                                                          \
    hipLaunchKernelGGL((attn_pool_bwd_kernel<D, NH, TB, DR>), dim3(grid), dim3(threads), lds, st, fm, (long)B * D, B, S, \
                       n_layers, use_ln, use_res, theta, tok_save, o_save, ml_save, dout, dfm, dtheta, part, drop)
    if (drop.seed) {
        if (threads <= 512) XDFM_ATTN_BWD(512, true); else XDFM_ATTN_BWD(1024, true);
    } else {
        if (threads <= 512) XDFM_ATTN_BWD(512, false); else XDFM_ATTN_BWD(1024, false);
    }
#undef XDFM_ATTN_BWD
    if (part) {
        const int asz = n_layers * (4 * D * D + (use_ln ? 2 * D : 0)) + D * D + 2 * D;
        hipLaunchKernelGGL(attn_dtheta_finish_kernel, dim3(ceil_div(asz, 64)), dim3(64 * ATTN_FIN_R), 0, st, part, asz, grid, dtheta);
    }
    return xdfm_check_launch("cin_attn_pool_bwd");
}

// The keep mask the kernels above regenerate, written out: keep[layer][b][h][q][t] (1 = kept).  Test hook -- the
// parity tests feed it to the oracle, which applies it where the reference applies nn.Dropout.
__global__ void attn_dropout_mask_kernel(int B, int S, int nh, int n_layers, AttnDrop drop, unsigned char* keep) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (layer, b, h, q)
    if (row >= (long)n_layers * B * nh * S) return;
    const int q = (int)(row % S);
    long r = row / S;
    const int h = (int)(r % nh); r /= nh;
    const int b = (int)(r % B);
    const int layer = (int)(r / B);
    const unsigned rk = drop_row_key(*drop.seed, b, layer, n_layers, h, nh, q);
    for (int t = 0; t < S; ++t) keep[row * S + t] = drop_keep(rk, t, drop.thresh, 1.f) != 0.f;
}

static int attn_drop_args(float p_drop, const unsigned long long* seed, AttnDrop* d, const char* who) {
    d->seed = nullptr; d->thresh = 0; d->keep_scale = 1.f;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return xdfm_fail(XDFM_ERR_INVALID, "%s: dropout p=%g outside [0, 1)", who, p_drop);
    if (p_drop == 0.f) return 0;
    if (!seed) return xdfm_fail(XDFM_ERR_INVALID, "%s: dropout p=%g needs a device seed", who, p_drop);
    d->seed = seed;
    const double t = (double)p_drop * 4294967296.0;
    d->thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    d->keep_scale = 1.0f / (1.0f - p_drop);
    return 0;
}

extern "C" {

size_t xdfm_cin_attn_theta_elems(int D, int n_layers, int use_ln) {
    return (size_t)n_layers * (4 * D * D + (use_ln ? 2 * D : 0)) + (size_t)D * D + 2 * D;
}

int xdfm_cin_attn_pool_fwd(const float* fm, int B, int S, int D, int nh, int n_layers, int use_ln, int use_res,
                           const float* theta, float* out, float* tok_save, float* o_save, float* ml_save,
                           float p_drop, const unsigned long long* drop_seed, void* stream) {
    XDFM_REQUIRE(fm && theta && out && ml_save && tok_save && o_save, "cin_attn_pool_fwd: null pointer");
    XDFM_REQUIRE(B > 0 && S > 0 && S <= 1024 && n_layers >= 1, "cin_attn_pool_fwd: bad shape B=%d S=%d layers=%d", B, S,
                 n_layers);
    XDFM_REQUIRE(attn_fwd_lds(S, D) <= 160 * 1024, "cin_attn_pool_fwd: S=%d D=%d does not fit LDS", S, D);
    AttnDrop drop;
    if (int rc = attn_drop_args(p_drop, drop_seed, &drop, "cin_attn_pool_fwd")) return rc;
    hipStream_t st = (hipStream_t)stream;
    ATTN_DISPATCH(launch_attn_fwd, fm, B, S, n_layers, use_ln, use_res, theta, out, tok_save, o_save, ml_save, drop, st)
}

int xdfm_cin_attn_pool_bwd(const float* fm, int B, int S, int D, int nh, int n_layers, int use_ln, int use_res,
                           const float* theta, const float* tok_save, const float* o_save, const float* ml_save,
                           const float* dout, float* dfm, float* dtheta, float p_drop,
                           const unsigned long long* drop_seed, void* stream) {
    XDFM_REQUIRE(fm && theta && tok_save && o_save && ml_save && dout && dfm && dtheta,
                 "cin_attn_pool_bwd: null pointer");
    XDFM_REQUIRE(B > 0 && S > 0 && S <= 1024 && n_layers >= 1, "cin_attn_pool_bwd: bad shape B=%d S=%d layers=%d", B, S,
                 n_layers);
    AttnDrop drop;
    if (int rc = attn_drop_args(p_drop, drop_seed, &drop, "cin_attn_pool_bwd")) return rc;
    hipStream_t st = (hipStream_t)stream;
    ATTN_DISPATCH(launch_attn_bwd, fm, B, S, n_layers, use_ln, use_res, theta, tok_save, o_save, ml_save, dout, dfm, dtheta,
                  (float*)nullptr, drop, st)
}

size_t xdfm_cin_attn_pool_bwd_ws_elems(int B, int D, int n_layers, int use_ln) {
    if (B <= 0 || D <= 0 || n_layers <= 0) return 0;
    return (size_t)attn_bwd_groups(B) * xdfm_cin_attn_theta_elems(D, n_layers, use_ln);
}

int xdfm_cin_attn_pool_bwd_det(const float* fm, int B, int S, int D, int nh, int n_layers, int use_ln, int use_res,
                               const float* theta, const float* tok_save, const float* o_save, const float* ml_save,
                               const float* dout, float* dfm, float* dtheta, float* ws, float p_drop,
                               const unsigned long long* drop_seed, void* stream) {
    XDFM_REQUIRE(fm && theta && tok_save && o_save && ml_save && dout && dfm && dtheta && ws,
                 "cin_attn_pool_bwd: null pointer");
    XDFM_REQUIRE(B > 0 && S > 0 && S <= 1024 && n_layers >= 1, "cin_attn_pool_bwd: bad shape B=%d S=%d layers=%d", B, S,
                 n_layers);
    AttnDrop drop;
    if (int rc = attn_drop_args(p_drop, drop_seed, &drop, "cin_attn_pool_bwd")) return rc;
    hipStream_t st = (hipStream_t)stream;
    ATTN_DISPATCH(launch_attn_bwd, fm, B, S, n_layers, use_ln, use_res, theta, tok_save, o_save, ml_save, dout, dfm, dtheta,
                  ws, drop, st)
}

int xdfm_cin_attn_dropout_mask(int B, int S, int nh, int n_layers, float p_drop, const unsigned long long* drop_seed,
                               unsigned char* keep, void* stream) {
    XDFM_REQUIRE(keep && drop_seed, "cin_attn_dropout_mask: null pointer");
    XDFM_REQUIRE(B > 0 && S > 0 && nh > 0 && n_layers > 0, "cin_attn_dropout_mask: bad shape");
    AttnDrop drop;
    if (int rc = attn_drop_args(p_drop, drop_seed, &drop, "cin_attn_dropout_mask")) return rc;
    XDFM_REQUIRE(drop.seed, "cin_attn_dropout_mask: p_drop must be > 0");
    const long rows = (long)n_layers * B * nh * S;
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       B, S, nh, n_layers, drop, keep);
    return xdfm_check_launch("cin_attn_dropout_mask");
}

}  // extern "C"
