// Arithmetic of one Adam update (K7 / K7d), in two spellings that give THE SAME BITS:
//
//   adam_one        the reference spelling: ATen's fused-Adam sequence with IEEE sqrt and two IEEE divisions, as hipcc
//                   expands them (46 VALU instructions per element, three of them quarter-rate transcendentals).  The
//                   dense sweep is HBM-bound and uses it as it is.
//   adam_replay4    one step of a 16-byte chunk NO batch touched (gradient = the L2 term alone), as the deferred update
//                   replays it -- ALU-bound, 575 M elements x every step at the Criteo-card vocabulary.  It computes the
//                   same correctly rounded results with about half the issue slots: two elements per packed instruction
//                   (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), the square root from ONE v_rsq_f32 and a coupled
//                   Newton step whose last correction rounds correctly (checked for every fp32 input, see below), the
//                   division by the step's constant sqrt(1 - beta2^t) with its exactly rounded reciprocal from the
//                   step's constant table (Markstein's correction: no transcendental, no scaling), and the general
//                   division as hipcc's own Newton chain WITHOUT v_div_scale / v_div_fmas / v_div_fixup.  Those three
//                   only act when an operand or the quotient leaves the normal range; a guard on the step's new moments
//                   (0 < v' < 2^62, 2^-50 <= |m'| < 2^30; the step's constants are range-checked by adam_tick) proves
//                   that they would not have acted, and a wave in which any lane fails the guard takes the reference
//                   spelling for that step.  So: same fused multiply-adds on the same operands -> the same bits.
//
// Verification (xdfm_adam_selftest, tests/test_gpu_host.py::test_fast_adam_replay_primitives_are_correctly_rounded):
// the square root against sqrtf for ALL 2^32 bit patterns (through the 2^64 pre-scaling the replay uses); the division by
// a constant against IEEE division for every fp32 numerator in two binades x the constants of 300 steps; the general
// division against IEEE division on 2^32 random operand pairs inside the guard plus the guard's corners; whole replayed
// steps against adam_one on random states incl. zeros, denormals and huge values (the guard's fall-back).
#pragma once
#include <hip/hip_runtime.h>

typedef float adam_f2 __attribute__((ext_vector_type(2)));

struct AdamCoef { float w1, b2, w2, lr, eps; };

// The fusions are spelled out and the compiler's own contraction is off: left to itself it fuses differently in
// the marked and the dense loop, and the two must give the same bits.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float bc2_sqrt, const AdamCoef& c) {
#pragma clang fp contract(off)
    m = fmaf(c.w1, g - m, m);
    v = fmaf(c.w2 * g, g, c.b2 * v);
    const float denom = sqrtf(v) / bc2_sqrt + c.eps;
    p -= step_size * m / denom;
}

// Constants of one replayed step (wave-uniform; written by adam_tick_kernel into the clock's table, 4 floats per step):
//   ss   = lr / (1 - beta1^t)            bc  = sqrt(1 - beta2^t)
//   c2   = bc * 2^32                     rc2 = RN(1 / bc) * 2^-32, or 0 when the step must take the reference spelling
//                                              (a constant outside the range the guard's proof assumes)
struct AdamStepConst { float ss, bc, c2, rc2; };
#define ADAM_CONSTS_PER_STEP 4

__device__ __forceinline__ adam_f2 adam_pkfma(adam_f2 a, adam_f2 b, adam_f2 c) { return __builtin_elementwise_fma(a, b, c); }

// RN(sqrt(x)) * 2^32 for x2 = x * 2^64 (normal, > 0): one v_rsq_f32 (1 ulp), coupled Newton step on (s, h ~ 1 / (2 s)),
// exact residual, final correction (Markstein).  Correct rounding checked exhaustively (xdfm_adam_selftest mode 0).
__device__ __forceinline__ adam_f2 adam_sqrt_scaled2(adam_f2 x2) {
#pragma clang fp contract(off)
    adam_f2 y;
    y.x = __builtin_amdgcn_rsqf(x2.x);
    y.y = __builtin_amdgcn_rsqf(x2.y);
    const adam_f2 half = {0.5f, 0.5f};
    const adam_f2 s0 = x2 * y, h0 = half * y;
    const adam_f2 r = adam_pkfma(-s0, h0, half);
    const adam_f2 s1 = adam_pkfma(s0, r, s0), h1 = adam_pkfma(h0, r, h0);
    const adam_f2 dd = adam_pkfma(-s1, s1, x2);
    return adam_pkfma(dd, h1, s1);
}

// RN(S / bc) from s = S * 2^32, c2 = bc * 2^32, rc2 = RN(1 / bc) * 2^-32: quotient estimate and two corrections with the
// exactly rounded reciprocal (the second one is Markstein's: a faithful quotient + exact residual * RN(1/b) rounds
// correctly).  All scalings are powers of two, so every rounding is the rounding of the unscaled quantity.
__device__ __forceinline__ adam_f2 adam_div_const2(adam_f2 s, float c2, float rc2) {
#pragma clang fp contract(off)
    const adam_f2 C = {c2, c2}, R = {rc2, rc2};
    const adam_f2 t0 = s * R;
    const adam_f2 r0 = adam_pkfma(-t0, C, s);
    const adam_f2 t1 = adam_pkfma(r0, R, t0);
    const adam_f2 r1 = adam_pkfma(-t1, C, s);
    return adam_pkfma(r1, R, t1);
}

// RN(n / d): hipcc's expansion of an fp32 division on gfx950 (v_rcp_f32, one Newton step on the reciprocal, quotient, two
// residual corrections) without v_div_scale_f32 / v_div_fmas_f32 / v_div_fixup_f32 -- identical bits whenever those would
// not have rescaled: n == 0, or 2^-103 <= |n|, d and 1/d normal, n/d normal, exponent(n) - exponent(d) < 96.
__device__ __forceinline__ adam_f2 adam_div2(adam_f2 n, adam_f2 d) {
#pragma clang fp contract(off)
    adam_f2 y0;
    y0.x = __builtin_amdgcn_rcpf(d.x);
    y0.y = __builtin_amdgcn_rcpf(d.y);
    const adam_f2 one = {1.f, 1.f};
    const adam_f2 e = adam_pkfma(-d, y0, one);
    const adam_f2 y1 = adam_pkfma(e, y0, y0);
    const adam_f2 q0 = n * y1;
    const adam_f2 r0 = adam_pkfma(-d, q0, n);
    const adam_f2 q1 = adam_pkfma(r0, y1, q0);
    const adam_f2 r1 = adam_pkfma(-d, q1, n);
    return adam_pkfma(r1, y1, q1);
}

#define ADAM_V_MAX 4.611686018427388e18f      // 2^62
#define ADAM_M_MIN 8.881784197001252e-16f     // 2^-50
#define ADAM_M_MAX 1073741824.0f              // 2^30
#define ADAM_EPS_MIN 9.094947017729282e-13f   // 2^-40: the guard's proof needs d = t + eps >= 2^-40 (and eps <= 1)

// One missed step of a chunk no batch touched: the gradient is the L2 term alone.  Spelled exactly like the sweep's
// update of an unmarked chunk (gradient = fmaf(2*l2, w, opaque zero)), so a replayed step gives the sweep's bits.
// `sq2` collects the L2 VALUE of the replayed steps (x^2 + z^2, y^2 + w^2 per lane half; its total feeds a fixed-point
// accumulator that is compared at summation-noise tolerance, so its association is free).
__device__ __forceinline__ void adam_replay4(float4& p, float4& m, float4& v, const AdamStepConst& k, float g2, float zf,
                                             const AdamCoef& c, adam_f2& sq2, bool eps_ok) {
#pragma clang fp contract(off)
    adam_f2 pa = {p.x, p.y}, pb = {p.z, p.w};
    sq2 += adam_pkfma(pb, pb, pa * pa);
    const adam_f2 G2 = {g2, g2}, Z = {zf, zf}, W1 = {c.w1, c.w1}, W2 = {c.w2, c.w2}, B2 = {c.b2, c.b2};
    const adam_f2 ga = adam_pkfma(G2, pa, Z), gb = adam_pkfma(G2, pb, Z);
    adam_f2 ma = {m.x, m.y}, mb = {m.z, m.w}, va = {v.x, v.y}, vb = {v.z, v.w};
    const adam_f2 ma1 = adam_pkfma(W1, ga - ma, ma), mb1 = adam_pkfma(W1, gb - mb, mb);
    const adam_f2 va1 = adam_pkfma(W2 * ga, ga, B2 * va), vb1 = adam_pkfma(W2 * gb, gb, B2 * vb);
    // guard (see the header comment): every element's new moments inside the range for which the short forms are proven
    const float vmin = fminf(fminf(va1.x, va1.y), fminf(vb1.x, vb1.y)), vmax = fmaxf(fmaxf(va1.x, va1.y), fmaxf(vb1.x, vb1.y));
    const float mmin = fminf(fminf(fabsf(ma1.x), fabsf(ma1.y)), fminf(fabsf(mb1.x), fabsf(mb1.y)));
    const float mmax = fmaxf(fmaxf(fabsf(ma1.x), fabsf(ma1.y)), fmaxf(fabsf(mb1.x), fabsf(mb1.y)));
    const bool fast = vmin > 0.f && vmax < ADAM_V_MAX && mmin >= ADAM_M_MIN && mmax < ADAM_M_MAX;   // false for NaN too
    if (eps_ok && k.rc2 != 0.f && __builtin_amdgcn_ballot_w64(!fast) == 0) {
        const adam_f2 big = {18446744073709551616.f, 18446744073709551616.f};      // 2^64
        const adam_f2 E = {c.eps, c.eps}, SS = {k.ss, k.ss};
        const adam_f2 sa = adam_sqrt_scaled2(va1 * big), sb = adam_sqrt_scaled2(vb1 * big);
        const adam_f2 da = adam_div_const2(sa, k.c2, k.rc2) + E, db = adam_div_const2(sb, k.c2, k.rc2) + E;
        const adam_f2 neg1 = {-1.f, -1.f};                  // p - q as fma(q, -1, p): the same single rounding, one packed instruction
        pa = adam_pkfma(adam_div2(SS * ma1, da), neg1, pa);
        pb = adam_pkfma(adam_div2(SS * mb1, db), neg1, pb);
        p = make_float4(pa.x, pa.y, pb.x, pb.y);
        m = make_float4(ma1.x, ma1.y, mb1.x, mb1.y);
        v = make_float4(va1.x, va1.y, vb1.x, vb1.y);
    } else {
        adam_one(p.x, ga.x, m.x, v.x, k.ss, k.bc, c); adam_one(p.y, ga.y, m.y, v.y, k.ss, k.bc, c);
        adam_one(p.z, gb.x, m.z, v.z, k.ss, k.bc, c); adam_one(p.w, gb.y, m.w, v.w, k.ss, k.bc, c);
    }
}

__device__ __forceinline__ AdamStepConst adam_step_const(const float* __restrict__ consts, int s) {
    const float4 t = *reinterpret_cast<const float4*>(consts + ADAM_CONSTS_PER_STEP * s);
    return AdamStepConst{t.x, t.y, t.z, t.w};
}

// Steps old + 1 .. t_end of this lane's chunk; lanes with active == false take no part.  EVERY lane of the wave must reach
// the call (it starts with a wave reduction): the lanes are lined up on the step number, so that all of them replay the
// same step in the same iteration and the step's constants are four scalar registers instead of a per-lane load.
__device__ __forceinline__ void adam_replay_span(float4& p, float4& m, float4& v, bool active, int old, int t_end,
                                                 const float* __restrict__ consts, float g2, float zf, const AdamCoef& c,
                                                 adam_f2& sq2, bool eps_ok) {
    int mn = active ? old : 0x7fffffff;
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(mn, o);
        mn = other < mn ? other : mn;
    }
    const int first = __builtin_amdgcn_readfirstlane(mn);
    if (first >= t_end) return;                          // also: no active lane
    for (int s = first + 1; s <= t_end; ++s) {
        const AdamStepConst k = adam_step_const(consts, s);
        if (active && s > old) adam_replay4(p, m, v, k, g2, zf, c, sq2, eps_ok);
    }
}

// Exactly rounded reciprocal of a normal fp32 c (host or device, one thread): the double quotient rounded to fp32 is
// RN(1/c) unless it sits within double rounding of a midpoint, so the candidate and its neighbours are compared by their
// exact residuals 1 - c*y (48-bit products: exact in double).
__host__ __device__ inline float adam_exact_rcp(float c) {
    float y = (float)(1.0 / (double)c);
    double best = fabs(1.0 - (double)c * (double)y);
    const float cand[2] = {nextafterf(y, 0.f), nextafterf(y, 3.0e38f)};
    float out = y;
    for (int k = 0; k < 2; ++k) {
        const double r = fabs(1.0 - (double)c * (double)cand[k]);
        if (r < best) { best = r; out = cand[k]; }
    }
    return out;
}
