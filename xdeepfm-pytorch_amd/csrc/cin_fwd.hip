// K3: one CIN level, forward.  out[h][n] = act( sum_{i,j} W[h][i*m+j] * xp[i][n] * x0[j][n] + bias[h] )
//
// replaces deepctr/layers/interaction.py:218-229 (einsum outer product -> reshape -> Conv1d(k=1)
// -> activation).  The outer product Z is never stored: each lane forms its Z element with one
// v_mul and feeds it to v_mfma_f32_32x32x2_f32 as the B operand.
//
// GEMM view:  Out[H x N] = W[H x K] * Z[K x N],  K = Hp*m pairs (i,j),  N = B*D columns.
//   A operand = W, pre-packed by cin_fwd_pack_kernel into fragment order so one coalesced
//               dwordx4 (MT floats) per lane per k-step delivers the wave's MT row tiles;
//   B operand = Z, lane (c = lane&31, s = lane>>5) of column fragment f holds
//               Z[k = (i, 2u+s)][n0 + 32f + c] = xp[i][n] * x0[2u+s][n]   for k-step t = i*MP + u.
// A wave owns 32*NF columns and 32*MT output rows; it keeps x0 for its columns in a wave-private
// LDS slice (no barriers anywhere in the kernel) and streams xp[i] / packed W from L2.
#include "xdfm_internal.h"

// ---------------------------------------------------------------------------------------------
// Wf[mb][t][lane][MT], t in [0, Tpad + 4): rows mb*32*MT + mt*32 + (lane&31), k = (t / MP, 2*(t % MP) + (lane>>5))
template <int MT>
__global__ void cin_fwd_pack_kernel(const float* __restrict__ W, int H, int Hp, int m, int MP,
                                    long TP, long total, float* __restrict__ Wf) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int mt = (int)(idx % MT);
    long r1 = idx / MT;
    int lane = (int)(r1 & 63);
    long r2 = r1 >> 6;
    long t = r2 % TP;
    int mb = (int)(r2 / TP);
    int row = mb * 32 * MT + mt * 32 + (lane & 31);
    long i = t / MP;
    int j = 2 * (int)(t - i * MP) + (lane >> 5);
    float v = 0.f;
    if (row < H && i < Hp && j < m) v = W[(long)row * ((long)Hp * m) + i * m + j];
    Wf[idx] = v;
}

template <int MT>
__device__ __forceinline__ void load_afrag(const float* __restrict__ p, float (&a)[MT]) {
    if constexpr (MT == 1) {
        a[0] = p[0];
    } else if constexpr (MT == 2) {
        float2 v = *reinterpret_cast<const float2*>(p);
        a[0] = v.x; a[1] = v.y;
    } else if constexpr (MT == 4) {
        float4 v = *reinterpret_cast<const float4*>(p);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    } else {
        float4 v = *reinterpret_cast<const float4*>(p);
        float4 w = *reinterpret_cast<const float4*>(p + 4);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
        a[4] = w.x; a[5] = w.y; a[6] = w.z; a[7] = w.w;
    }
}

#define FWD_IC 32   // x_prev rows staged per refill of the wave-private LDS chunk

template <int MT, int NF>
__global__ __launch_bounds__(256, 2) void cin_fwd_kernel(
    const float* __restrict__ xp, const float* __restrict__ x0, const float* __restrict__ Wf,
    const float* __restrict__ bias, int H, int Hp, int m, long N, int TP, int Tpad, int T,
    int act, float* __restrict__ out, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & 31, s = lane >> 5;
    const int MP = (m + 1) >> 1;
    const long n0 = ((long)blockIdx.x * 4 + wave) * (32 * NF);
    if (n0 >= N) return;                       // no barriers below: a whole wave may leave
    const int mb = blockIdx.y;
    constexpr int WC = 32 * NF;                // columns per wave
    const int tmask = (dbg & 1) ? 7 : 0x7fffffff;

    // wave-private LDS: x0s[jj][cc] (jj < 2*MP, zero padded) and a chunk of FWD_IC rows of x_prev
    float* x0s = smem + wave * ((2 * MP + FWD_IC) * WC);
    float* xps = x0s + 2 * MP * WC;
    for (int idx = lane; idx < 2 * MP * WC; idx += 64) {
        int jj = idx / WC, cc = idx - jj * WC;
        long n = n0 + cc;
        const float v = x0[(long)(jj < m ? jj : m - 1) * N + (n < N ? n : N - 1)];
        x0s[idx] = v * ((jj < m && n < N) ? 1.f : 0.f);
    }
    // refill: FWD_IC rows x WC columns, RP rows per 64-lane pass; all loads are issued before the
    // first LDS write so they overlap (clamped addresses, no per-element branch).
    constexpr int RP = 64 / WC;
    const int lr = lane / WC;
    const long ncs = (n0 + (lane % WC) < N) ? n0 + (lane % WC) : N - 1;
    auto stage_xp = [&](int i0) {
        float tmp[FWD_IC / RP];
#pragma unroll
        for (int k = 0; k < FWD_IC / RP; ++k) {
            int row = i0 + k * RP + lr;
            row = row < Hp ? row : Hp - 1;   // rows >= Hp feed only padded k-steps (b forced to 0)
            tmp[k] = xp[(long)row * N + ncs];
        }
#pragma unroll
        for (int k = 0; k < FWD_IC / RP; ++k) xps[k * 64 + lane] = tmp[k];
    };
    stage_xp(0);

    f32x16 acc[MT][NF];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][f][r] = 0.f;

    const float* wp = Wf + ((long)mb * TP * 64 + lane) * MT;   // k-step t lives at wp + t*64*MT
    float a[4][MT];
    load_afrag<MT>(wp, a[0]);
    load_afrag<MT>(wp + 64 * MT, a[1]);
    load_afrag<MT>(wp + 2 * 64 * MT, a[2]);

    int i = 0, u = 0, il = 0;                  // k-step t = i*MP + u;  il = i % FWD_IC
    const float* x0l = x0s + s * WC + c;       // + u*2*WC + f*32
    const float* xpl = xps + c;                // + il*WC + f*32
    float bc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) bc[f] = xpl[f * 32] * x0l[f * 32];

    for (int t = 0; t < Tpad; t += 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            load_afrag<MT>(wp + (long)((t + e + FWD_PD) & tmask) * (64 * MT), a[(e + FWD_PD) & 3]);
            // advance to k-step t+e+1 and start its operand reads (consumed after this step's MFMAs)
            if (++u == MP) {
                u = 0;
                ++i;
                if (++il == FWD_IC) {
                    il = 0;
                    stage_xp(i);
                }
            }
            float xr[NF], zr[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                xr[f] = xpl[il * WC + f * 32];
                zr[f] = x0l[u * 2 * WC + f * 32];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    acc[mt][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e][mt], bc[f], acc[mt][f], 0, 0, 0);
            const bool live = (t + e + 1) < T;   // wave-uniform; padded k-steps contribute exactly 0
#pragma unroll
            for (int f = 0; f < NF; ++f) bc[f] = live ? xr[f] * zr[f] : 0.f;
        }
    }

    // epilogue: bias + activation, FM-layout store (each register: two 128-B row segments)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mb * 32 * MT + mt * 32 + frag_row(r, s);
            if (row < H) {
                const float bv = bias[row];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const long n = n0 + f * 32 + c;
                    if (n < N) {
                        float v = acc[mt][f][r] + bv;
                        if (act == XDFM_ACT_RELU) v = fmaxf(v, 0.f);
                        out[(long)row * N + n] = v;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Specialisation for a compile-time number of j-pairs per i (MPT = ceil(m/2): 13 for the 26 Criteo
// fields, 11 for the 22 Avazu fields).  The k-steps of one i are straight-line code: x0 lives in
// registers, the A ring uses static slots and the only branch is the loop over i.  (The generic
// kernel above pays ~3 taken branches and an LDS round trip per k-step, which caps it near 70 %
// MFMA utilisation.)
template <int MPT>
struct RingSize {   // smallest R >= 4 such that slot(u) = u % R never collides across the wrap of i
    static constexpr int pick() {
        for (int r = 4; r < 64; ++r) {
            const int rem = MPT % r;
            if (rem == 0 || rem >= 4) return r;
        }
        return MPT;
    }
    static constexpr int value = pick();
};

template <int MT, int NF, int MPT>
__global__ __launch_bounds__(256, 2) void cin_fwd_mp_kernel(
    const float* __restrict__ xp, const float* __restrict__ x0, const float* __restrict__ Wf,
    const float* __restrict__ bias, int H, int Hp, int m, long N, int TP, int act, float* __restrict__ out,
    int dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & 31, s = lane >> 5;
    const long n0 = ((long)blockIdx.x * 4 + wave) * (32 * NF);
    if (n0 >= N) return;
    const int mb = blockIdx.y;
    constexpr int WC = 32 * NF;
    constexpr int R = RingSize<MPT>::value;
    const int wstep = (dbg & 1) ? 0 : MPT * 64 * MT;     // timing experiment: re-read one cached slab
    constexpr int RP = 64 / WC;

    float* xps = smem + wave * (FWD_IC * WC);
    const int lr = lane / WC;
    const long ncs = (n0 + (lane % WC) < N) ? n0 + (lane % WC) : N - 1;
    auto stage_xp = [&](int i0) {
        float tmp[FWD_IC / RP];
#pragma unroll
        for (int k = 0; k < FWD_IC / RP; ++k) {
            int row = i0 + k * RP + lr;
            row = row < Hp ? row : Hp - 1;
            tmp[k] = xp[(long)row * N + ncs];
        }
#pragma unroll
        for (int k = 0; k < FWD_IC / RP; ++k) xps[k * 64 + lane] = tmp[k];
    };
    stage_xp(0);

    // x0 for this lane's columns: x0r[f][u] = x0[2u+s][n0 + 32f + c]
    float x0r[NF][MPT];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const long n = n0 + f * 32 + c;
        const long ncl = n < N ? n : N - 1;
#pragma unroll
        for (int u = 0; u < MPT; ++u) {
            const int j = 2 * u + s;
            x0r[f][u] = x0[(long)(j < m ? j : m - 1) * N + ncl] * ((j < m && n < N) ? 1.f : 0.f);
        }
    }

    f32x16 acc[MT][NF];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][f][r] = 0.f;

    const float* wp = Wf + ((long)mb * TP * 64 + lane) * MT;
    float a[R][MT];
#pragma unroll
    for (int k = 0; k < FWD_PD; ++k) load_afrag<MT>(wp + k * 64 * MT, a[k % R]);

    // two-level loop so that the hot loop over i has a single path (a conditional refill inside it
    // makes hipcc drain vmcnt(0) at every i)
    for (int i0 = 0; i0 < Hp; i0 += FWD_IC) {
        if (i0 > 0) stage_xp(i0);
        const int cnt = (Hp - i0 < FWD_IC) ? Hp - i0 : FWD_IC;
        const float* wpi = wp + (long)i0 * (MPT * 64 * MT);
        const float* xl = xps + c;
        float xv[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) xv[f] = xl[f * 32];
        float b[NF];                                     // B operand of the NEXT k-step, formed one step early
#pragma unroll
        for (int f = 0; f < NF; ++f) b[f] = xv[f] * x0r[f][0];
        // One i = MPT straight-line k-steps.  hipcc drains vmcnt(0) at a loop header whose back edge
        // carries in-flight loads (one L2 round trip per iteration), so the loop over i is unrolled by
        // hand to make iterations 4 x MPT k-steps long.
        auto one_i = [&](int il) {
            float xn[NF];                                // x_prev of the next i (clamped inside the chunk)
            const int iln = (il + 1 < cnt) ? il + 1 : il;
#pragma unroll
            for (int f = 0; f < NF; ++f) xn[f] = xps[iln * WC + f * 32 + c];
#pragma unroll
            for (int u = 0; u < MPT; ++u) {
                load_afrag<MT>(wpi + (u + FWD_PD) * (64 * MT), a[(u + FWD_PD) % MPT % R]);
                float bcur[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) bcur[f] = b[f];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        acc[mt][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u % R][mt], bcur[f], acc[mt][f], 0, 0, 0);
                    if (mt == 0) {
                        // one v_mul in the shadow of the first MFMA: the next step's operand is ready
                        // long before its MFMAs issue (no VALU -> MFMA read stall at the step boundary)
#pragma unroll
                        for (int f = 0; f < NF; ++f)
                            b[f] = (u + 1 < MPT) ? xv[f] * x0r[f][(u + 1) % MPT] : xn[f] * x0r[f][0];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch distance hipcc would otherwise collapse
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) xv[f] = xn[f];
            wpi += wstep;
        };
        int il = 0;
        for (; il + 4 <= cnt; il += 4) {
            one_i(il);
            one_i(il + 1);
            one_i(il + 2);
            one_i(il + 3);
        }
        for (; il < cnt; ++il) one_i(il);
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mb * 32 * MT + mt * 32 + frag_row(r, s);
            if (row < H) {
                const float bv = bias[row];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const long n = n0 + f * 32 + c;
                    if (n < N) {
                        float v = acc[mt][f][r] + bv;
                        if (act == XDFM_ACT_RELU) v = fmaxf(v, 0.f);
                        out[(long)row * N + n] = v;
                    }
                }
            }
        }
    }
}

// sum over d of `rows` feature maps: res[b*ldres + off + r] = sum_d A[row0 + r][b*D + d]
// POW2: D is a power of two <= 64 -> coalesced loads + xor-shuffle reduction inside D-lane groups.
// D a power of two in [4, 64] and 16-byte aligned rows: one float4 per lane (D/4 lanes per example), xor-shuffle
// over those lanes; 1 KB per wave-instruction instead of 256 B.
__global__ __launch_bounds__(256) void cin_direct_sum_vec_kernel(const float* __restrict__ A, int row0, long N, int D,
                                                                float* __restrict__ res, long ldres, int off) {
    const int r = blockIdx.y;
    const long n = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    float v = 0.f;
    if (n < N) {
        const float4 a = *reinterpret_cast<const float4*>(A + (long)(row0 + r) * N + n);
        v = (a.x + a.y) + (a.z + a.w);
    }
    for (int o = D >> 3; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (n < N && (n & (D - 1)) == 0) res[(n / D) * ldres + off + r] = v;
}

template <bool POW2>
__global__ void cin_direct_sum_kernel(const float* __restrict__ A, int row0, int rows, int B, int D,
                                      float* __restrict__ res, long ldres, int off) {
    const int r = blockIdx.y;
    if constexpr (POW2) {
        const long N = (long)B * D;
        const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
        float v = n < N ? A[(long)(row0 + r) * N + n] : 0.f;
        for (int o = D >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (n < N && (n & (D - 1)) == 0) res[(n / D) * ldres + off + r] = v;
    } else {
        const int b = blockIdx.x * blockDim.x + threadIdx.x;
        if (b >= B) return;
        const float* p = A + ((long)(row0 + r) * B + b) * D;
        float sacc = 0.f;
        for (int d = 0; d < D; ++d) sacc += p[d];
        res[(long)b * ldres + off + r] = sacc;
    }
}

// ---------------------------------------------------------------------------------------------
template <int MT>
static int launch_pack(const float* W, int H, int Hp, int m, float* Wf, hipStream_t st) {
    const int MP = fwd_mp(m);
    const long TP = fwd_tpad(Hp, m) + 4;
    const int MB = ceil_div(H, 32 * MT);
    const long total = (long)MB * TP * 64 * MT;
    hipLaunchKernelGGL((cin_fwd_pack_kernel<MT>), dim3(ceil_div(total, 256)), dim3(256), 0, st, W, H, Hp, m, MP,
                       TP, total, Wf);
    return xdfm_check_launch("cin_fwd_pack");
}

template <int MT, int NF, int MPT>
static int launch_fwd_mp(const float* xp, const float* x0, const float* Wf, const float* bias, int H, int Hp,
                         int m, long N, int act, float* out, hipStream_t st) {
    const int TP = (int)fwd_tpad(Hp, m) + 4;
    const int MB = ceil_div(H, 32 * MT);
    size_t lds = (size_t)4 * FWD_IC * 32 * NF * sizeof(float);
    if (xdfm_opt(OPT_DBG) & 4) lds = 80 * 1024;
    dim3 grid(ceil_div(N, 128L * NF), MB);
    hipLaunchKernelGGL((cin_fwd_mp_kernel<MT, NF, MPT>), grid, dim3(256), lds, st, xp, x0, Wf, bias, H, Hp, m, N, TP,
                       act, out, xdfm_opt(OPT_DBG));
    return xdfm_check_launch("cin_level_fwd");
}

template <int MT, int NF>
static int launch_fwd(const float* xp, const float* x0, const float* Wf, const float* bias, int H, int Hp,
                      int m, long N, int act, float* out, hipStream_t st) {
    const int MP = fwd_mp(m);
    const long Tpad_l = fwd_tpad(Hp, m);
    if (Tpad_l + 8 > 0x7fffffffL / (64 * 8)) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_fwd: Hp*m too large");
    if constexpr (MT >= 4) {
        // straight-line specialisations for the field counts of the benchmark configs
        if (!(xdfm_opt(OPT_DBG) & 2)) {
            if (MP == 13) return launch_fwd_mp<MT, NF, 13>(xp, x0, Wf, bias, H, Hp, m, N, act, out, st);
            if (MP == 11) return launch_fwd_mp<MT, NF, 11>(xp, x0, Wf, bias, H, Hp, m, N, act, out, st);
        }
    }
    const int Tpad = (int)Tpad_l;
    const int TP = Tpad + 4;
    const int T = Hp * MP;
    const int MB = ceil_div(H, 32 * MT);
    const size_t lds = (size_t)4 * (2 * MP + FWD_IC) * 32 * NF * sizeof(float);
    if (lds > 160 * 1024) return xdfm_fail(XDFM_ERR_INVALID, "cin_level_fwd: m=%d needs %zu B of LDS", m, lds);
    dim3 grid(ceil_div(N, 128L * NF), MB);
    hipLaunchKernelGGL((cin_fwd_kernel<MT, NF>), grid, dim3(256), lds, st, xp, x0, Wf, bias, H, Hp, m, N, TP,
                       Tpad, T, act, out, xdfm_opt(OPT_DBG));
    return xdfm_check_launch("cin_level_fwd");
}

extern "C" {

size_t xdfm_cin_fwd_pack_elems(int H, int Hp, int m) {
    if (H <= 0 || Hp <= 0 || m <= 0) return 0;
    if (x3_fwd_usable(H, Hp, m)) return x3_fwd_pack_elems(H, Hp, m);
    const int MT = fwd_mt(H);
    return (size_t)ceil_div(H, 32 * MT) * (size_t)(fwd_tpad(Hp, m) + 4) * 64 * MT;
}

int xdfm_cin_fwd_pack(const float* W, int H, int Hp, int m, float* Wf, void* stream) {
    XDFM_REQUIRE(W && Wf, "cin_fwd_pack: null pointer");
    XDFM_REQUIRE(H > 0 && Hp > 0 && m > 0, "cin_fwd_pack: bad shape H=%d Hp=%d m=%d", H, Hp, m);
    hipStream_t st = (hipStream_t)stream;
    if (x3_fwd_usable(H, Hp, m)) return x3_fwd_pack(W, H, Hp, m, Wf, st);
    switch (fwd_mt(H)) {
        case 1: return launch_pack<1>(W, H, Hp, m, Wf, st);
        case 2: return launch_pack<2>(W, H, Hp, m, Wf, st);
        case 4: return launch_pack<4>(W, H, Hp, m, Wf, st);
        default: return launch_pack<8>(W, H, Hp, m, Wf, st);
    }
}

int xdfm_cin_pack_all_supported(int H, int Hp, int m) {
    return (H > 0 && Hp > 0 && m > 0 && x3_pack_all_usable(H, Hp, m)) ? 1 : 0;
}

int xdfm_cin_pack_all(const xdfm_cin_pack_job* jobs, int L, void* stream) {
    XDFM_REQUIRE(jobs && L > 0 && L <= 8, "cin_pack_all: 1..8 jobs");
    for (int l = 0; l < L; ++l) {
        XDFM_REQUIRE(jobs[l].W && (jobs[l].fwd_pack || jobs[l].bwd_pack), "cin_pack_all: job %d has no weight / output", l);
        XDFM_REQUIRE(xdfm_cin_pack_all_supported(jobs[l].H, jobs[l].Hp, jobs[l].m),
                     "cin_pack_all: level %d (H=%d Hp=%d m=%d) has no f16x3 kernels in both directions", l, jobs[l].H,
                     jobs[l].Hp, jobs[l].m);
        XDFM_REQUIRE(((((size_t)jobs[l].fwd_pack) | ((size_t)jobs[l].bwd_pack)) & 15) == 0, "cin_pack_all: packs must be 16-byte aligned");
    }
    return x3_pack_all(jobs, L, (hipStream_t)stream);
}

int xdfm_cin_level_fwd(const float* xp, const float* x0, const float* Wf, const float* bias, int H, int Hp,
                       int m, long N, int act, float* out, void* stream) {
    XDFM_REQUIRE(xp && x0 && Wf && bias && out, "cin_level_fwd: null pointer");
    XDFM_REQUIRE(H > 0 && Hp > 0 && m > 0 && N > 0, "cin_level_fwd: bad shape H=%d Hp=%d m=%d N=%ld", H, Hp, m, N);
    XDFM_REQUIRE(act == XDFM_ACT_LINEAR || act == XDFM_ACT_RELU, "cin_level_fwd: unsupported activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    if (x3_fwd_usable(H, Hp, m)) { xdfm_opt_note(OPT_LAST_FWD, xdfm_opt(OPT_CIN_MATH)); return x3_level_fwd(xp, x0, Wf, bias, H, Hp, m, N, act, out, x3_fwd_epi_plain(H), st); }
    xdfm_opt_note(OPT_LAST_FWD, 0);
    xdfm_opt_note(OPT_LAST_SYM, xdfm_opt(OPT_LAST_SYM) & ~1);
    const int nf = xdfm_opt(OPT_FWD_NF) == 2 ? 2 : 1;
    switch (fwd_mt(H)) {
        case 1: return launch_fwd<1, 1>(xp, x0, Wf, bias, H, Hp, m, N, act, out, st);
        case 2: return launch_fwd<2, 1>(xp, x0, Wf, bias, H, Hp, m, N, act, out, st);
        case 4:
            return nf == 2 ? launch_fwd<4, 2>(xp, x0, Wf, bias, H, Hp, m, N, act, out, st)
                           : launch_fwd<4, 1>(xp, x0, Wf, bias, H, Hp, m, N, act, out, st);
        default: return launch_fwd<8, 1>(xp, x0, Wf, bias, H, Hp, m, N, act, out, st);
    }
}

int xdfm_cin_level_fwd_ex_supported(int H, int Hp, int m, int D) {
    return (H > 0 && Hp > 0 && m > 0 && (D == 4 || D == 8 || D == 16 || D == 32) && x3_fwd_usable(H, Hp, m)) ? 1 : 0;
}

int xdfm_cin_level_fwd_ex(const float* xp, const float* x0, const float* Wf, const float* bias, int H, int Hp, int m, long N,
                          int act, float* out, int keep_rows, float* res, long ldres, int res_off, int dir0, int D,
                          unsigned* mask, long mask_ld, void* stream) {
    XDFM_REQUIRE(xp && x0 && Wf && bias, "cin_level_fwd_ex: null pointer");
    XDFM_REQUIRE(H > 0 && Hp > 0 && m > 0 && N > 0, "cin_level_fwd_ex: bad shape H=%d Hp=%d m=%d N=%ld", H, Hp, m, N);
    XDFM_REQUIRE(act == XDFM_ACT_LINEAR || act == XDFM_ACT_RELU, "cin_level_fwd_ex: unsupported activation %d", act);
    XDFM_REQUIRE(xdfm_cin_level_fwd_ex_supported(H, Hp, m, D), "cin_level_fwd_ex: no f16x3 / bf16 forward kernel with this epilogue "
                 "for H=%d Hp=%d m=%d D=%d (xdfm_cin_level_fwd_ex_supported)", H, Hp, m, D);
    XDFM_REQUIRE(keep_rows >= 0 && keep_rows <= H && (keep_rows == 0 || out), "cin_level_fwd_ex: keep_rows %d of %d", keep_rows, H);
    XDFM_REQUIRE(!res || (dir0 >= 0 && dir0 <= H && ldres >= res_off + (H - dir0) && res_off >= 0 && N % D == 0),
                 "cin_level_fwd_ex: bad direct-sum arguments");
    XDFM_REQUIRE(!mask || (mask_ld >= H && mask_ld % 4 == 0), "cin_level_fwd_ex: mask pitch %ld", mask_ld);
    int logD = 0;
    while ((1 << logD) < D) ++logD;
    const X3FwdEpi epi = {keep_rows, res, ldres, res_off, res ? dir0 : H, logD, mask, mask_ld};
    xdfm_opt_note(OPT_LAST_FWD, xdfm_opt(OPT_CIN_MATH));
    return x3_level_fwd(xp, x0, Wf, bias, H, Hp, m, N, act, out, epi, (hipStream_t)stream);
}

int xdfm_cin_direct_sum(const float* A, int row0, int rows, int B, int D, float* res, long ldres, int off,
                        void* stream) {
    XDFM_REQUIRE(A && res, "cin_direct_sum: null pointer");
    XDFM_REQUIRE(rows >= 0 && B > 0 && D > 0 && row0 >= 0, "cin_direct_sum: bad shape");
    if (rows == 0) return XDFM_OK;
    const long N = (long)B * D;
    if (D >= 4 && D <= 64 && (D & (D - 1)) == 0 && N % 4 == 0 && (((size_t)A) & 15) == 0)
        hipLaunchKernelGGL(cin_direct_sum_vec_kernel, dim3(ceil_div(N, 1024), rows), dim3(256), 0, (hipStream_t)stream, A,
                           row0, N, D, res, ldres, off);
    else if (D <= 64 && (D & (D - 1)) == 0)
        hipLaunchKernelGGL(cin_direct_sum_kernel<true>, dim3(ceil_div((long)B * D, 256), rows), dim3(256), 0,
                           (hipStream_t)stream, A, row0, rows, B, D, res, ldres, off);
    else
        hipLaunchKernelGGL(cin_direct_sum_kernel<false>, dim3(ceil_div(B, 256), rows), dim3(256), 0,
                           (hipStream_t)stream, A, row0, rows, B, D, res, ldres, off);
    return xdfm_check_launch("cin_direct_sum");
}

}  // extern "C"
