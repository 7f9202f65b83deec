// Shared helpers for the libxdfm_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/xdfm.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define XDFM_WAVE 64

// error plumbing (api.hip)
int xdfm_fail(int code, const char* fmt, ...);
int xdfm_opt(int idx);
void xdfm_opt_note(int idx, int value);    // library-internal: record a probe value
enum {
    OPT_FWD_NF = 0,      // column fragments (32 cols each) per wave in the forward kernel: 1 or 2
    OPT_BWW_NSPLIT,      // 0 = auto; n-range splits of the dW kernel
    OPT_BWW_SLAB,        // 1 (default) = per-split slabs + ordered sum (deterministic); 0 = fp32 atomics
    OPT_BWW_MT,          // 0 = auto; 1/2/4 = row tiles (of 32) per wave in the dW kernel
    OPT_DBG,             // timing experiments only (results become wrong): bit0 = A operand from one cached line
    OPT_CIN_MATH,        // 0 = v_mfma_f32_32x32x2_f32 on fp32 operands; 1 = f16x3 split (cin_x3*.hip) where a kernel exists
    OPT_X3_FWD_MT,       // 0 = auto; 2 / 4 = at most that many row tiles per wave in the f16x3 forward kernel
    OPT_X3_BWX_ROWS,     // 0 = auto (256); rows of the contraction per f16x3 dX launch (host side reads it)
    OPT_X3_WAVES,        // 0 = auto (8 waves per workgroup share a weight ring where the piece count allows); 4 = force 4
    OPT_BWW_PHASE,       // 0 = whole dW call; 1 / 2 / 3 = only the pre-passes / the MFMA kernel / the slab sum (f16x3; timing)
    OPT_ADAM_BX,         // 0 = default (128); blocks per tensor of the Adam kernel
    OPT_LAST_FWD,        // read-only probes: arithmetic of the kernel the last xdfm_cin_level_fwd / _bwd_x / _bwd_w call
    OPT_LAST_BWX,        //   launched: 0 = v_mfma_f32_32x32x2_f32, 1 = f16x3, 2 = bf16 (tests assert which kernel ran)
    OPT_LAST_BWW,
    OPT_LAST_SYM,        // read-only probe: bit 0 / 1 / 2 = the last f16x3 / bf16 forward / dX / dW launch ran the folded level-0 kernel
    OPT_X3_SYM,          // 1 (default) = level 0 (x_prev is x0) contracts over the pairs i <= j with folded weights; 0 = full (i, j) grid
    OPT_BWW_XCD,         // 1 (default) = the f16x3 / bf16 dW kernel keeps the workgroups of an n-split on one XCD; 0 = launch order
    OPT_COUNT
};

// Ticket board (api.hip, xdfm_set_ticket_board): a small zeroed device array the host registers once per device.  A kernel
// that leaves per-block partials takes a ticket when a block is done; the block that draws the last one sums the
// partials in their fixed order and resets the ticket -- the separate one-block "finish" launch (4-6 us of launch floor
// each, ten of them per train step) disappears, the result keeps its bits.  One slot per kernel family: launches of a
// family on one stream are ordered, so they can share it.  Without a board (nullptr) the finish kernels run as before.
// Tickets are drawn with a device-scope atomic on ONE address, and those serialise (~4 ns each): a family whose grids hold
// thousands of blocks gets one ticket per ROW of its grid instead (TK_ROWS slots from TK_ROW0 on: the last block of a row
// finishes that row), the others a single one.
enum { TK_HEAD_FWD = 0, TK_HEAD_BWD, TK_ADAM_L2, TK_SINGLES = 16, TK_ROW0 = 16, TK_ROWS = 1024, TK_COUNT = TK_ROW0 + TK_ROWS };
unsigned* xdfm_ticket(int slot);            // [host] device address of the current device's slot, or nullptr
#ifdef __HIPCC__
// A partial another block of the same launch will read: written through to the level all 8 XCDs see (an agent-scope
// store), so that NO release fence is needed -- on gfx950 an agent-scope release writes back the XCD's whole L2
// (buffer_wbl2), and one of those per block made the fused dOut pass five times slower.
__device__ __forceinline__ void xdfm_publish(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... and read by the last block with ordinary (pipelined) loads after ONE acquire, which xdfm_last_block_done issues for
// the block that returns true (per-load agent-scope atomics cost a round trip each: 128 in a row tripled a 9-us kernel)
__device__ __forceinline__ float xdfm_peer(const float* p) { return *p; }
// true in every thread of the ONE block that finishes last (all threads of every block call it once, at the end, after
// their xdfm_publish stores).  Order: the published stores are acknowledged (vmcnt 0) before the block draws its ticket.
__device__ __forceinline__ bool xdfm_last_block_done(unsigned* __restrict__ ticket, unsigned blocks) {
    __shared__ unsigned xdfm_tk_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xdfm_tk_last = t == blocks - 1 ? 1u : 0u;
        if (t == blocks - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the family's next launch
    }
    __syncthreads();
    const bool last = xdfm_tk_last != 0u;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the other blocks' published partials, past this XCD's cache
    return last;
}
#endif

#define XDFM_REQUIRE(cond, ...) \
    do { if (!(cond)) return xdfm_fail(XDFM_ERR_INVALID, __VA_ARGS__); } while (0)

static inline int xdfm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xdfm_fail(XDFM_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return XDFM_OK;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

// ---- geometry shared by pack and compute kernels --------------------------------------------
// Forward: the wave's output tile is (32*MT rows) x (32*NF cols); MT is a function of H only so
// that xdfm_cin_fwd_pack and xdfm_cin_level_fwd agree on the packed layout.
static inline int fwd_mt(int H) {
    int t = ceil_div(H, 32);
    if (t >= 8) return 8;
    if (t > 2) return 4;
    return t;            // 1 or 2
}
#define FWD_PD 3          // A-fragment prefetch distance in k-steps (ring of 4)
#define BWX_PD 3          // A-group prefetch distance in the dX kernel (ring of 4 float4)
// zero groups appended to the packed W^T stream: the ring of a wave that runs 2 chains looks ahead into
// the same chain of the next chain pair, i.e. up to 2*HS4 + PD groups past the last real one
#define BWX_TAIL(hs4) (2 * (hs4) + BWX_PD)
static inline int fwd_mp(int m) { return (m + 1) / 2; }                       // j-pairs per i
static inline long fwd_tpad(int Hp, int m) { return round_up((long)Hp * fwd_mp(m), 4); }
static inline int bwx_hs4(int H) {      // float4 groups along the contraction (h) axis
    int g = ceil_div(H, 8);
    int p = 1;
    while (p < g) p <<= 1;
    return p;                            // 1,2,4,...,32
}

// C/D fragment of v_mfma_f32_32x32x2_f32: register r of lane (c = lane&31, s = lane>>5)
// holds element [row = (r&3) + 8*(r>>2) + 4*s][col = c].
__device__ __forceinline__ int frag_row(int r, int s) { return (r & 3) + 8 * (r >> 2) + 4 * s; }

// ---- f16x3 / bf16 path (cin_x3*.hip) -------------------------------------------------------------
// MFMA terms per fp32 product of the selected arithmetic: 3 = f16x3 (hi*hi + hi*lo + lo*hi on fp16 halves, cin_math 1),
// 1 = bf16 operands (cin_math 2: one v_mfma_f32_32x32x16_bf16, no range fitting: bf16 has fp32's exponent), 0 = neither
static inline int x3_terms() { const int m = xdfm_opt(OPT_CIN_MATH); return m == 1 ? 3 : (m == 2 ? 1 : 0); }
#define X3_HDR 128        // floats in front of a packed weight stream: [0] scale, [1] 1/scale, [64..127] partial maxima of |W|
// forward ring: R stage slots in LDS; a stage = SPS consecutive MFMA steps (one barrier per stage, not per step)
#define X3_RING 4
#define X3_STAGE_KB 16
#define X3_BWX_RING 4          // dX kernel: ring slots of its weight stream
// steps per stage for MT row tiles per wave, nt MFMA terms (3 = f16x3: 2 KB per tile and step, 1 = bf16: 1 KB) and m
// fields: X3_STAGE_KB worth of steps, but few enough that the next block's x_prev rows (issued at the first stage
// boundary of a block of m/2 steps, published R-2 boundaries later) are there before the block's last step
constexpr __host__ __device__ inline int x3_fwd_sps(int MT, int nt, int m) {
    const int stepkb = (nt == 3 ? 2 : 1) * MT;
    int sps = X3_STAGE_KB / stepkb;
    if (sps < 1) sps = 1;
    const int lim = (m / 2 - 1) / (X3_RING - 1);
    while (sps > 1 && sps > lim) sps >>= 1;
    return sps;
}
// Level 0 of the CIN multiplies x0 with itself: Z[(i, j)] == Z[(j, i)], so the contraction runs over the pairs
// i <= j with the folded weights W[h][(i, j)] + W[h][(j, i)] (W[h][(i, i)] on the diagonal).  The forward kernel's two
// lane halves hold the same column, the second one with its x0 registers in reverse field order: slot q of the pair
// list {(i, j): i <= j, i + j <= m - 1} (rows i = 0 .. m/2 - 1 of lengths m - 2i) then means (i, j) to lane half 0 and
// (m-1-i, m-1-j) to lane half 1 -- together every pair; those with i + j == m - 1 twice, so half 1's weight is 0 there.
constexpr __host__ __device__ inline int x3_sym_pairs(int m) { return (m / 2) * (m / 2 + 1); }     // slots per lane half (even)
constexpr __host__ __device__ inline int x3_sym_steps(int m) { return (x3_sym_pairs(m) + 7) / 8; }
constexpr __host__ __device__ inline int x3_sym_i(int m, int q) {       // m / 2 for the padding slots behind the list
    int i = 0;
    while (i < m / 2 && q >= m - 2 * i) { q -= m - 2 * i; ++i; }
    return i;
}
constexpr __host__ __device__ inline int x3_sym_j(int m, int q) {
    int i = 0;
    while (i < m / 2 && q >= m - 2 * i) { q -= m - 2 * i; ++i; }
    return i + q;
}
constexpr __host__ __device__ inline int x3_fwd_sps_sym(int MT, int nt) {
    const int sps = X3_STAGE_KB / ((nt == 3 ? 2 : 1) * MT);
    return sps < 1 ? 1 : sps;
}
static inline bool x3_sym_m(int m) { return m == 22 || m == 26; }      // field counts with a folded-level-0 kernel instance
struct X3Geom {
    int MT, MB;           // row tiles (of 32) per wave, row groups (blockIdx.y)
    int MP;               // MFMA steps per full block of 8 x_prev rows (= m / 2)
    int FB, RH, TS, NS;   // full blocks, rows per lane half / steps of the ragged last block, total steps
    int SPS, NSA;         // steps per ring stage; steps stored per row group (whole stages + one spare, zero past NS)
};
// What the f16x3 / bf16 forward kernel does with a level's output besides (or instead of) storing it (xdfm_cin_level_fwd_ex):
//   keep_rows  rows [0, keep_rows) are stored to `out` (pitch N); rows beyond exist only in the accumulators.  The CIN
//              stores its hidden half (the next level's x_prev) and, in sum pooling, nothing of its direct-connect half:
//              those rows are consumed right here by
//   res        res[b * ldres + res_off + row - dir0] = sum_d out[row][b * D + d] for rows >= dir0 (interaction.py:245-246),
//              D = 1 << logD in {4, 8, 16, 32}: a D-lane segment sum in the accumulator layout (lane = column), and
//   mask       bit (n & 31) of mask[(n >> 5) * mask_ld + row] = out[row][n] > 0 (mask_ld >= H, a multiple of 4): all the
//              backward needs of a ReLU level's saved output once x_prev is saved apart (1 bit instead of 4 bytes per
//              element, written and read).
struct X3FwdEpi {
    int keep_rows;
    float* res; long ldres; int res_off, dir0, logD;
    unsigned* mask; long mask_ld;
};
static inline X3FwdEpi x3_fwd_epi_plain(int H) { return X3FwdEpi{H, nullptr, 0, 0, 0, 0, nullptr, 0}; }
X3Geom x3_fwd_geom(int H, int Hp, int m);
X3Geom x3_fwd_geom_sym(int H, int m);          // level 0 with folded weights (x3_sym_*): one block of x3_sym_steps(m) steps
bool x3_fwd_usable(int H, int Hp, int m);
size_t x3_fwd_pack_elems(int H, int Hp, int m);
int x3_fwd_pack(const float* W, int H, int Hp, int m, float* pack, hipStream_t st);
int x3_level_fwd(const float* xp, const float* x0, const float* pack, const float* bias, int H, int Hp, int m, long N,
                 int act, float* out, const X3FwdEpi& epi, hipStream_t st);
// level 0 with folded weights (cin_x3_fwd_sym.hip; m with x3_sym_m): `pack` = the folded pack, g = x3_fwd_geom_sym
int x3_level_fwd_sym(const float* x0, const float* pack, const float* bias, int H, int m, long N, const X3Geom& g, int nt,
                     int act, float* out, const X3FwdEpi& epi, hipStream_t st);
// instances for the other even field counts (cin_x3_fwd_ma.hip: m < 22, cin_x3_fwd_mb.hip: 22 < m <= 40)
int x3_level_fwd_ma(const float* xp, const float* x0, const float* pack, const float* bias, int H, int Hp, int m, long N,
                    const X3Geom& g, int nt, int act, float* out, const X3FwdEpi& epi, hipStream_t st);
int x3_level_fwd_mb(const float* xp, const float* x0, const float* pack, const float* bias, int H, int Hp, int m, long N,
                    const X3Geom& g, int nt, int act, float* out, const X3FwdEpi& epi, hipStream_t st);
struct X3BwxGeom {
    int HBT, HBS, IB;     // h-blocks (of 16) in total / per ring stage, i-blocks (of 32)
    long NT;              // tiles = IB * m
};
// Where the dX kernels take dOut from.  dOut != nullptr: the materialised [H][N] tensor.  Otherwise they FORM it in their
// prologue, exactly as cin_dout does (cin_bwd.hip) -- dOut[h][n] = relu'(out[h][n]) * (dHid[h][n] + direct-connect gradient) --
// from the gradient of the next level's x_prev, the pooled gradient and the sign bits the forward left (X3FwdEpi.mask,
// transposed layout): the fp32 dOut (67 MB at level 0 of config 2) is then neither written by xdfm_cin_bwd_prep nor read
// here.  Rows are those of the level: this launch covers rows [h0, h0 + H).
struct X3DoutSrc {
    const float* dOut;
    const unsigned* mask; long mask_ld;                  // nullptr: linear activation
    const float* dHid; int hid_rows;                     // rows [0, hid_rows) of the level
    const float* dDir; int dir_mode; long lddir; int dir_off, dir0, dir_rows, logD;
    int h0;
};
static inline X3DoutSrc x3_dout_plain(const float* dOut) { return X3DoutSrc{dOut, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, 0, 0, 0, 0}; }
X3BwxGeom x3_bwx_geom(int H, int Hp, int m);
bool x3_bwx_usable(int H, int Hp, int m);
size_t x3_bwx_pack_elems(int H, int Hp, int m);
int x3_bwx_pack(const float* W, int H, int Hp, int m, float* pack, hipStream_t st);
int x3_level_bwd_x(const X3DoutSrc& S, const float* xp, const float* x0, const float* pack, int H, int Hp, int m, long N,
                   float* dxp, float* dx0, int flags, hipStream_t st);

// level 0 with folded weights (cin_x3_bwx_sym.hip): the whole gradient goes to dx0, dxp is zero-filled if it is to be set
int x3_level_bwd_x_sym(const X3DoutSrc& S, const float* x0, const float* pack, int H, int m, long N, int HBT, int nt,
                       float* dxp, float* dx0, int flags, hipStream_t st);

// ---- dW geometry (cin_bwd.hip, cin_x3_bww.hip) ---------------------------------------------------
#define BWW_NC 32         // columns per staged chunk
static inline int bww_mt(int H) {
    const int o = xdfm_opt(OPT_BWW_MT);
    if (o == 1 || o == 2 || o == 4) return o;
    return H > 64 ? 4 : (H > 32 ? 2 : 1);
}
#define BWW_JT 2

struct BwwGeom {
    int MT, IB, JP, TPH, HG, Hpad, IPAD, gx, nsplit;
    long n_per_split, slab;        // slab = elements of one dWt copy
};
static inline BwwGeom bww_geometry(int H, int Hp, int m, long N) {
    BwwGeom g;
    g.MT = bww_mt(H);
    g.IB = ceil_div(Hp, 32);
    g.JP = ceil_div(m, BWW_JT);
    g.TPH = (int)round_up((long)g.JP * g.IB, 4);
    g.HG = ceil_div(H, 32 * g.MT);
    g.Hpad = g.HG * 32 * g.MT;
    g.IPAD = g.IB * 32;
    g.gx = g.HG * g.TPH / 4;
    int nsplit = xdfm_opt(OPT_BWW_NSPLIT);
    const int max_split = ceil_div(N, BWW_NC);
    // default: one resident round -- 2 workgroups per CU (LDS / VGPR limit) x 256 CUs.  More splits
    // only add reduction traffic and a ragged last round (1027 workgroups on 512 slots ran 3 rounds).
    if (nsplit <= 0) nsplit = 512 / g.gx > 0 ? 512 / g.gx : 1;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit > 65535) nsplit = 65535;
    g.n_per_split = round_up(ceil_div(N, nsplit), BWW_NC);
    g.nsplit = ceil_div(N, g.n_per_split);
    g.slab = (long)m * g.Hpad * g.IPAD;
    return g;
}


bool x3_bww_usable(const float* dOut, const float* xp, const float* x0, int H, long N);
size_t x3_bww_ws_elems(int H, int Hp, int m, long N);
int x3_level_bwd_w(const float* dOut, const float* xp, const float* x0, int H, int Hp, int m, long N, float* ws,
                   float* dW, bool prepared, hipStream_t st);
// cin_dout that also leaves the dW kernel's operands (planes, per-split scales) in the dW workspace (cin_x3_bww.hip)
int x3_bwd_prep_blocks(bool xp_is_x0, int H, int Hp, int m, long N);      // dbias partials per row
int x3_bwd_prep(const float* A, const unsigned* mask, long mask_ld, int H, long N, int D, int act, const float* dHid, int hid0, int hid_rows, const float* dDir,
                int dir_mode, long lddir, int dir_off, int dir0, int dir_rows, float* dOut, float* slots, float* dbias,
                unsigned* ticket, const float* xp, const float* x0, int Hp, int m, float* ws, hipStream_t st);

bool x3_pack_all_usable(int H, int Hp, int m);
int x3_pack_all(const xdfm_cin_pack_job* jobs, int L, hipStream_t st);

// LDS-DMA (global_load_lds) issued as inline assembly.  Through the builtin, hipcc books the instruction on vmcnt AND on
// lgkmcnt ("flat access that may touch LDS") and, while one is pending -- always, with a ring kept several stages ahead --
// turns every wait for an LDS read into s_waitcnt lgkmcnt(0): look-ahead ds_reads are then waited for the moment they
// are issued.  Hidden from the compiler, the DMA is ordered by the kernels' own counted vmcnt waits and barriers (it
// always was), and LDS reads get counted lgkmcnt waits.  gsrc: this lane's 16 (4) bytes; lds_wave_base: wave-uniform
// LDS byte address (x3_lds_addr), the hardware adds lane * 16 (4).
#ifdef __HIPCC__
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ unsigned x3_lds_addr(const void* p) {       // byte address inside the workgroup's LDS
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void x3_lds_dma16(const void* gsrc, unsigned lds_wave_base) {
    const unsigned lo = __builtin_amdgcn_readfirstlane(lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lo) : "memory", "m0");
}
__device__ __forceinline__ void x3_lds_dma4(const void* gsrc, unsigned lds_wave_base) {
    const unsigned lo = __builtin_amdgcn_readfirstlane(lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lo) : "memory", "m0");
}
#pragma clang diagnostic pop
#endif
