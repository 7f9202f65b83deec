/*
 * xdfm.h -- C ABI of libxdfm_hip.so: the MI355X (gfx950) hot path of xDeepFM.
 *
 * The reference (Syclus123/xDeepFM-pytorch) has no FFI: its boundary for this path is a
 * Python class API made of stock ATen calls.  Every entry point below replaces one group of
 * those calls; the citation after "replaces:" is the reference file:line (relative to the
 * reference root) whose arithmetic the kernel reproduces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked [host]; buffers are caller-allocated;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and never
 *     synchronise, allocate or free (safe under hipGraph capture);
 *   - return value: 0 = ok, otherwise an XDFM_ERR_* code; xdfm_last_error() gives the text;
 *   - "FM layout" (feature-map major) of an activation with R rows for a batch of B examples
 *     and embedding width D:  T[r][n], n = b*D + d, row pitch N = B*D floats.  It is the
 *     reference's [B, R, D] tensor with the R axis outermost, so that the 64 lanes of a
 *     wavefront read consecutive n and one example's D values are contiguous.
 *   - operands, accumulators and results are IEEE fp32 ("dtype f32").  The CIN contractions run either on
 *     v_mfma_f32_32x32x2_f32 or, by default, as three v_mfma_f32_32x32x16_f16 per fp32 product on operands
 *     split into fp16 hi + lo halves with fp32 accumulation, or (opt-in) as one v_mfma_f32_32x32x16_bf16 on operands
 *     rounded to bf16 (option "cin_math", see xdfm_set_option);
 *   - nothing in the library issues a memset (hipMemsetAsync nodes inside a captured graph are not ordered
 *     reliably on ROCm 7.2 / gfx950); reductions use per-block partials and fixed-order finish kernels.
 *
 * ABI 2 (this round): + cin_math option and f16x3 pack / workspace layouts, xdfm_cin_pack_all,
 * xdfm_cin_level_bwd_x_ex, xdfm_colsum, xdfm_head_fwd/bwd (K8), xdfm_adam_step (K7), xdfm_graph_node_census.
 * ABI 3: + xdfm_embed_scatter_bwd_marked and xdfm_adam_tensor.grad_marks (gradient buffer kept across steps),
 * xdfm_relu_bwd_colsum.
 * ABI 4: K2 is a sorted, segmented, exact reduce (no atomics; same entry points), xdfm_adam_step_lr (learning rate
 * from a device scalar), read-only options "last_fwd_kernel" / "last_bwx_kernel" / "last_bww_kernel".
 * ABI 5: attention dropout in K5 (p_drop, drop_seed on xdfm_cin_attn_pool_fwd/bwd; xdfm_cin_attn_dropout_mask).
 * ABI 6: deferred (exact) Adam for the tables: xdfm_adam_tensor.last, XDFM_ADAM_DEFERRED, xdfm_adam_clock,
 * xdfm_adam_step_deferred, xdfm_adam_catchup_rows, xdfm_adam_flush.
 * ABI 7: xdfm_cin_bwd_x_is_folded (a pure query instead of the "last_sym" probe on the product path); the deferred update's
 * constant table holds 4 floats per step (xdfm_adam_clock.consts: 4 * cap) and its replayed steps run the short forms of
 * csrc/adam_math.h (same bits, about half the issue slots); xdfm_adam_selftest; xdfm_cin_bwd_prep +
 * xdfm_cin_level_bwd_w_prepared (dOut, its fp16 planes and the dW kernel's scales in one pass); xdfm_set_ticket_board;
 * xdfm_cin_attn_pool_bwd_det (K5's parameter gradients without float atomics); xdfm_cin_level_fwd_ex (direct-connect sums
 * and ReLU sign bits from the forward's epilogue; the direct-connect half of a level is never stored).
 */
#ifndef XDFM_H
#define XDFM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XDFM_ABI_VERSION 8

enum {
    XDFM_OK = 0,
    XDFM_ERR_INVALID = 1,   /* bad shape / null pointer / unsupported option */
    XDFM_ERR_LAUNCH = 2,    /* HIP reported an error at launch */
    XDFM_ERR_NO_DEVICE = 3
};

enum { XDFM_ACT_LINEAR = 0, XDFM_ACT_RELU = 1 };

int xdfm_abi_version(void);
const char* xdfm_last_error(void);          /* [host] thread-local, never NULL */
int xdfm_device_count(void);                /* <0: HIP error code negated */
/* tuning knobs for A/B runs (e.g. "fwd_nf", "bww_nsplit"); unknown key -> XDFM_ERR_INVALID.
 * "cin_math": arithmetic of the CIN contraction (K3 / K4).  Operands, accumulators and results are fp32
 * in both modes; the packed-weight and workspace layouts (and the *_elems sizes) depend on the mode, so
 * set it before the *_pack_elems / *_ws_elems calls of a step and keep it for that step.
 *   1 (default) "f16x3": each fp32 operand is split into two fp16 halves (hi + lo, 22 mantissa bits, exact
 *      power-of-two range fitting) and each product is three v_mfma_f32_32x32x16_f16 accumulated in fp32;
 *      error against an fp64 evaluation <= that of mode 0 (tests/test_gpu_parity.py); shapes without an
 *      f16x3 kernel (odd field counts in the forward, H <= 64 in dW, ...) run mode 0 kernels;
 *   0 "f32mfma": v_mfma_f32_32x32x2_f32 on the fp32 operands;
 *   2 "bf16" (BASELINE config 5's arithmetic): operands rounded to bf16 (RNE), ONE v_mfma_f32_32x32x16_bf16 per
 *      product, fp32 accumulation, fp32 results; no range fitting (bf16 has fp32's exponent range).  Its tolerance is
 *      its own: 2e-2 (outputs) / 4e-2 (gradients) of a tensor's largest magnitude against the fp32 reference,
 *      measured 3e-3 / 8e-3 (tests/test_gpu_parity.py::test_cin_bf16_mfma_path_vs_fp32_oracle).  Kernels exist for
 *      H > 64 (forward, dW), 32 < H <= 256 per call (dX), m in {22, 26}; other shapes run mode 0. */
/* "x3_sym" (default 1): level 0 of a CIN, where x_prev IS x0 (the calls get xp == x0; interaction.py:214-224), is
 * symmetric in (i, j).  With x3_sym != 0 the f16x3 / bf16 kernels of that level contract over the pairs i <= j with the
 * folded weights W(i,j) + W(j,i) -- about half the MFMAs; fields m in {22, 26}.  The packs of every level with Hp == m
 * carry the folded layout behind the plain one (the *_pack_elems sizes include it whatever the option says), and which
 * one a launch reads is decided by xp == x0.  Results: forward and dW as before up to rounding (dW exactly symmetric);
 * dX puts the WHOLE gradient of the level into dx0 and zero-fills dxp when XDFM_BWX_SET_DXP is given (leaves it alone
 * otherwise) -- with xp == x0 only the sum dxp + dx0 was ever meaningful.  Probe "last_sym": bit 0 / 1 / 2 = the last
 * f16x3 / bf16 forward / dX / dW launch ran the folded kernel.
 * "bww_xcd" (default 1): the f16x3 / bf16 dW kernel maps its workgroups so that those of one n-split, which stream the same
 * dOut planes, run on one XCD (one L2) instead of in launch order; same results, 4.7x fewer bytes fetched at config 2. */
/* read-only probes (xdfm_get_option): "last_fwd_kernel", "last_bwx_kernel", "last_bww_kernel" = arithmetic of the kernel
 * the last xdfm_cin_level_fwd / _bwd_x / _bwd_w call launched (0 f32mfma, 1 f16x3, 2 bf16; -1 before the first call):
 * a shape without a kernel in the selected mode runs mode 0, and the tests assert which one ran. */
int xdfm_set_option(const char* key, int value);
int xdfm_get_option(const char* key);
/* Ticket board: `board` = a DEVICE array of `slots` >= 1040 unsigned ints, zeroed once by the caller and kept alive, registered
 * for the CURRENT device (NULL unregisters).  With a board, every kernel of the library that leaves per-block partials lets
 * the block that finishes last add them up (in the same fixed order) instead of a one-block "finish" launch of its own:
 * ten launches fewer per train step, identical results.  The kernels reset the tickets they use; one calling thread and
 * one stream per device at a time (as for the rest of the library). */
int xdfm_set_ticket_board(unsigned* board, int slots);
/* [host] node types of a captured hipGraph_t (the host side replays the train step from a HIP graph):
 * memset nodes are counted separately because they are not ordered reliably against neighbouring kernel
 * nodes on ROCm 7.2 / gfx950 (tools/graph_memset_probe.py) -- a graph with n_memset > 0 must not be replayed;
 * n_unexpected counts nodes that are neither kernel, memcpy, empty nor event nodes. */
int xdfm_graph_node_census(void* graph, int* n_nodes, int* n_memset, int* n_unexpected);

/* ------------------------------------------------------------------ embedding gather (K1)
 * replaces: deepctr/models/basemodel.py:368-370 (26x slice -> .long() -> nn.Embedding),
 *           deepctr/models/basemodel.py:63-92 (Linear: 1-dim tables, sum, dense @ weight),
 *           deepctr/models/xdeepfm.py:86 + deepctr/inputs.py:126-132 (the two concatenations).
 * X        [B][ldx] fp32; ids travel as fp32 exactly as basemodel.py:242 makes them.
 * tables   device array of m table base pointers, table j is [vocab[j]][D];
 * lin_tables device array of m pointers to [vocab[j]][1] tables, or NULL (no linear part);
 * cols     device int[m]: column of X that holds field j's id;   vocab: device int[m];
 * dense_cols device int[nd] (may be NULL when nd == 0); dense_w [nd] (linear_model.weight);
 * emb_fm   out, FM layout [m][B*D]   (the CIN input, reference layout [B,m,D]);
 * dnn_in   out [B][m*D + nd] or NULL (combined_dnn_input);
 * lin_out  out [B] or NULL           (linear_logit);
 * err_flag device int[1] or NULL: bit 0 is OR-ed in when an id is outside [0, vocab) (the id is
 *          then clamped; the reference raises IndexError / device-asserts).
 */
int xdfm_embed_gather_fwd(const float* X, long ldx, int B,
                          const float* const* tables, const float* const* lin_tables,
                          const int* cols, const int* vocab, int m, int D,
                          const int* dense_cols, const float* dense_w, int nd,
                          float* emb_fm, float* dnn_in, float* lin_out,
                          int* err_flag, void* stream);

/* ------------------------------------------------------------------ embedding scatter (K2)
 * replaces: autograd of the above (aten::embedding_dense_backward x52, sparse=False,
 *           deepctr/inputs.py:168) and d(linear_model.weight).
 * d_emb_fm [m][B*D] or NULL, d_dnn_in [B][m*D+nd] or NULL, d_lin [B] or NULL are the incoming
 * gradients.  The dense gradient tables live in ONE caller-initialised buffer d_flat (zeros, or the
 * L2 gradient written by xdfm_l2_reg_bwd): table j starts at d_flat + tab_off[j], its linear table
 * at d_flat + lin_off[j] (device long[m], element offsets; either may be NULL); d_dense_w [nd] is added to as well.
 * No atomics: per field the (id, example) keys of a chunk of <= 4096 examples are sorted in LDS, runs of equal ids
 * are summed EXACTLY (addends rounded once to a fixed-point grid 2^-13 ulp below the run's largest magnitude, added
 * as integers in doubles, rounded once to fp32) and added to d_flat by one lane per row.  The result is a function
 * of the multiset of rows: bit-identical from run to run, under any permutation of the examples of a chunk and on
 * every rank of a row-parallel run (the reference's CPU scatter, deepctr/inputs.py:168, is deterministic too; it adds
 * in example order in fp32, i.e. within a few ulp of this).  Chunks (B > 4096) are launched in ascending order.
 */
int xdfm_embed_scatter_bwd(const float* X, long ldx, int B,
                           const int* cols, const int* vocab, int m, int D,
                           const int* dense_cols, int nd,
                           const float* d_emb_fm, const float* d_dnn_in, const float* d_lin,
                           float* d_flat, const long* tab_off, const long* lin_off, float* d_dense_w,
                           void* stream);
/* Same, and marks[e >> 2] = 1 for every element offset e of d_flat (16-byte aligned) that received a gradient
 * -- one byte per 16-byte chunk; d_dense_w must then point into d_flat too.  A caller that keeps d_flat across
 * steps hands the marks to K7 (xdfm_adam_tensor.grad_marks), which reads and re-zeroes only the marked chunks:
 * the table-sized zero fill of the gradients (optim.zero_grad over dense [V, D] gradients) and
 * the table-sized gradient read of the optimizer step both shrink to the rows the batch touched.
 * marks may be NULL.  ld_dnn / ld_lin: row strides (floats) of d_dnn_in and d_lin, 0 = dense (m*D + nd and 1):
 * the row-parallel exchange hands over ONE gathered buffer whose rows hold [row gradients | X row | d_lin]. */
int xdfm_embed_scatter_bwd_marked(const float* X, long ldx, int B,
                                  const int* cols, const int* vocab, int m, int D,
                                  const int* dense_cols, int nd,
                                  const float* d_emb_fm, const float* d_dnn_in, long ld_dnn,
                                  const float* d_lin, long ld_lin,
                                  float* d_flat, const long* tab_off, const long* lin_off, float* d_dense_w,
                                  unsigned char* marks, void* stream);

/* ------------------------------------------------------------------ CIN level (K3 / K4)
 * One level of deepctr/layers/interaction.py:216-243 (same loop in
 * deepctr/layers/cin_attention.py:257-289, :417-446):
 *     Z[b,(i,j),d] = x_prev[b,i,d] * x0[b,j,d]                 (einsum, :218-222, k = i*m + j)
 *     out[b,h,d]   = act( sum_k W[h,k] * Z[b,k,d] + bias[h] )   (nn.Conv1d k=1 :224, act :226-229)
 * Z is never materialised: it is formed in registers as the B operand of the MFMA contraction.
 * W is conv1ds[i].weight [H][Hp*m] (the trailing 1 of Conv1d dropped), xp is [Hp][N], x0 [m][N],
 * out [H][N], all FM layout, N = B*D.
 */
size_t xdfm_cin_fwd_pack_elems(int H, int Hp, int m);                 /* floats in Wf */
int xdfm_cin_fwd_pack(const float* W, int H, int Hp, int m, float* Wf, void* stream);
int xdfm_cin_level_fwd(const float* xp, const float* x0, const float* Wf, const float* bias,
                       int H, int Hp, int m, long N, int act, float* out, void* stream);

/* The forward of a level with what the CIN does with its output fused into the kernel's epilogue (f16x3 / bf16 arithmetic,
 * D in {4, 8, 16, 32}: xdfm_cin_level_fwd_ex_supported): rows [0, keep_rows) are stored to out [keep_rows][N]; rows >= dir0 are
 * summed over the embedding axis into res (res[b * ldres + res_off + row - dir0], interaction.py:245-246: the
 * direct-connect half of a level in sum pooling is never written out), res == NULL: no sums; mask != NULL: bit n & 31 of
 * mask[(n >> 5) * mask_ld + row] = out[row][n] > 0 for every row (mask_ld >= H, a multiple of 4; mask 16-byte aligned: all
 * the backward needs of a ReLU level's output).
 * At BASELINE config 2 this removes 84 MB of stores, the three xdfm_cin_direct_sum launches and their 84 MB of reads, and
 * (with the mask given to xdfm_cin_bwd_prep) 134 MB of reads in the backward, per step. */
int xdfm_cin_level_fwd_ex_supported(int H, int Hp, int m, int D);
int xdfm_cin_level_fwd_ex(const float* xp, const float* x0, const float* Wf, const float* bias, int H, int Hp, int m, long N,
                          int act, float* out, int keep_rows, float* res, long ldres, int res_off, int dir0, int D,
                          unsigned* mask, long mask_ld, void* stream);

/* sum over the embedding axis of `rows` feature maps (interaction.py:245-246):
 * res[b*ldres + off + r] = sum_d A[(row0 + r)][b*D + d] */
int xdfm_cin_direct_sum(const float* A, int row0, int rows, int B, int D,
                        float* res, long ldres, int off, void* stream);

/* dOut = act'(A) * (dHid + dDirect), dbias[h] += sum_n dOut[h][n]   (autograd of :224-243).
 * A [H][N] is the saved post-activation output of the level.  Rows [hid0, hid0+hid_rows) take
 * dHid [hid_rows][N] (gradient w.r.t. next level's x_prev); rows [dir0, dir0+dir_rows) take the
 * direct-connect gradient: dir_mode 0: dDir is d(result) [B][lddir], value dDir[b*lddir+dir_off+r]
 * broadcast over d (sum pooling); dir_mode 1: dDir is FM layout [.][N], row dir_off + r
 * (attention pooling).  Either part may be absent (rows == 0 / NULL).  dbias must be zeroed by the
 * caller. */
int xdfm_cin_dout(const float* A, int H, int B, int D, int act,
                  const float* dHid, int hid0, int hid_rows,
                  const float* dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
                  float* dOut, float* dbias, void* stream);
/* Same with a workspace of xdfm_cin_dout_ws_elems(H, B, D) floats: the per-block sums of dbias are stored and added up
 * in a fixed order (a second tiny launch) instead of by one float atomic per block -- the same bits on every run. */
size_t xdfm_cin_dout_ws_elems(int H, int B, int D);
int xdfm_cin_dout_det(const float* A, int H, int B, int D, int act,
                  const float* dHid, int hid0, int hid_rows,
                  const float* dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
                  float* dOut, float* dbias, float* ws, void* stream);

/* dx_prev[i][n] += sum_j dZ[(i,j)][n] * x0[j][n];  dx0[j][n] += sum_i dZ[(i,j)][n] * x_prev[i][n]
 * with dZ = W^T dOut recomputed tile by tile.  dxp [Hp][N] and dx0 [m][N] are ACCUMULATED into
 * (caller zero-initialises); dxp and dx0 must NOT alias (for level 0, where x_prev is x0, pass a
 * scratch dxp and add it to dx0 afterwards: how the level's gradient is divided between the two is then up to the
 * kernel -- see option "x3_sym").  H <= 256 rows of the contraction per call. */
size_t xdfm_cin_bwd_pack_elems(int H, int Hp, int m);
int xdfm_cin_bwd_pack(const float* W, int H, int Hp, int m, float* Wz, void* stream);
int xdfm_cin_level_bwd_x(const float* dOut, const float* xp, const float* x0, const float* Wz,
                         int H, int Hp, int m, long N, float* dxp, float* dx0, void* stream);
/* same with flags: XDFM_BWX_SET_DXP / XDFM_BWX_SET_DX0 = store the result instead of adding it to dxp / dx0
 * (the buffer then needs no zero-fill and is not read). */
enum { XDFM_BWX_SET_DXP = 1, XDFM_BWX_SET_DX0 = 2 };
int xdfm_cin_level_bwd_x_ex(const float* dOut, const float* xp, const float* x0, const float* Wz, int H, int Hp,
                            int m, long N, float* dxp, float* dx0, int flags, void* stream);

/* 1 when xdfm_cin_level_bwd_x_ex called with these shapes and xp == x0 (xp_is_x0 != 0), dxp != dx0, under the current
 * "cin_math" / "x3_sym" options runs the folded level-0 kernel, i.e. leaves the level's WHOLE gradient in dx0 (dxp zero);
 * 0 when the caller still has to add dxp to dx0.  A pure function of its arguments and the two options: the host asks
 * it instead of reading the process-global "last_sym" probe, which another thread's launch could have rewritten. */
int xdfm_cin_bwd_x_is_folded(int H, int Hp, int m, int xp_is_x0);

/* dW[h][i*m+j] = sum_n dOut[h][n] * x_prev[i][n] * x0[j][n]   (overwrites dW [H][Hp*m]).
 * ws: workspace of xdfm_cin_bwd_w_ws_elems floats (one partial copy of dW per n-split, summed in a
 * fixed order: bitwise reproducible). */
size_t xdfm_cin_bwd_w_ws_elems(int H, int Hp, int m, long N);
int xdfm_cin_level_bwd_w(const float* dOut, const float* xp, const float* x0,
                         int H, int Hp, int m, long N, float* ws, float* dW, void* stream);

/* The level's backward up to its MFMA kernels in ONE pass over dOut (f16x3 / bf16 arithmetic): xdfm_cin_dout_det that
 * also leaves, in the dW workspace `bww_ws` (xdfm_cin_bwd_w_ws_elems floats), what xdfm_cin_level_bwd_w would otherwise
 * produce with two more passes over dOut -- its fp16 hi / lo planes and the row scales of dOut, x_prev and x0.  The
 * scales are per n-split of the dW kernel (a split's workgroups contract over the split's columns only), each block
 * finds them for its own columns, and the dW kernel removes them from its accumulators before it stores the split's
 * slab.  *prepared [host] = 1 when that happened: the caller then calls xdfm_cin_level_bwd_w_prepared with the same xp,
 * x0, shapes, workspace and options; 0 when the shape has no f16x3 / bf16 dW kernel (H <= 64, D % 4 != 0, unaligned
 * rows, cin_math 0): the call then was xdfm_cin_dout_det and xdfm_cin_level_bwd_w does its own passes.
 * dout_ws: xdfm_cin_bwd_prep_ws_elems floats (dbias partials, added up in a fixed order).  dbias is added to.
 * mask != NULL: the ReLU mask comes from the sign bits xdfm_cin_level_fwd_ex left (bit n & 31 of mask[(n >> 5) * mask_ld + h]
 * = out[h][n] > 0) and A is not read (may be NULL): the level's output then need not be kept at all beyond its hidden
 * rows, which travel as xp. */
size_t xdfm_cin_bwd_prep_ws_elems(int H, int Hp, int m, int B, int D);
/* dOut == NULL in xdfm_cin_bwd_prep (allowed when xdfm_cin_bwd_nodout_supported and a mask is given): the fp32 dOut is not
 * written at all -- the dW kernel takes the planes, and the dX kernel forms its dOut operand itself, from the very
 * sources of this pass (xdfm_cin_level_bwd_x_src: sign bits, dHid, pooled gradient).  67 MB less written and 67 MB less
 * read at level 0 of config 2.  xdfm_cin_level_bwd_x_src: one call covers rows [h0, h0 + H) of the level, H <= 256. */
int xdfm_cin_bwd_nodout_supported(int H, int Hp, int m, int B, int D);
int xdfm_cin_level_bwd_x_src(const unsigned* mask, long mask_ld, const float* dHid, int hid_rows, const float* dDir, int dir_mode,
                             long lddir, int dir_off, int dir0, int dir_rows, int D, int h0, const float* xp, const float* x0,
                             const float* Wz, int H, int Hp, int m, long N, float* dxp, float* dx0, int flags, void* stream);
int xdfm_cin_bwd_prep(const float* A, const unsigned* mask, long mask_ld, int H, int B, int D, int act, const float* dHid,
                      int hid0, int hid_rows, const float* dDir, int dir_mode, long lddir, int dir_off, int dir0, int dir_rows,
                      float* dOut, float* dbias, float* dout_ws, const float* xp, const float* x0, int Hp, int m, float* bww_ws,
                      int* prepared, void* stream);
int xdfm_cin_level_bwd_w_prepared(const float* dOut, const float* xp, const float* x0, int H, int Hp, int m, long N,
                                  float* ws, float* dW, void* stream);

/* ------------------------------------------------------------------ attention pooling (K5)
 * replaces: deepctr/layers/cin_attention.py:63-97 (MultiHeadSelfAttention), :130-144
 *           (AttentionPooling) and the tails of CINAttention.forward :302-313 /
 *           CINAttentionV2.forward :452-464: n_layers x (MHSA -> +residual -> LayerNorm) then
 *           softmax_s(w2 . tanh(W1 x_s + b1)) weighted sum.  The [B,heads,S,S] score tensor of the
 *           reference never exists; one workgroup handles one example with its S tokens on chip.
 * fm     FM layout [S][B*D]: the concatenated direct-connect feature maps (tokens x_s = fm[s][b*D..]).
 * theta  packed parameters: per layer Wq Wk Wv Wo ([D][D], nn.Linear weight layout), then gamma, beta
 *        ([D] each, only if use_ln); then W1 [D][D], b1 [D], w2 [D]  (xdfm_cin_attn_theta_elems floats).
 * nh     number of heads (must divide D; pairs (D, nh) are compiled for D in {4,8,10,16,32}).
 * out    [B][D] pooled vector.  (CINAttention's output_proj D -> featuremap_num is a plain GEMM.)
 * tok_save, o_save [n_layers][B][S][D] (each layer's output tokens / attention output before W_o),
 * ml_save [n_layers][B][S][nh][2] (base-2 log-sum-exp of the scaled scores, 1/sum): written by fwd, read by bwd.
 * bwd: dout [B][D]; dfm [S][B*D] is overwritten; dtheta (same layout as theta) is ACCUMULATED into with
 * fp32 atomics and must be zeroed by the caller.  S <= 1024.
 * p_drop, drop_seed: attention dropout (cin_attention.py:86, nn.Dropout on the softmax output, training mode
 * only).  p_drop = 0 (drop_seed may be NULL) is the reference's default and the evaluation path.  With
 * 0 < p_drop < 1 every attention weight is kept with probability 1-p_drop and scaled by 1/(1-p_drop); the keep
 * bit is a hash of (*drop_seed, example, layer, head, query, key) -- drop_seed is a DEVICE 64-bit scalar so a
 * captured graph draws a new mask per replay -- and nothing is stored: bwd must be given the same p_drop and
 * seed value as the fwd whose buffers it reads.  The stream of bits is not torch's Philox stream; like the
 * reference's CPU and CUDA generators, two back ends agree in distribution, not bit for bit.
 */
size_t xdfm_cin_attn_theta_elems(int D, int n_layers, int use_ln);
int xdfm_cin_attn_pool_fwd(const float* fm, int B, int S, int D, int nh, int n_layers, int use_ln, int use_res,
                           const float* theta, float* out, float* tok_save, float* o_save, float* ml_save,
                           float p_drop, const unsigned long long* drop_seed, void* stream);
int xdfm_cin_attn_pool_bwd(const float* fm, int B, int S, int D, int nh, int n_layers, int use_ln, int use_res,
                           const float* theta, const float* tok_save, const float* o_save, const float* ml_save,
                           const float* dout, float* dfm, float* dtheta, float p_drop,
                           const unsigned long long* drop_seed, void* stream);
/* Same without float atomics: every workgroup stores its share of the parameter gradients in `ws`
 * (xdfm_cin_attn_pool_bwd_ws_elems floats) and a second small launch adds the shares in workgroup order -- dtheta is
 * OVERWRITTEN (no zero-fill) and has the same bits on every run. */
size_t xdfm_cin_attn_pool_bwd_ws_elems(int B, int D, int n_layers, int use_ln);
int xdfm_cin_attn_pool_bwd_det(const float* fm, int B, int S, int D, int nh, int n_layers, int use_ln, int use_res,
                               const float* theta, const float* tok_save, const float* o_save, const float* ml_save,
                               const float* dout, float* dfm, float* dtheta, float* ws, float p_drop,
                               const unsigned long long* drop_seed, void* stream);
/* keep[n_layers][B][nh][S(query)][S(key)] (1 = kept): the mask the two calls above generate for this seed.
 * Test hook: lets a CPU oracle apply the same mask where the reference applies nn.Dropout. */
int xdfm_cin_attn_dropout_mask(int B, int S, int nh, int n_layers, float p_drop, const unsigned long long* drop_seed,
                               unsigned char* keep, void* stream);

/* ------------------------------------------------------------------ L2 regulariser (K6)
 * replaces: deepctr/models/basemodel.py:412-428 (per-tensor square / mul / sum / add loop over
 *           ~58 tensors, every embedding table in full) and its autograd.
 * ptrs: device array of T tensor base pointers, numel: device long[T], coeff: device float[T]
 * (the l2 strength of each tensor).  fwd: out[0] = sum_t coeff[t] * sum(w_t^2), deterministic;
 * partials: scratch of 32*T floats.  bwd: g_t = 2*coeff[t]*gscale[0]*w_t written (accumulate=0) or
 * added (accumulate=1) at gflat + goff[t] (device long[T], element offsets); gscale is a device
 * scalar.
 */
int xdfm_l2_reg_fwd(const float* const* ptrs, const long* numel, const float* coeff, int T,
                    float* partials, float* out, void* stream);
int xdfm_l2_reg_bwd(const float* const* ptrs, const long* numel, const float* coeff, int T,
                    const float* gscale, float* gflat, const long* goff, int accumulate, void* stream);

/* ------------------------------------------------------------------ all weight packs of a step
 * The packs depend on the weights only: with the f16x3 arithmetic one call prepares the forward pack and
 * the dX pack of every level in two launches (max|W| of all levels, then all packs) instead of two per pack.
 * jobs: HOST array; fwd_pack / bwd_pack sized by xdfm_cin_fwd_pack_elems / xdfm_cin_bwd_pack_elems (either may
 * be NULL); every level must satisfy xdfm_cin_pack_all_supported (f16x3 kernels both ways, H <= 256). */
typedef struct {
    const float* W;
    int H, Hp, m;
    float* fwd_pack;
    float* bwd_pack;
} xdfm_cin_pack_job;
int xdfm_cin_pack_all_supported(int H, int Hp, int m);
int xdfm_cin_pack_all(const xdfm_cin_pack_job* jobs, int L, void* stream);

/* ------------------------------------------------------------------ dense-layer bias gradient
 * replaces: autograd of deepctr/layers/core.py:120-134 w.r.t. the bias, grad_bias[c] = sum_r g[r][c].
 * Atomics-free and without any memset (safe inside a captured HIP graph), fixed summation order.
 * g [rows][ld] fp32 (ld >= cols); ws: xdfm_colsum_ws_elems(cols) floats; out [cols]. */
size_t xdfm_colsum_ws_elems(int cols);
int xdfm_colsum(const float* g, long rows, int cols, long ld, float* ws, float* out, void* stream);
/* Backward of y = relu(x W^T + b) up to the GEMMs (deepctr/layers/core.py:120-134 with nn.ReLU, autograd's
 * threshold_backward + the bias gradient): gz [rows][cols] = (y > 0) ? g : 0 and out [cols] = column sums of gz,
 * fixed order; ws as for xdfm_colsum. */
int xdfm_relu_bwd_colsum(const float* g, const float* y, long rows, int cols, long ldg, long ldy, float* ws, float* gz,
                         float* out, void* stream);

/* ------------------------------------------------------------------ output head of the binary task (K8)
 * z_b = lin_b + <u_b, wu> + <v_b, wv> + bias,  pred = sigmoid(z),  loss = sum_b BCE(pred_b, y_b).
 * replaces: cin_linear / dnn_linear (the two [B,K]x[K,1] products of deepctr/models/xdeepfm.py:95-105), the
 * logit sum, PredictionLayer (deepctr/layers/core.py:150-160) and F.binary_cross_entropy(reduction='sum')
 * (basemodel.py:254) with their autograd -- 2 GEMV + ~10 small launches forward, 4 skinny GEMMs + ~6 small
 * launches backward -- by two launches each way; fixed summation order.
 * lin [B] or NULL; u [B][Ku], wu [Ku] (or NULL); v [B][Kv], wv [Kv] (or NULL); bias [1] or NULL; y [B];
 * ws: xdfm_head_ws_elems(Ku, Kv) floats.  Backward: gloss [1]; dlin [B] or NULL; du [B][Ku]; dv [B][Kv];
 * grads [Ku + Kv + 1] = d wu | d wv | d bias. */
size_t xdfm_head_ws_elems(int Ku, int Kv);
int xdfm_head_fwd(const float* lin, const float* u, const float* wu, int Ku, const float* v, const float* wv, int Kv,
                  const float* bias, const float* y, int B, float* pred, float* loss, float* ws, void* stream);
int xdfm_head_bwd(const float* pred, const float* y, const float* gloss, const float* u, const float* wu, int Ku,
                  const float* v, const float* wv, int Kv, int B, float* dlin, float* du, float* dv, float* grads,
                  float* ws, void* stream);

/* ------------------------------------------------------------------ Adam (K7)
 * replaces: torch.optim.Adam.step() (deepctr/models/basemodel.py:452).  The embedding / linear tables carry
 * dense gradients (deepctr/inputs.py:168), so the step streams every parameter: 28 B per parameter, arithmetic
 * of ATen's fused Adam in fp32.  `tensors` is a HOST array of T descriptors (device pointers inside; they travel
 * by value in the kernel arguments, 52 per launch); `step` points to the fp32 step counter torch keeps per
 * parameter, already incremented for this step.
 * l2 > 0 in a descriptor: the kernel uses g + 2*l2*w as the gradient (the term l2 * sum(w^2) of
 * basemodel.py:412-428 with unit upstream gradient); with l2_value != NULL it also returns
 * sum_t l2_t * sum(w_t^2) of the weights BEFORE the update (l2_ws: xdfm_adam_step_ws_elems(T) floats).
 * ABI 3 added grad_marks. */
typedef struct {
    float* param;
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    const float* step;
    long numel;
    float l2;
    /* NULL: grad is read in full and left alone.  Otherwise one byte per 16-byte chunk of grad (which must be
     * 16-byte aligned; see xdfm_embed_scatter_bwd_marked): a chunk whose mark is 0 is taken as zeros without
     * being read; a marked chunk is read, then overwritten with zeros, and its mark cleared -- after the step
     * the gradient buffer is all zeros again, ready for the next scatter.  The numel % 4 tail elements are always
     * read and zeroed. */
    unsigned char* grad_marks;
    /* XDFM_ADAM_LAZY (needs grad_marks): OPT-IN deviation from the reference.  Only marked chunks are updated at all --
     * an untouched row keeps its weight and moments (no moment decay, no L2 pull) until a batch touches it again, as
     * in "lazy" / row-sparse Adam; the step's cost then follows the rows a batch touches (plus one mark byte per 16
     * bytes of table) instead of the vocabulary.  The reference's torch.optim.Adam over dense gradients (deepctr/
     * inputs.py:168 sparse=False, basemodel.py:452) updates every row every step: 0 keeps that arithmetic. */
    int flags;
    /* XDFM_ADAM_DEFERRED (needs grad_marks): one byte per 16-byte chunk of param = the step (counted since the last
     * flush, see xdfm_adam_clock) up to which the chunk's weight and moments have been updated. */
    unsigned char* last;
} xdfm_adam_tensor;
enum { XDFM_ADAM_LAZY = 1, XDFM_ADAM_DEFERRED = 2 };
size_t xdfm_adam_step_ws_elems(int T);
int xdfm_adam_step(const xdfm_adam_tensor* tensors, int T, double lr, double beta1, double beta2, double eps,
                   float* l2_ws, float* l2_value, void* stream);
/* Same with the learning rate read from device memory when lr_dev != NULL (one double; `lr` is then ignored): a
 * captured HIP graph of the step follows a learning-rate schedule (the host rewrites the scalar between replays)
 * instead of being captured again for every value. */
int xdfm_adam_step_lr(const xdfm_adam_tensor* tensors, int T, double lr, const double* lr_dev, double beta1, double beta2,
                      double eps, float* l2_ws, float* l2_value, void* stream);

/* ------------------------------------------------------------------ deferred Adam for the tables (K7d)
 * Same arithmetic, same results, bit for bit, as the dense sweep above -- but the sweep's 24 bytes per table parameter
 * and step are not moved every step.  The reference's torch.optim.Adam over dense table gradients (inputs.py:168,
 * basemodel.py:452) updates every row every step; a row no batch touches sees the gradient 2*l2*w only, so its
 * (w, m, v) after k untouched steps is a function of its state k steps ago and of the steps' bias corrections.  A
 * chunk is therefore updated when it is NEEDED: before a batch gathers it (xdfm_adam_catchup_rows replays its missed
 * steps, in registers), when a gradient arrives for it (XDFM_ADAM_DEFERRED in the step), and every F steps for all
 * chunks (xdfm_adam_flush) -- which bounds every replay to F steps and turns the HBM-bound sweep (2.5 ms per step at
 * 575 M parameters) into an ALU-bound one (0.63 ms per step's worth of updates, tools/ubench/replay.hip) that touches
 * memory once per F steps.  The value of the L2 term of the replayed steps (of the weights before each replayed
 * update) is accumulated into `backlog`: summed over an epoch it equals the dense path's.
 * clock: device int[2] = {steps since the last flush, steps before it}; consts: device float[4*cap] (16-byte aligned), per
 * step since the last flush the step size lr / (1 - beta1^t), c = sqrt(1 - beta2^t), c * 2^32 and RN(1/c) * 2^-32 (0 when the
 * step's constants fall outside the range the replay's short forms are proven for), written by the step itself. */
typedef struct {
    int* clock;
    float* consts;
    int cap;            /* steps the table holds: flush before clock[0] reaches it */
} xdfm_adam_clock;
/* xdfm_adam_step_lr with a clock: advances it, records the step's constants, and treats XDFM_ADAM_DEFERRED tensors by
 * their marks only (replaying, for a marked chunk, whatever steps it still misses, then this one). */
int xdfm_adam_step_deferred(const xdfm_adam_tensor* tensors, int T, const xdfm_adam_clock* clk, double lr,
                            const double* lr_dev, double beta1, double beta2, double eps, float* l2_ws, float* l2_value,
                            void* stream);
/* Device-side tables of one gather's fields: pointers to the rows' parameter / moment / `last` arrays, L2 strengths. */
typedef struct {
    float* const* param;            /* device array [m] */
    float* const* exp_avg;
    float* const* exp_avg_sq;
    unsigned char* const* last;
    const float* l2;                /* device array [m] */
    float* const* grad;             /* xdfm_adam_apply_rows only: the tables' dense gradients and their mark bytes */
    unsigned char* const* marks;
} xdfm_adam_rows;
/* Brings the rows a batch is about to gather (X, cols, vocab as in xdfm_embed_gather_fwd; lin may be NULL) up to the
 * clock.  backlog: one 64-bit device cell (8-byte aligned, zeroed once by the caller); the L2 value of the replayed
 * steps is ADDED to it in 2^-40 fixed point (integer adds: the total does not depend on the order of the threads). */
int xdfm_adam_catchup_rows(const float* X, long ldx, int B, const int* cols, const int* vocab, int m, int D,
                           const xdfm_adam_rows* emb, const xdfm_adam_rows* lin, const xdfm_adam_clock* clk,
                           double beta1, double beta2, double eps, float* backlog, void* stream);
/* The step's update of the deferred tables, keyed by the batch instead of by a scan of the mark bytes: for the rows of X
 * (the batch whose gradients the scatter just wrote: single process), apply step clock[0] -- to be called after
 * xdfm_adam_step_deferred over the OTHER tensors, which advances the clock.  Reads the rows' gradient chunks, zeroes
 * them and their marks, updates the numel % 4 tail elements of every table densely.  l2_cell: 8-byte device scratch
 * (zero before the first call); l2_value[0] += the L2 value of the touched rows (may be NULL). */
int xdfm_adam_apply_rows(const float* X, long ldx, int B, const int* cols, const int* vocab, int m, int D,
                         const xdfm_adam_rows* emb, const xdfm_adam_rows* lin, const xdfm_adam_clock* clk,
                         double beta1, double beta2, double eps, float* l2_cell, float* l2_value, void* stream);
/* Brings every chunk of the XDFM_ADAM_DEFERRED tensors up to the clock, then resets the clock (clock[1] += clock[0],
 * clock[0] = 0, every `last` byte 0). */
int xdfm_adam_flush(const xdfm_adam_tensor* tensors, int T, const xdfm_adam_clock* clk, double beta1, double beta2,
                    double eps, float* backlog, void* stream);

/* Test hook: compares the replay's short forms (csrc/adam_math.h) with the reference spellings ON THE DEVICE.
 * mode 0: square root, n = 2^32 bit patterns; 1: division by a step's constant, n = steps * 2^24 numerators; 2: general
 * division on n random operand pairs inside the guard; 3: whole replayed steps on n random chunks (zeros, denormals and
 * huge values included: the guard's fall-back).  out: device u64[3], zeroed by the caller = {cases compared, mismatches,
 * an encoding of one mismatching case}. */
int xdfm_adam_selftest(int mode, unsigned long long n, unsigned long long seed, double lr, double beta1, double beta2, double eps,
                       unsigned long long* out, void* stream);

/* ------------------------------------------------------------------ SFG heads of xdeepfm_pro: tiled vocabulary CE
 * replaces: the elementwise / reduction chains around the tile GEMMs of nn.Linear(K, V) + F.cross_entropy
 *           (deepctr/xdeepfm_pro/sfg_decoder.py:146-149, :277-283) when the vocabulary is walked in tiles.
 * z [rows][ld]: the logits of one vocabulary tile (T columns).  lse_update: (m[r], s[r]) <- online log-sum-exp of
 * (m[r], s[r]) and the row's T logits (m = -inf, s = 0 before the first tile).  softmax_grad: z <- exp(z - lse[r]) * g[r]
 * in place. */
int xdfm_vocab_lse_update(const float* z, long ld, int rows, int T, float* m, float* s, void* stream);
int xdfm_vocab_softmax_grad(float* z, long ld, int rows, int T, const float* lse, const float* g, void* stream);

/* The same cross-entropy without the logits in HBM (csrc/vocab_ce_x3.hip), for ALL sparse fields' heads over the same
 * hidden rows in one launch per pass: for heads of width K = 32 or 64 a 32 x 32 block of logits lives only in the
 * accumulators of one wave (f16x3 products: hi*hi + hi*lo + lo*hi on fp16 halves of power-of-two scaled operands, fp32
 * accumulation, base-2 exp / log).  Replaces, per sparse field, nn.Linear(K, V) + F.cross_entropy(reduction='none')
 * (deepctr/xdeepfm_pro/sfg_decoder.py:146-149, :277-283) and their autograd backward.  No float atomics: partial
 * results are merged in a fixed order, every row of dW / db is written once.
 *   supported       1 when K is handled and the f16x3 arithmetic is selected (option cin_math == 1), else 0: callers
 *                   then use the tiled path above.
 *   plan            host-side: fills V, vr, item0, blk0, ws_off of fields[0..F) (W, bias, dW, db are the caller's) and
 *                   the work items of the rows-stationary kernels (pass items == NULL to count); returns the number of
 *                   items, *ws_elems = floats of scratch, *n_blk = 128-row weight blocks of all fields.  The caller
 *                   copies both tables to the device.
 *   pack_hidden     hidden [R][K] fp32, contiguous -> `pack` (pack_elems(R, K) floats): MFMA fragments of H and H^T,
 *                   hi / lo halves, one power-of-two scale.  Once per step: every field's head reads the same rows.
 *   fwd             ce[f][r] = logsumexp_v(h_r.W_v + b_v) - (h_r.W_t + b_t), t = targets[f][r] (int64, clamped to
 *                   [0, V_f)); lse2 [F][rows_padded(R)] (zeroed by the caller) receives the base-2 log-sum-exp and
 *                   wmax [F] the bits of max|W_f|: both are inputs of the backward.
 *   pack_g          upstream gradients g [F][R] -> gpack (F * (4 + rows_padded(R)) floats): g scaled to fp16 range.
 *   bwd_h           dh[r][:] = sum_f [ sum_v g[f][r] softmax_f[r][v] W_f[v][:] - g[f][r] W_f[t][:] ]
 *   bwd_w           fields[f].dW [V_f][K], fields[f].db [V_f] = (g (softmax - onehot(target)))^T H, column sums
 *                   (either pointer may be NULL). */
typedef struct { const float* W; const float* bias; float* dW; float* db; long ws_off; int V, vr, item0, blk0; } xdfm_vce_field;
typedef struct { int field, sb0, sb1, range; } xdfm_vce_item;
int xdfm_vocab_ce_x3_supported(int K);
long xdfm_vocab_ce_pack_elems(int R, int K);
long xdfm_vocab_ce_rows_padded(int R);
long xdfm_vocab_ce_plan(int F, const int* V, int R, int K, xdfm_vce_field* fields, xdfm_vce_item* items, long max_items,
                        long* ws_elems, int* n_blk);
int xdfm_vocab_ce_pack_hidden(const float* hidden, long ldh, int R, int K, float* pack, void* stream);
int xdfm_vocab_ce_fwd(const float* pack, const float* hidden, long ldh, int R, int K, const xdfm_vce_field* fields, int F,
                      const xdfm_vce_item* items, long n_items, const long* targets, float* ws, float* ce, float* lse2,
                      unsigned* wmax, void* stream);
int xdfm_vocab_ce_pack_g(const float* g, int F, int R, float* gpack, void* stream);
int xdfm_vocab_ce_bwd_h(const float* pack, int R, int K, const xdfm_vce_field* fields, int F, const xdfm_vce_item* items, long n_items,
                        const long* targets, const float* g, const float* gpack, const float* lse2, unsigned* wmax, float* ws,
                        float* dh, long lddh, void* stream);
int xdfm_vocab_ce_bwd_w(const float* pack, int R, int K, const xdfm_vce_field* fields, int F, int n_blk, const long* targets,
                        const float* gpack, const float* lse2, const unsigned* wmax, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XDFM_H */
