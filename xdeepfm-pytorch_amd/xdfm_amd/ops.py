"""torch.autograd wrappers around the C ABI of libxdfm_hip.so.

PyTorch is plumbing here (device memory, the current HIP stream, the autograd tape); every
arithmetic step of the embedding gather and of the CIN stack runs in the hand-written kernels.
All ops require float32 CUDA (ROCm) tensors and raise otherwise -- there is no CPU path.
"""
import ctypes
import os

import numpy as np
from typing import List, Optional, Sequence

import torch

from . import _lib

ACT_CODES = {"linear": 0, "relu": 1}

# bench.py sets this to a list to collect (name, algorithmic work, start event, end event) for each
# heavy launch; the events are recorded on torch's current stream, the one the kernels run on.
PROFILE = None


def _run(name, work, fn):
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    PROFILE.append((name, work, e0, e1))
    return rc


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _need_cuda(t: torch.Tensor, name: str):
    if t.is_cuda:
        _lib.ticket_board(t.device)
    if not t.is_cuda:
        raise RuntimeError(
            "xdfm: %s is on %s -- the xDeepFM hot path only runs on an MI355X through libxdfm_hip.so; "
            "there is no CPU fallback (the CPU restatement lives in oracle/ and is test-only)." % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("xdfm: %s must be float32, got %s" % (name, t.dtype))


def activation_code(name) -> int:
    key = name.lower() if isinstance(name, str) else name
    if key not in ACT_CODES:
        raise NotImplementedError("xdfm CIN kernels implement activation 'relu' and 'linear', got %r" % (name,))
    return ACT_CODES[key]


# --------------------------------------------------------------------------------------------- #
# embedding gather                                                                               #
# --------------------------------------------------------------------------------------------- #
class EmbedPlan:
    """Device-side metadata of one gather: which column of X feeds which table."""

    def __init__(self, sparse_cols: Sequence[int], vocab: Sequence[int], dense_cols: Sequence[int], emb_dim: int):
        self.m = len(sparse_cols)
        self.nd = len(dense_cols)
        self.D = int(emb_dim)
        self.sparse_cols = [int(c) for c in sparse_cols]
        self.vocab = [int(v) for v in vocab]
        self.dense_cols = [int(c) for c in dense_cols]
        self._dev = {}
        self._ptr_cache = {}
        self._off_cache = {}
        self.dp = None            # set by xdfm_amd.dist to exchange row gradients across ranks
        self.reg_defer = None     # (gscale, L2Plan) left by L2Reg.backward: table L2 gradient still owed
        self.stash = None         # list: gather backwards park their inputs here instead of scattering (split step)
        self.arena_on = False     # the model's train step sets this when its optimizer consumes marked gradients
        self._arenas = {}
        self.catchup = None       # set by optim.TableAdam (deferred update): brings the rows of X up to date before they are read
        self.last_gather = None   # (X, embedding tables, linear tables) of the latest gather: the rows a deferred update is keyed by

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            i32 = dict(dtype=torch.int32, device=device)
            self._dev[key] = (torch.tensor(self.sparse_cols, **i32), torch.tensor(self.vocab, **i32),
                              torch.tensor(self.dense_cols if self.nd else [0], **i32),
                              torch.zeros(1, **i32))
        return self._dev[key]

    def pointer_table(self, tensors: Sequence[torch.Tensor], tag: str):
        """int64 device array of base pointers (re-uploaded only when a pointer changed)."""
        ptrs = tuple(t.data_ptr() for t in tensors)
        hit = self._ptr_cache.get(tag)
        if hit is None or hit[0] != ptrs:
            hit = (ptrs, torch.tensor(ptrs, dtype=torch.int64, device=tensors[0].device))
            self._ptr_cache[tag] = hit
        return hit[1]

    def grad_layout(self, shapes, device):
        """Element offsets of every table gradient inside the flat gradient buffer (device int64)."""
        key = (tuple(shapes), str(device))
        hit = self._off_cache.get(key)
        if hit is None:
            sizes = [sh[0] * sh[1] for sh in shapes]
            offs, off = [], 0
            for n in sizes:                  # every gradient starts on a 16-byte boundary (K7's float4 path; marks)
                offs.append(off)
                off += (n + 3) // 4 * 4
            hit = (sizes, offs, off, torch.tensor(offs, dtype=torch.int64, device=device))
            self._off_cache[key] = hit
        return hit

    def arena(self, shapes, device):
        """The flat gradient buffer kept ACROSS steps (GradArena), or None when that mode is off."""
        if not self.arena_on or os.environ.get("XDFM_GRAD_ARENA", "1") == "0":
            return None
        key = (tuple(shapes), str(device))
        hit = self._arenas.get(key)
        if hit is None:
            _, _, total, _ = self.grad_layout(shapes, device)
            hit = self._arenas[key] = GradArena(total + max(self.nd, 1), device)
        return hit

    def arenas(self):
        return list(self._arenas.values())

    def check_ids(self, device) -> bool:
        """True when a gather since the last call saw an id outside [0, vocab) (syncs the stream)."""
        flag = self.on(device)[3]
        bad = bool(flag.item())
        flag.zero_()
        return bad


class GradArena:
    """Dense table gradients without the table-sized passes (`optim.zero_grad` of dense [V, D] grads).

    The reference's tables carry dense gradients (deepctr/inputs.py:168), all zeros except the <= B rows per table a
    batch touches; a step pays a table-sized zero fill and a table-sized gradient read in the optimizer for them.
    Here the flat gradient buffer lives across steps: the scatter marks the 16-byte chunks it adds to
    (`xdfm_embed_scatter_bwd_marked`), K7 reads only marked chunks and writes zeros back (`xdfm_adam_tensor.grad_marks`),
    so the buffer is clean again when the step ends.  `pending` = byte offsets of the views handed to autograd that
    no optimizer step has consumed yet; a scatter that finds leftovers (a step without K7, a gradient that autograd
    copied instead of adopting) clears everything the slow way first.  `EmbedPlan.arena_on` is raised by the model's
    own train step only (models.py::_own_step_scope): K7 never sees what other code adds to a gradient outside the
    marked chunks, so user-driven autograd loops get ordinary dense gradients.  Consequence worth knowing: after the
    model's own step the tables' `.grad` read zeros (they are views of this buffer)."""

    def __init__(self, numel, device):
        self.flat = torch.zeros(numel, dtype=torch.float32, device=device)
        self.marks = torch.zeros(numel // 4 + 2, dtype=torch.uint8, device=device)
        self.base, self.nbytes = self.flat.data_ptr(), numel * 4
        self.pending = set()
        self.full_clears = 0

    def begin(self):
        if self.pending:
            self.flat.zero_()
            self.marks.zero_()
            self.pending.clear()
            self.full_clears += 1

    def hand_out(self, views):
        for v in views:
            if v is not None:
                self.pending.add(v.data_ptr() - self.base)

    def marks_ptr(self, grad_ptr):
        """Address of the mark bytes of a gradient that is one of this arena's views (else None)."""
        off = grad_ptr - self.base
        if 0 <= off < self.nbytes and off % 16 == 0 and off in self.pending:
            return self.marks.data_ptr() + off // 16
        return None

    def consumed(self, grad_ptr):
        self.pending.discard(grad_ptr - self.base)


class EmbedGather(torch.autograd.Function):
    """X [B, cols] -> (emb_fm [m, B*D], dnn_in [B, m*D+nd], lin [B, 1]).

    replaces deepctr/models/basemodel.py:354-380 + :63-92 + deepctr/inputs.py:126-132."""

    @staticmethod
    def forward(ctx, X, dense_w, plan: EmbedPlan, has_lin: bool, *tables):
        _need_cuda(X, "X")
        lib = _lib.load()
        m, D, nd = plan.m, plan.D, plan.nd
        emb_tables = tables[:m]
        lin_tables = tables[m:2 * m] if has_lin else ()
        for t in tables:
            _need_cuda(t, "embedding table")
            if not t.is_contiguous():
                raise ValueError("xdfm: embedding tables must be contiguous")
        X = X.contiguous()
        B = X.shape[0]
        cols, vocab, dcols, flag = plan.on(X.device)
        emb_fm = torch.empty((m, B * D), dtype=torch.float32, device=X.device)
        dnn_in = torch.empty((B, m * D + nd), dtype=torch.float32, device=X.device)
        lin = torch.empty((B, 1), dtype=torch.float32, device=X.device)
        plan.last_gather = (X, tuple(emb_tables), tuple(lin_tables))
        if plan.catchup is not None:
            plan.catchup(plan, X, emb_tables, lin_tables)
        tp = plan.pointer_table(emb_tables, "emb")
        lp = plan.pointer_table(lin_tables, "lin") if has_lin else None
        dw = dense_w.contiguous() if (dense_w is not None and nd > 0) else None
        # algorithmic bytes: X row + table rows (+ linear rows) + both outputs + logit
        nbytes = B * (4 * (m + nd) + m * (4 * D + (4 if has_lin else 0)) + 4 * m * D + 4 * (m * D + nd) + 4)
        _lib.check(_run("embed_gather_fwd[bytes]", nbytes, lambda: lib.xdfm_embed_gather_fwd(
            _ptr(X), X.stride(0), B, _ptr(tp), _ptr(lp), _ptr(cols), _ptr(vocab), m, D,
            _ptr(dcols) if nd else None, _ptr(dw), nd, _ptr(emb_fm), _ptr(dnn_in), _ptr(lin), _ptr(flag),
            _stream())), "embed_gather_fwd")
        plan.reg_defer = None      # a deferral is only valid between this forward and its backward
        ctx.plan, ctx.has_lin = plan, has_lin
        ctx.shapes = [tuple(t.shape) for t in tables]
        ctx.save_for_backward(X, *tables)
        return emb_fm, dnn_in, lin

    @staticmethod
    def backward(ctx, d_emb, d_dnn, d_lin):
        X = ctx.saved_tensors[0]
        tables = ctx.saved_tensors[1:]
        plan = ctx.plan
        if plan.stash is not None:
            # split step (row-parallel run replayed from a HIP graph): the scatter needs the other ranks' rows, so
            # it runs after the exchange, outside the captured part -- see apply_stashed_scatter
            plan.stash.append((plan, X, d_emb, d_dnn, d_lin, ctx.has_lin, ctx.shapes, tables, ctx.needs_input_grad[1]))
            return (None, None, None, None) + (None,) * len(tables)
        need_w = ctx.needs_input_grad[1]
        grads, d_w = EmbedGather.scatter(plan, X, d_emb, d_dnn, d_lin, ctx.has_lin, ctx.shapes, tables, need_w)
        return (None, d_w if (need_w and plan.nd) else None, None, None) + tuple(grads)

    @staticmethod
    def scatter(plan, X, d_emb, d_dnn, d_lin, has_lin, shapes, tables, need_w=True):
        """Dense table gradients (views of one flat buffer) and the dense-weight gradient from the row gradients."""
        lib = _lib.load()
        m, D, nd = plan.m, plan.D, plan.nd
        dev = X.device
        sizes, offs, total, off_dev = plan.grad_layout(shapes, dev)
        # ONE buffer holds every dense table gradient (+ the dense-weight gradient at its end).  It
        # starts as zeros, or -- when L2Reg.backward deferred the tables' L2 term to us -- as
        # 2*l2*gscale*w, which saves a memset, a table-sized temporary and one add per table.
        defer, plan.reg_defer = plan.reg_defer, None
        arena = plan.arena(shapes, dev) if plan.arena_on and defer is None and all(t.grad is None for t in tables) else None
        marks = None
        if arena is not None:
            arena.begin()
            flat, marks = arena.flat, arena.marks
        elif defer is not None:
            gscale, l2plan = defer
            flat = torch.empty(total + max(nd, 1), dtype=torch.float32, device=dev)
            flat[total:].zero_()
            ptrs, numel, coeff = l2plan.tables(tables)
            _lib.check(lib.xdfm_l2_reg_bwd(_ptr(ptrs), _ptr(numel), _ptr(coeff), len(tables), _ptr(gscale),
                                           _ptr(flat), _ptr(off_dev), 0, _stream()), "l2_reg_bwd (deferred)")
        else:
            flat = torch.zeros(total + max(nd, 1), dtype=torch.float32, device=dev)
        grads = [flat[o:o + n].view(sh) for o, n, sh in zip(offs, sizes, shapes)]
        d_w = flat[total:total + nd].view(nd, 1) if nd else None
        if arena is not None:
            arena.hand_out(grads + ([d_w] if need_w else []))
        cols, vocab, dcols, _ = plan.on(dev)
        tab_off = off_dev[:m]
        lin_off = off_dev[m:2 * m] if has_lin else None
        pieces = [(X, d_emb, d_dnn, d_lin)]
        if plan.dp is not None:
            pieces = plan.dp.exchange_rows(X, d_emb, d_dnn, d_lin)
        for (Xr, de, dd, dl) in pieces:
            B = Xr.shape[0]
            # the exchange hands over strided views of one gathered buffer: rows may be wider than their payload
            de = de.contiguous() if de is not None else None
            if dd is not None and dd.stride(-1) != 1:
                dd = dd.contiguous()
            dl = dl.reshape(B) if dl is not None else None
            ld_dnn = dd.stride(0) if dd is not None else 0
            ld_lin = dl.stride(0) if dl is not None and B > 1 else (1 if dl is not None else 0)
            nbytes = B * (4 * (m + nd) + 2 * 4 * m * D + 4 + 4 * m * (D + 1))
            _lib.check(_run("embed_scatter_bwd[bytes]", nbytes, lambda: lib.xdfm_embed_scatter_bwd_marked(
                _ptr(Xr), Xr.stride(0), B, _ptr(cols), _ptr(vocab), m, D, _ptr(dcols) if nd else None, nd,
                _ptr(de), _ptr(dd), ld_dnn, _ptr(dl), ld_lin, _ptr(flat), _ptr(tab_off), _ptr(lin_off), _ptr(d_w),
                _ptr(marks), _stream())), "embed_scatter_bwd")
        return grads, d_w


def apply_stashed_scatter(stash, dense_w, params=None):
    """Second half of a split step: exchange + scatter for every stashed gather backward; the gradients are
    assigned to `.grad` of the tables (`params`: the model's own parameter objects, in the gather's order) and of
    `dense_w`, the linear part's dense weight, directly."""
    with torch.no_grad():
        for (plan, X, d_emb, d_dnn, d_lin, has_lin, shapes, tables, need_w) in stash:
            grads, d_w = EmbedGather.scatter(plan, X, d_emb, d_dnn, d_lin, has_lin, shapes, tables,
                                             bool(need_w and dense_w is not None))
            owners = params if params is not None and len(params) == len(tables) else tables
            for t, g in zip(owners, grads):
                t.grad = g
            if need_w and d_w is not None and dense_w is not None:
                dense_w.grad = d_w if plan.arena_on else d_w.clone()


# --------------------------------------------------------------------------------------------- #
# CIN stack                                                                                      #
# --------------------------------------------------------------------------------------------- #
def cin_geometry(m: int, layer_size: Sequence[int], split_half: bool):
    """Per level: (H, Hp, hid_rows, dir0, dir_rows, dir_off).  Mirrors field_nums of
    deepctr/layers/interaction.py:183-201 and the split of :231-240."""
    levels, Hp, off = [], m, 0
    L = len(layer_size)
    for i, H in enumerate(layer_size):
        if split_half:
            if i != L - 1:
                hid, dir0, drows = H // 2, H // 2, H - H // 2
            else:
                hid, dir0, drows = 0, 0, H
        else:
            hid, dir0, drows = H, 0, H
        levels.append((H, Hp, hid, dir0, drows, off))
        off += drows
        Hp = hid
    return levels, off


_FALLBACK_WARNED = set()


def _warn_if_fp32_fallback(what, probe, H, Hp, m):
    """cin_math 1 / 2 asked for the f16 / bf16 MFMA kernels; a level they do not cover (forward: odd or > 40 field
    count, or H <= 32 / 64 rows) runs in the fp32-MFMA kernel -- same results, about a third of the rate.  Said once
    per shape instead of silently (the library records the arithmetic of its last launch in the probe option)."""
    want = _lib.get_option("cin_math")
    if want == 0 or _lib.get_option(probe) == want:
        return
    key = (what, want, H, Hp, m)
    if key in _FALLBACK_WARNED:
        return
    _FALLBACK_WARNED.add(key)
    import warnings
    warnings.warn("xdfm: CIN %s level (H=%d, H_prev=%d, fields=%d) has no %s kernel; it runs in fp32-MFMA arithmetic"
                  % (what, H, Hp, m, "f16x3" if want == 1 else "bf16"), RuntimeWarning, stacklevel=3)


class CINStack(torch.autograd.Function):
    """All CIN levels.  x0 is FM layout [m, B*D].

    pool == "sum": returns [B, featuremap_num]   (deepctr/layers/interaction.py:207-248)
    pool == "fm" : returns the direct-connect feature maps, FM layout [featuremap_num, B*D]
                   (the tensor deepctr/layers/cin_attention.py:292 feeds to the attention block).
    """

    @staticmethod
    def forward(ctx, x0, B, D, layer_size, split_half, act, pool, *params):
        _need_cuda(x0, "cin input")
        lib = _lib.load()
        m = x0.shape[0]
        N = B * D
        assert x0.shape[1] == N and x0.is_contiguous()
        levels, fm = cin_geometry(m, layer_size, split_half)
        dev = x0.device
        outs: List[torch.Tensor] = []
        xp = x0
        result = torch.empty((B, fm), dtype=torch.float32, device=dev) if pool == "sum" else None
        # The packed weights depend on the weights only.  When every level runs the f16x3 kernels both ways, all
        # forward packs and (if a backward will follow) all dX packs are produced by two launches up front.
        W2s = []
        for l, (H, Hp, hid, dir0, drows, off) in enumerate(levels):
            _need_cuda(params[2 * l], "cin weight")
            W2s.append(params[2 * l].reshape(H, Hp * m).contiguous())
        wfs, wzs = [None] * len(levels), None
        if len(levels) <= 8 and all(lib.xdfm_cin_pack_all_supported(H, Hp, m) for (H, Hp, *_r) in levels):
            need_bwd = any(ctx.needs_input_grad)
            jobs = (_lib.PackJob * len(levels))()
            wzs = [None] * len(levels)
            for l, (H, Hp, *_r) in enumerate(levels):
                wfs[l] = torch.empty(lib.xdfm_cin_fwd_pack_elems(H, Hp, m), dtype=torch.float32, device=dev)
                if need_bwd:
                    wzs[l] = torch.empty(lib.xdfm_cin_bwd_pack_elems(H, Hp, m), dtype=torch.float32, device=dev)
                jobs[l].W, jobs[l].H, jobs[l].Hp, jobs[l].m = W2s[l].data_ptr(), H, Hp, m
                jobs[l].fwd_pack = wfs[l].data_ptr()
                jobs[l].bwd_pack = wzs[l].data_ptr() if need_bwd else None
            _lib.check(lib.xdfm_cin_pack_all(ctypes.cast(jobs, ctypes.c_void_p), len(levels), _stream()), "cin_pack_all")
            if not need_bwd:
                wzs = None
        # "lean" levels (sum pooling, f16x3 / bf16 forward, D in {4, 8, 16, 32}): the forward's epilogue sums the direct-connect
        # rows into `result` and leaves the ReLU sign bits, so only the hidden rows (the next level's x_prev) are ever
        # written -- no direct_sum launches, no direct-connect half in memory, and the backward reads 1 bit per element
        # instead of the saved output (xdfm_cin_level_fwd_ex / xdfm_cin_bwd_prep)
        lean = pool == "sum" and os.environ.get("XDFM_CIN_LEAN", "1") != "0" and \
            all(lib.xdfm_cin_level_fwd_ex_supported(H, Hp, m, D) for (H, Hp, *_r) in levels)
        need_bwd = any(ctx.needs_input_grad)
        masks = []
        mask_chunks = (N + 31) // 32
        for l, (H, Hp, hid, dir0, drows, off) in enumerate(levels):
            W, bias = params[2 * l], params[2 * l + 1]
            W2 = W2s[l]
            wf = wfs[l]
            if wf is None:
                wf = torch.empty(lib.xdfm_cin_fwd_pack_elems(H, Hp, m), dtype=torch.float32, device=dev)
                _lib.check(lib.xdfm_cin_fwd_pack(_ptr(W2), H, Hp, m, _ptr(wf), _stream()), "cin_fwd_pack")
            bias_c = bias.contiguous()
            if lean:
                A = torch.empty((hid, N), dtype=torch.float32, device=dev) if hid > 0 else None
                mask_ld = (H + 3) // 4 * 4
                mk = torch.empty((mask_chunks, mask_ld), dtype=torch.int32, device=dev) if need_bwd else None    # [chunk][row]
                _lib.check(_run("cin_level_fwd", 2.0 * H * Hp * m * N, lambda: lib.xdfm_cin_level_fwd_ex(
                    _ptr(xp), _ptr(x0), _ptr(wf), _ptr(bias_c), H, Hp, m, N, act, _ptr(A), hid, _ptr(result), fm, off, dir0, D,
                    _ptr(mk), mask_ld, _stream())), "cin_level_fwd")
                masks.append(mk)
                outs.append(A if A is not None else x0[:0])      # a placeholder keeps the saved-tensor list regular
                xp = A
                continue
            A = torch.empty((H, N), dtype=torch.float32, device=dev)
            _lib.check(_run("cin_level_fwd", 2.0 * H * Hp * m * N, lambda: lib.xdfm_cin_level_fwd(
                _ptr(xp), _ptr(x0), _ptr(wf), _ptr(bias_c), H, Hp, m, N, act, _ptr(A), _stream())), "cin_level_fwd")
            _warn_if_fp32_fallback("forward", "last_fwd_kernel", H, Hp, m)
            if pool == "sum":
                _lib.check(lib.xdfm_cin_direct_sum(_ptr(A), dir0, drows, B, D, _ptr(result), fm, off, _stream()),
                           "cin_direct_sum")
            outs.append(A)
            xp = A[:hid] if hid > 0 else None
        ctx.cfg = (B, D, tuple(layer_size), split_half, act, pool, m)
        ctx.wzs = wzs                                # dX packs made up front (or None), arithmetic mode they belong to
        ctx.cin_math = _lib.get_option("cin_math")
        ctx.lean = lean and need_bwd
        if ctx.lean:
            ctx.save_for_backward(x0, *outs, *params, *masks)
        else:
            ctx.save_for_backward(x0, *outs, *params)
        if pool == "sum":
            return result
        return torch.cat([A[dir0:dir0 + drows] for A, (H, Hp, hid, dir0, drows, off) in zip(outs, levels)], dim=0)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, D, layer_size, split_half, act, pool, m = ctx.cfg
        L = len(layer_size)
        saved = ctx.saved_tensors
        x0, outs, params = saved[0], saved[1:1 + L], saved[1 + L:1 + 3 * L]
        masks = saved[1 + 3 * L:] if ctx.lean else None        # lean levels: outs[l] holds the hidden rows only, the ReLU mask is bits
        N = B * D
        dev = x0.device
        levels, fm = cin_geometry(m, layer_size, split_half)
        g = g.contiguous()
        dx0 = torch.empty((m, N), dtype=torch.float32, device=dev)      # first dX launch stores, later ones add
        dx0_set = False
        grads = [None] * (2 * L)
        dhid = None                                  # gradient w.r.t. this level's hidden rows
        dbias_all = torch.zeros(sum(lv[0] for lv in levels), dtype=torch.float32, device=dev)   # one fill for all levels
        dbias_off = [sum(lv[0] for lv in levels[:k]) for k in range(L)]
        for l in range(L - 1, -1, -1):
            H, Hp, hid, dir0, drows, off = levels[l]
            A = None if ctx.lean else outs[l]
            xp = x0 if l == 0 else outs[l - 1][:levels[l - 1][2]]
            W, bias = params[2 * l], params[2 * l + 1]
            mk = masks[l] if ctx.lean else None
            # lean levels with f16x3 / bf16 dW and dX kernels: dOut is never materialised in fp32 -- the dW kernel takes its
            # fp16 planes, the dX kernel forms its operand from dOut's sources (sign bits, dhid, pooled gradient) itself
            no_dout = mk is not None and bool(lib.xdfm_cin_bwd_nodout_supported(H, Hp, m, B, D)) and \
                os.environ.get("XDFM_CIN_NODOUT", "1") != "0"
            dOut = None if no_dout else torch.empty((H, N), dtype=torch.float32, device=dev)
            dbias = dbias_all[dbias_off[l]:dbias_off[l] + H]
            has_hid = dhid is not None and hid > 0
            need_dw = ctx.needs_input_grad[7 + 2 * l]
            # dOut (+ dbias), and in the same pass -- when the f16x3 / bf16 dW kernel will run for this level -- the fp16
            # planes of dOut and the dW kernel's scales, straight into the dW workspace (xdfm_cin_bwd_prep)
            dws = torch.empty(lib.xdfm_cin_bwd_prep_ws_elems(H, Hp, m, B, D), dtype=torch.float32, device=dev)   # fixed-order dbias
            ws = torch.empty(lib.xdfm_cin_bwd_w_ws_elems(H, Hp, m, N), dtype=torch.float32, device=dev) if need_dw else None
            prepared = ctypes.c_int(0)
            _lib.check(_run("cin_dout", 0.0, lambda: lib.xdfm_cin_bwd_prep(
                _ptr(A), _ptr(mk), mk.shape[1] if mk is not None else 0, H, B, D, act, _ptr(dhid) if has_hid else None, 0, hid if has_hid else 0, _ptr(g),
                0 if pool == "sum" else 1, fm if pool == "sum" else N, off, dir0, drows, _ptr(dOut), _ptr(dbias), _ptr(dws),
                _ptr(xp), _ptr(x0), Hp, m, _ptr(ws), ctypes.byref(prepared), _stream())), "cin_dout")
            if need_dw:
                dW = torch.empty((H, Hp * m), dtype=torch.float32, device=dev)
                if prepared.value:
                    call = lambda: lib.xdfm_cin_level_bwd_w_prepared(_ptr(dOut), _ptr(xp), _ptr(x0), H, Hp, m, N, _ptr(ws), _ptr(dW),
                                                                      _stream())
                else:
                    call = lambda: lib.xdfm_cin_level_bwd_w(_ptr(dOut), _ptr(xp), _ptr(x0), H, Hp, m, N, _ptr(ws), _ptr(dW),
                                                             _stream())
                if PROFILE is not None and _lib.get_option("cin_math") in (1, 2):
                    # per-kernel timing: run the call's three phases separately, events around the MFMA kernel only
                    # (a shape without an f16x3 dW kernel ignores the knob: its whole call then runs once, in phase 2)
                    try:
                        _lib.set_option("bww_phase", 1)
                        _lib.check(_run("cin_level_bwd_w passes", 0.0, call), "cin_level_bwd_w")
                        _lib.set_option("bww_phase", 2)
                        _lib.check(_run("cin_level_bwd_w", 2.0 * H * Hp * m * N, call), "cin_level_bwd_w")
                        _lib.set_option("bww_phase", 3)
                        _lib.check(_run("cin_level_bwd_w passes", 0.0, call), "cin_level_bwd_w")
                    finally:
                        _lib.set_option("bww_phase", 0)
                else:
                    _lib.check(_run("cin_level_bwd_w", 2.0 * H * Hp * m * N, call), "cin_level_bwd_w")
                grads[2 * l] = dW.view(W.shape)
            if ctx.needs_input_grad[8 + 2 * l]:
                grads[2 * l + 1] = dbias
            # dX: H <= 256 rows of the contraction per launch
            dxp = torch.empty((Hp, N), dtype=torch.float32, device=dev)
            W2 = W.reshape(H, Hp * m)
            hstep = _lib.get_option("x3_bwx_rows") or 256          # rows of the contraction per launch (<= 256)
            prepacked = ctx.wzs is not None and ctx.wzs[l] is not None and hstep >= H and \
                ctx.cin_math == _lib.get_option("cin_math")
            for h0 in range(0, H, hstep):
                hc = min(hstep, H - h0)
                if prepacked:
                    wz = ctx.wzs[l]
                else:
                    wz = torch.empty(lib.xdfm_cin_bwd_pack_elems(hc, Hp, m), dtype=torch.float32, device=dev)
                    wc = W2[h0:h0 + hc].contiguous()
                    _lib.check(lib.xdfm_cin_bwd_pack(_ptr(wc), hc, Hp, m, _ptr(wz), _stream()), "cin_bwd_pack")
                flags = (1 if h0 == 0 else 0) | (0 if dx0_set else 2)     # XDFM_BWX_SET_DXP | XDFM_BWX_SET_DX0
                if no_dout:
                    _lib.check(_run("cin_level_bwd_x", 2.0 * hc * Hp * m * N, lambda: lib.xdfm_cin_level_bwd_x_src(
                        _ptr(mk) if act == 1 else None, mk.shape[1], _ptr(dhid) if has_hid else None, hid if has_hid else 0, _ptr(g),
                        0 if pool == "sum" else 1, fm if pool == "sum" else N, off, dir0, drows, D, h0, _ptr(xp), _ptr(x0), _ptr(wz),
                        hc, Hp, m, N, _ptr(dxp), _ptr(dx0), flags, _stream())), "cin_level_bwd_x")
                else:
                    dOc = dOut[h0:h0 + hc]
                    _lib.check(_run("cin_level_bwd_x", 2.0 * hc * Hp * m * N, lambda: lib.xdfm_cin_level_bwd_x_ex(
                        _ptr(dOc), _ptr(xp), _ptr(x0), _ptr(wz), hc, Hp, m, N, _ptr(dxp), _ptr(dx0), flags, _stream())),
                        "cin_level_bwd_x")
                dx0_set = True
            if l == 0 and not all(lib.xdfm_cin_bwd_x_is_folded(min(hstep, H - h0), Hp, m, 1) for h0 in range(0, H, hstep)):
                dx0 += dxp                               # x_prev of level 0 is x0 itself (the folded dX kernel has put the
                #                                          whole gradient into dx0 already: include/xdfm.h, option x3_sym)
            dhid = dxp
        return (dx0, None, None, None, None, None, None) + tuple(grads)


def cin_stack(x0_fm, B, D, layer_size, split_half, activation, pool, weights, biases):
    act = activation_code(activation)
    params = []
    for w, b in zip(weights, biases):
        params += [w, b]
    return CINStack.apply(x0_fm, B, D, tuple(layer_size), bool(split_half), act, pool, *params)


def to_fm_layout(x: torch.Tensor) -> torch.Tensor:
    """[B, R, D] -> FM layout [R, B*D] (a transpose; used when a caller hands the reference layout)."""
    B, R, D = x.shape
    return x.permute(1, 0, 2).reshape(R, B * D).contiguous()


def from_fm_layout(t: torch.Tensor, B: int, D: int) -> torch.Tensor:
    """FM layout [R, B*D] -> [B, R, D]."""
    R = t.shape[0]
    return t.view(R, B, D).permute(1, 0, 2)


# --------------------------------------------------------------------------------------------- #
# attention pooling                                                                              #
# --------------------------------------------------------------------------------------------- #
class AttnPool(torch.autograd.Function):
    """FM-layout feature maps [S, B*D] -> pooled [B, D] through n_layers x (MHSA, +residual, LayerNorm)
    and the attention pooling (deepctr/layers/cin_attention.py:63-144, :302-313, :452-464).
    params: per layer Wq, Wk, Wv, Wo (+ gamma, beta when use_ln), then W1, b1, w2.
    p_drop > 0: attention dropout (cin_attention.py:86); the seed of the keep mask is drawn on the device from
    torch's generator (torch.manual_seed makes a run repeatable, a captured graph draws a new one per replay)
    and kept for backward, which regenerates the mask instead of storing it."""

    @staticmethod
    def forward(ctx, fm, B, D, nh, n_layers, use_ln, use_res, p_drop, *params):
        _need_cuda(fm, "feature maps")
        lib = _lib.load()
        S = fm.shape[0]
        assert fm.shape[1] == B * D and fm.is_contiguous()
        theta = torch.cat([p.reshape(-1) for p in params])
        assert theta.numel() == lib.xdfm_cin_attn_theta_elems(D, n_layers, int(use_ln))
        dev = fm.device
        out = torch.empty((B, D), dtype=torch.float32, device=dev)
        tok = torch.empty((n_layers, B, S, D), dtype=torch.float32, device=dev)
        osv = torch.empty((n_layers, B, S, D), dtype=torch.float32, device=dev)
        ml = torch.empty((n_layers, B, S, nh, 2), dtype=torch.float32, device=dev)
        # S^2 * (3 D FMA) per example and layer, counted as 2 FLOP per FMA
        flops = 2.0 * 3 * D * S * S * B * n_layers
        p_drop = float(p_drop)
        seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=dev) if p_drop > 0 else None
        _lib.check(_run("cin_attn_pool_fwd", flops, lambda: lib.xdfm_cin_attn_pool_fwd(
            _ptr(fm), B, S, D, nh, n_layers, int(use_ln), int(use_res), _ptr(theta), _ptr(out), _ptr(tok), _ptr(osv),
            _ptr(ml), p_drop, _ptr(seed) if seed is not None else None, _stream())), "cin_attn_pool_fwd")
        ctx.cfg = (B, D, nh, n_layers, use_ln, use_res, p_drop, [tuple(p.shape) for p in params])
        ctx.drop_seed = seed
        AttnPool.last_drop_seed = seed          # read by the parity tests (xdfm_cin_attn_dropout_mask)
        ctx.save_for_backward(fm, theta, tok, osv, ml)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        fm, theta, tok, osv, ml = ctx.saved_tensors
        B, D, nh, n_layers, use_ln, use_res, p_drop, shapes = ctx.cfg
        seed = ctx.drop_seed
        S = fm.shape[0]
        dfm = torch.empty_like(fm)
        dtheta = torch.empty_like(theta)         # overwritten: per-workgroup shares + a fixed-order sum, no float atomics
        ws = torch.empty(lib.xdfm_cin_attn_pool_bwd_ws_elems(B, D, n_layers, int(use_ln)), dtype=torch.float32, device=fm.device)
        dout = dout.contiguous()
        flops = 2.0 * 7 * D * S * S * B * n_layers
        _lib.check(_run("cin_attn_pool_bwd", flops, lambda: lib.xdfm_cin_attn_pool_bwd_det(
            _ptr(fm), B, S, D, nh, n_layers, int(use_ln), int(use_res), _ptr(theta), _ptr(tok), _ptr(osv), _ptr(ml),
            _ptr(dout), _ptr(dfm), _ptr(dtheta), _ptr(ws), p_drop, _ptr(seed) if seed is not None else None, _stream())),
            "cin_attn_pool_bwd")
        grads, off = [], 0
        for sh in shapes:
            n = 1
            for k in sh:
                n *= k
            grads.append(dtheta[off:off + n].view(sh))
            off += n
        return (dfm, None, None, None, None, None, None, None) + tuple(grads)


def attn_pool(fm, B, D, mhsa_layers, layer_norms, pooling, use_res):
    """mhsa_layers: modules with W_q/W_k/W_v/W_o (nn.Linear, bias-free) and .num_heads; layer_norms:
    matching nn.LayerNorm list or None; pooling: module whose .attention is Sequential(Linear, Tanh,
    Linear(bias=False))."""
    params = []
    for l, att in enumerate(mhsa_layers):
        params += [att.W_q.weight, att.W_k.weight, att.W_v.weight, att.W_o.weight]
        if layer_norms is not None:
            params += [layer_norms[l].weight, layer_norms[l].bias]
    params += [pooling.attention[0].weight, pooling.attention[0].bias, pooling.attention[2].weight]
    nh = mhsa_layers[0].num_heads
    att0 = mhsa_layers[0]
    p_drop = float(att0.dropout.p) if att0.training else 0.0      # every layer is built with the same attn_dropout
    if any(float(a.dropout.p) != float(att0.dropout.p) for a in mhsa_layers):
        raise ValueError("cin_attn_pool: the attention layers must share one dropout rate")
    if p_drop >= 1.0:
        raise ValueError("cin_attn_pool: attention dropout must be < 1, got %g" % p_drop)
    return AttnPool.apply(fm, B, D, nh, len(mhsa_layers), layer_norms is not None, bool(use_res), p_drop, *params)


# --------------------------------------------------------------------------------------------- #
# dense layer                                                                                    #
# --------------------------------------------------------------------------------------------- #
class Dense(torch.autograd.Function):
    """y = x W^T + b (deepctr/layers/core.py:120-134).  The three GEMMs stay with hipBLASLt; the bias
    gradient is the library's own column sum, because ATen's `sum(0)` zeroes a semaphore buffer with
    hipMemsetAsync and a memset node inside a captured HIP graph is not ordered reliably on this stack
    (tools/graph_memset_probe.py) -- with it the train step contains no memset at all."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return torch.nn.functional.linear(x, W, b)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous()
        dx = g.mm(W) if ctx.needs_input_grad[0] else None
        dW = g.t().mm(x) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            lib = _lib.load()
            rows, cols = g.shape
            ws = torch.empty(lib.xdfm_colsum_ws_elems(cols), dtype=torch.float32, device=g.device)
            db = torch.empty(cols, dtype=torch.float32, device=g.device)
            _lib.check(lib.xdfm_colsum(_ptr(g), rows, cols, g.stride(0), _ptr(ws), _ptr(db), _stream()), "colsum")
        return dx, dW, db


class DenseReLU(torch.autograd.Function):
    """y = relu(x W^T + b): the ReLU rides in the GEMM's epilogue (hipBLASLt, `torch._addmm_activation`), and its
    backward mask is applied by the pass that sums the bias gradient (`xdfm_relu_bwd_colsum`) -- two launches fewer
    each way per hidden layer than linear + nn.ReLU (deepctr/layers/core.py:120-134)."""

    @staticmethod
    def forward(ctx, x, W, b):
        y = torch._addmm_activation(b, x, W.t(), use_gelu=False)
        ctx.save_for_backward(x, W, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, y = ctx.saved_tensors
        lib = _lib.load()
        g = g.contiguous()
        rows, cols = g.shape
        ws = torch.empty(lib.xdfm_colsum_ws_elems(cols), dtype=torch.float32, device=g.device)
        gz = torch.empty_like(g)
        db = torch.empty(cols, dtype=torch.float32, device=g.device)
        _lib.check(lib.xdfm_relu_bwd_colsum(_ptr(g), _ptr(y), rows, cols, g.stride(0), y.stride(0), _ptr(ws), _ptr(gz),
                                            _ptr(db), _stream()), "relu_bwd_colsum")
        dx = gz.mm(W) if ctx.needs_input_grad[0] else None
        dW = gz.t().mm(x) if ctx.needs_input_grad[1] else None
        return dx, dW, db if ctx.needs_input_grad[2] else None


def dense_relu(x, W, b):
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and b is not None and x.is_contiguous():
        return DenseReLU.apply(x, W, b)
    return torch.relu(dense(x, W, b))


def dense(x, W, b):
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
        return Dense.apply(x, W, b)
    return torch.nn.functional.linear(x, W, b)


# --------------------------------------------------------------------------------------------- #
# output head                                                                                    #
# --------------------------------------------------------------------------------------------- #
class Head(torch.autograd.Function):
    """(y [B,1], bias [1]|None, lin [B,1]|None, u [B,Ku]|None, wu [1,Ku]|None, v [B,Kv]|None, wv [1,Kv]|None) ->
    (pred [B], loss [1]) with pred = sigmoid(lin + u wu^T + v wv^T + bias), loss = sum BCE(pred, y): cin_linear /
    dnn_linear + logit sum (deepctr/models/xdeepfm.py:95-105) + PredictionLayer (core.py:150-160) +
    F.binary_cross_entropy(reduction='sum') (basemodel.py:254), two launches each way (K8).  `pred` is returned
    for metrics and is not differentiable here (the model's train step differentiates the loss only)."""

    @staticmethod
    def forward(ctx, y, bias, lin, u, wu, v, wv):
        lib = _lib.load()
        yv = y.reshape(-1).contiguous()
        B = yv.numel()
        dev = yv.device
        linv = lin.reshape(-1).contiguous() if lin is not None else None
        u = u.contiguous() if u is not None else None
        v = v.contiguous() if v is not None else None
        wu = wu.reshape(-1).contiguous() if wu is not None else None
        wv = wv.reshape(-1).contiguous() if wv is not None else None
        Ku = u.shape[1] if u is not None else 0
        Kv = v.shape[1] if v is not None else 0
        for t, k in ((u, Ku), (v, Kv)):
            if t is not None and (t.dim() != 2 or t.shape[0] != B):
                raise ValueError("xdfm head: operands must be [B, K]")
        pred = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.xdfm_head_ws_elems(Ku, Kv), dtype=torch.float32, device=dev)
        _lib.check(lib.xdfm_head_fwd(_ptr(linv), _ptr(u), _ptr(wu), Ku, _ptr(v), _ptr(wv), Kv, _ptr(bias), _ptr(yv), B,
                                     _ptr(pred), _ptr(loss), _ptr(ws), _stream()), "head_fwd")
        ctx.save_for_backward(pred, yv, u, wu, v, wv)
        ctx.cfg = (bias is not None, lin is not None, tuple(lin.shape) if lin is not None else None, Ku, Kv)
        ctx.mark_non_differentiable(pred)
        ctx.set_materialize_grads(False)          # no [B] zero fill for pred's (unused) gradient slot
        return pred, loss

    @staticmethod
    def backward(ctx, _gpred, gloss):
        if gloss is None:
            return (None,) * 7
        lib = _lib.load()
        pred, yv, u, wu, v, wv = ctx.saved_tensors
        has_bias, has_lin, lin_shape, Ku, Kv = ctx.cfg
        B = pred.numel()
        dev = pred.device
        f32 = dict(dtype=torch.float32, device=dev)
        dlin = torch.empty(B, **f32) if has_lin else None
        du = torch.empty((B, Ku), **f32) if u is not None else None
        dv = torch.empty((B, Kv), **f32) if v is not None else None
        grads = torch.empty(Ku + Kv + 1, **f32)
        ws = torch.empty(lib.xdfm_head_ws_elems(Ku, Kv), **f32)
        gl = gloss.reshape(1).contiguous()
        _lib.check(lib.xdfm_head_bwd(_ptr(pred), _ptr(yv), _ptr(gl), _ptr(u), _ptr(wu), Ku, _ptr(v), _ptr(wv), Kv, B,
                                     _ptr(dlin), _ptr(du), _ptr(dv), _ptr(grads), _ptr(ws), _stream()), "head_bwd")
        return (None, grads[Ku + Kv:Ku + Kv + 1] if has_bias else None,
                dlin.view(lin_shape) if has_lin else None,
                du, grads[:Ku].view(1, Ku) if u is not None else None,
                dv, grads[Ku:Ku + Kv].view(1, Kv) if v is not None else None)


# --------------------------------------------------------------------------------------------- #
# vocabulary-wide softmax cross-entropy (SFG decoder heads)                                       #
# --------------------------------------------------------------------------------------------- #
VOCAB_TILE_BYTES = 256 << 20        # logits of one vocabulary tile kept at a time ([rows, tile] fp32)


class VocabSoftmaxCE(torch.autograd.Function):
    """ce[r] = logsumexp_v(h_r . W_v + b_v) - (h_r . W_t + b_t), t = target[r]: nn.Linear(K, V) followed by
    F.cross_entropy(reduction='none') (deepctr/xdeepfm_pro/sfg_decoder.py:146-149, :277-283) WITHOUT the [rows, V]
    logits: the vocabulary is walked in tiles with an online log-sum-exp, and the backward recomputes each tile
    (softmax - onehot) for the three products.  At Criteo-scale vocabularies the logits of one field would be
    4096 x 10 M x 4 B = 164 GB.  The tile GEMMs are library GEMMs (hipBLASLt); everything stays on the device.
    hidden [rows, K], W [V, K], b [V], target [rows] (ids as float or integer, truncated like Tensor.long())."""

    @staticmethod
    def _tile(rows, V):
        return int(max(1024, min(V, VOCAB_TILE_BYTES // (4 * max(rows, 1)))))

    @staticmethod
    def forward(ctx, hidden, W, b, target):
        rows, V = hidden.shape[0], W.shape[0]
        tgt = target.long()
        T = VocabSoftmaxCE._tile(rows, V)
        m = torch.full((rows,), float("-inf"), dtype=torch.float32, device=hidden.device)
        ssum = torch.zeros(rows, dtype=torch.float32, device=hidden.device)
        lib = _lib.load()
        for v0 in range(0, V, T):
            z = torch.addmm(b[v0:v0 + T], hidden, W[v0:v0 + T].t())
            # one read of the tile: row maxima, then sum of exp against the running maximum (m, ssum updated in place)
            if rows > 0:
                _lib.check(lib.xdfm_vocab_lse_update(_ptr(z), z.stride(0), rows, z.shape[1], _ptr(m), _ptr(ssum), _stream()),
                           "vocab_lse_update")
        lse = m + torch.log(ssum)
        zt = (hidden * W.index_select(0, tgt)).sum(dim=1) + b.index_select(0, tgt)
        ctx.save_for_backward(hidden, W, b, tgt, lse)
        return lse - zt

    @staticmethod
    def backward(ctx, g):
        hidden, W, b, tgt, lse = ctx.saved_tensors
        rows, V = hidden.shape[0], W.shape[0]
        T = VocabSoftmaxCE._tile(rows, V)
        g = g.contiguous()
        dW = torch.empty_like(W) if ctx.needs_input_grad[1] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[2] else None
        dh = torch.zeros_like(hidden) if ctx.needs_input_grad[0] else None
        for v0 in range(0, V, T):
            dz = torch.addmm(b[v0:v0 + T], hidden, W[v0:v0 + T].t())
            if rows > 0:                                                 # g * softmax, in place
                _lib.check(_lib.load().xdfm_vocab_softmax_grad(_ptr(dz), dz.stride(0), rows, dz.shape[1], _ptr(lse), _ptr(g),
                                                               _stream()), "vocab_softmax_grad")
            if dW is not None:
                torch.mm(dz.t(), hidden, out=dW[v0:v0 + T])
            if db is not None:
                torch.sum(dz, dim=0, out=db[v0:v0 + T])
            if dh is not None:
                dh.addmm_(dz, W[v0:v0 + T])
        # - g * onehot(target)
        if dW is not None:
            dW.index_add_(0, tgt, -(g[:, None] * hidden))
        if db is not None:
            db.index_add_(0, tgt, -g)
        if dh is not None:
            dh.sub_(g[:, None] * W.index_select(0, tgt))
        return dh, dW, db, None


def vocab_softmax_ce(hidden, W, b, target):
    _need_cuda(hidden, "decoder hidden layer")
    return VocabSoftmaxCE.apply(hidden, W, b, target)


_VCE_FIELD = np.dtype([("W", "u8"), ("bias", "u8"), ("dW", "u8"), ("db", "u8"), ("ws_off", "i8"), ("V", "i4"), ("vr", "i4"),
                       ("item0", "i4"), ("blk0", "i4")])            # xdfm_vce_field (include/xdfm.h)
_VCE_ITEM = np.dtype([("field", "i4"), ("sb0", "i4"), ("sb1", "i4"), ("range", "i4")])      # xdfm_vce_item
_VCE_PLANS = {}


def _vce_plan(R, K, vocabs, dev):
    """Host tables of xdfm_vocab_ce_plan for (R, K, vocabularies) + the items on the device (they hold no pointers)."""
    key = (R, K, tuple(vocabs), dev)
    plan = _VCE_PLANS.get(key)
    if plan is None:
        lib = _lib.load()
        V = np.asarray(vocabs, dtype=np.int32)
        fields = np.zeros(len(vocabs), dtype=_VCE_FIELD)
        ws, nblk = ctypes.c_long(0), ctypes.c_int(0)
        n = lib.xdfm_vocab_ce_plan(len(vocabs), V.ctypes.data, R, K, fields.ctypes.data, None, 0, ctypes.byref(ws), ctypes.byref(nblk))
        if n <= 0:
            raise _lib.XdfmError("vocab_ce_plan: " + lib.xdfm_last_error().decode("utf-8", "replace"))
        items = np.zeros(n, dtype=_VCE_ITEM)
        lib.xdfm_vocab_ce_plan(len(vocabs), V.ctypes.data, R, K, fields.ctypes.data, items.ctypes.data, n, ctypes.byref(ws),
                               ctypes.byref(nblk))
        items_dev = torch.from_numpy(items.view(np.uint8)).to(dev)
        if len(_VCE_PLANS) > 64:
            _VCE_PLANS.clear()
        plan = _VCE_PLANS[key] = (fields, items_dev, int(n), int(ws.value), int(nblk.value))
    return plan


def _vce_fields(plan, Ws, bs, dWs, dbs, dev):
    fields = plan[0].copy()
    fields["W"] = [w.data_ptr() for w in Ws]
    fields["bias"] = [b.data_ptr() for b in bs]
    fields["dW"] = [0 if t is None else t.data_ptr() for t in dWs]
    fields["db"] = [0 if t is None else t.data_ptr() for t in dbs]
    return torch.from_numpy(fields.view(np.uint8)).to(dev)


class VocabHeadsCE(torch.autograd.Function):
    """ce[f][r] of ALL sparse fields' heads over the same hidden rows, logits never in HBM (csrc/vocab_ce_x3.hip; the
    heads of deepctr/xdeepfm_pro/sfg_decoder.py:146-149 + F.cross_entropy of :277-283).  hidden [R, K] (K = 32 or 64),
    targets [F, R] int64, then the F weights [V_f, K] and the F biases [V_f].  The hidden layer is packed into MFMA
    fragments once per pass; all fields share each launch (a table of per-field pointers and work items): log-sum-exp
    partials + merge in the forward; in the backward the hidden gradient (partial slabs summed in a fixed order) and
    the weight / bias gradients (every row written once, the target's -g included)."""

    calls = 0                 # forward passes taken (tests assert that a golden reached this path)

    @staticmethod
    def forward(ctx, hidden, targets, *params):
        lib = _lib.load()
        VocabHeadsCE.calls += 1
        F_ = len(params) // 2
        Ws, bs = params[:F_], params[F_:]
        R, K = hidden.shape
        dev = hidden.device
        hidden = hidden.contiguous()
        targets = targets.contiguous()
        for t in params:
            if not t.is_contiguous():
                raise ValueError("xdfm: head parameters must be contiguous")
        plan = _vce_plan(R, K, [w.shape[0] for w in Ws], dev)
        fields = _vce_fields(plan, Ws, bs, [None] * F_, [None] * F_, dev)
        pack = torch.empty(lib.xdfm_vocab_ce_pack_elems(R, K), dtype=torch.float32, device=dev)
        _lib.check(lib.xdfm_vocab_ce_pack_hidden(_ptr(hidden), K, R, K, _ptr(pack), _stream()), "vocab_ce_pack_hidden")
        Rpad = lib.xdfm_vocab_ce_rows_padded(R)
        ce = torch.empty(F_, R, dtype=torch.float32, device=dev)
        lse2 = torch.zeros(F_, Rpad, dtype=torch.float32, device=dev)
        wmax = torch.empty(F_, dtype=torch.int32, device=dev)
        ws = torch.empty(plan[3], dtype=torch.float32, device=dev)
        flops = 2.0 * R * K * sum(w.shape[0] for w in Ws)                  # of the reference's nn.Linear, all heads
        _lib.check(_run("vocab_ce_fwd", flops, lambda: lib.xdfm_vocab_ce_fwd(
            _ptr(pack), _ptr(hidden), K, R, K, _ptr(fields), F_, _ptr(plan[1]), plan[2], _ptr(targets), _ptr(ws), _ptr(ce), _ptr(lse2),
            _ptr(wmax), _stream())), "vocab_ce_fwd")
        ctx.save_for_backward(hidden, targets, pack, lse2, wmax, *params)
        ctx.plan = plan
        return ce

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        hidden, targets, pack, lse2, wmax = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        F_ = len(params) // 2
        Ws, bs = params[:F_], params[F_:]
        R, K = hidden.shape
        dev = hidden.device
        plan = ctx.plan
        g = g.contiguous()
        Rpad = lib.xdfm_vocab_ce_rows_padded(R)
        gpack = torch.empty(F_ * (4 + Rpad), dtype=torch.float32, device=dev)
        flops = 2.0 * R * K * sum(w.shape[0] for w in Ws)     # per product of the reference's backward (dH; dW), recompute not counted
        _lib.check(lib.xdfm_vocab_ce_pack_g(_ptr(g), F_, R, _ptr(gpack), _stream()), "vocab_ce_pack_g")
        dWs = [torch.empty_like(Ws[f]) if ctx.needs_input_grad[2 + f] else None for f in range(F_)]
        dbs = [torch.empty_like(bs[f]) if ctx.needs_input_grad[2 + F_ + f] else None for f in range(F_)]
        fields = _vce_fields(plan, Ws, bs, dWs, dbs, dev)
        dh = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(hidden)
            ws = torch.empty(plan[3], dtype=torch.float32, device=dev)
            _lib.check(_run("vocab_ce_bwd_h", flops, lambda: lib.xdfm_vocab_ce_bwd_h(
                _ptr(pack), R, K, _ptr(fields), F_, _ptr(plan[1]), plan[2], _ptr(targets), _ptr(g), _ptr(gpack), _ptr(lse2), _ptr(wmax),
                _ptr(ws), _ptr(dh), K, _stream())), "vocab_ce_bwd_h")
        if any(t is not None for t in dWs + dbs):
            _lib.check(_run("vocab_ce_bwd_w", flops, lambda: lib.xdfm_vocab_ce_bwd_w(
                _ptr(pack), R, K, _ptr(fields), F_, plan[4], _ptr(targets), _ptr(gpack), _ptr(lse2), _ptr(wmax), _stream())),
                "vocab_ce_bwd_w")
        return (dh, None, *dWs, *dbs)


def vocab_heads_ce_supported(K: int) -> bool:
    return bool(_lib.load().xdfm_vocab_ce_x3_supported(int(K)))


def vocab_heads_ce(hidden, targets, weights, biases):
    """[F, R] cross-entropies of F heads (weights[f] [V_f, K], biases[f] [V_f]) over hidden [R, K], targets [F, R] int64."""
    _need_cuda(hidden, "decoder hidden layer")
    return VocabHeadsCE.apply(hidden, targets, *weights, *biases)


# --------------------------------------------------------------------------------------------- #
# L2 regulariser                                                                                 #
# --------------------------------------------------------------------------------------------- #
class L2Plan:
    """Device-side description (pointers, sizes, strengths) of the tensors one L2 term covers."""

    def __init__(self, coeffs: Sequence[float]):
        self.coeffs = [float(c) for c in coeffs]
        self._key = None
        self._dev = None

    def tables(self, tensors: Sequence[torch.Tensor]):
        key = tuple((t.data_ptr(), t.numel()) for t in tensors)
        if key != self._key:
            dev = tensors[0].device
            offs, off = [], 0
            for k in key:
                offs.append(off)
                off += k[1]
            self._dev = (torch.tensor([k[0] for k in key], dtype=torch.int64, device=dev),
                         torch.tensor([k[1] for k in key], dtype=torch.int64, device=dev),
                         torch.tensor(self.coeffs, dtype=torch.float32, device=dev))
            self._offsets = (offs, off, torch.tensor(offs, dtype=torch.int64, device=dev))
            self._key = key
        return self._dev

    def offsets(self):
        return self._offsets


class L2Reg(torch.autograd.Function):
    """sum_t coeff_t * sum(w_t^2) -> tensor of shape [1]  (deepctr/models/basemodel.py:412-428, l1 == 0).

    `defer` = (EmbedPlan, L2Plan of the first n_defer tensors, L2Plan of the others) or None.  When
    given, the first n_defer tensors are exactly the tables of that gather, in its order: their L2
    gradient is not returned here but handed to EmbedGather.backward, which uses it as the initial
    content of the dense table gradients it has to build anyway.  Only the model's own train step
    passes `defer` (it guarantees that the gather's backward runs in the same pass)."""

    @staticmethod
    def forward(ctx, plan: L2Plan, defer, n_defer, *tensors):
        lib = _lib.load()
        for t in tensors:
            _need_cuda(t, "regularised tensor")
            if not t.is_contiguous():
                raise ValueError("xdfm: regularised tensors must be contiguous")
        ptrs, numel, coeff = plan.tables(tensors)
        T = len(tensors)
        dev = tensors[0].device
        partials = torch.empty(32 * T, dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(lib.xdfm_l2_reg_fwd(_ptr(ptrs), _ptr(numel), _ptr(coeff), T, _ptr(partials), _ptr(out), _stream()),
                   "l2_reg_fwd")
        ctx.plan, ctx.defer, ctx.n_defer = plan, defer, n_defer
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        tensors = ctx.saved_tensors
        gs = g.reshape(1).contiguous()
        skip, plan = 0, ctx.plan
        if ctx.defer is not None:
            embed_plan, sub_plan, plan = ctx.defer
            embed_plan.reg_defer = (gs, sub_plan)
            skip = ctx.n_defer
        rest = tensors[skip:]
        grads = [None] * skip
        if rest:
            ptrs, numel, coeff = plan.tables(rest)
            offs, total, off_dev = plan.offsets()
            flat = torch.empty(total, dtype=torch.float32, device=rest[0].device)
            _lib.check(lib.xdfm_l2_reg_bwd(_ptr(ptrs), _ptr(numel), _ptr(coeff), len(rest), _ptr(gs), _ptr(flat),
                                           _ptr(off_dev), 0, _stream()), "l2_reg_bwd")
            grads += [flat[o:o + t.numel()].view(t.shape) for o, t in zip(offs, rest)]
        return (None, None, None) + tuple(grads)


def l2_regulariser(tensors, coeffs, cache: dict, embed_plan=None, n_defer=0):
    """Evaluate the L2 term with plans cached in `cache` (a dict owned by the model)."""
    n_defer = n_defer if embed_plan is not None else 0
    key = (tuple(float(c) for c in coeffs), n_defer)
    plans = cache.get(key)
    if plans is None:
        plans = cache[key] = (L2Plan(coeffs), L2Plan(coeffs[:n_defer]) if n_defer else None,
                              L2Plan(coeffs[n_defer:]) if n_defer and len(coeffs) > n_defer else None)
    full, sub, rest = plans
    defer = (embed_plan, sub, rest) if n_defer else None
    return L2Reg.apply(full, defer, n_defer, *tensors)
