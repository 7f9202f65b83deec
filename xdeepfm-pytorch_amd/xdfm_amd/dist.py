"""Row-parallel data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" for the CPU tests).

The reference's only multi-GPU mechanism is single-process nn.DataParallel with
`batch_size *= len(gpus)` (deepctr/models/basemodel.py:206-209): per-GPU batch = the batch_size
argument, loss = SUM over the global batch (basemodel.py:254), L2 added once (basemodel.py:255-257).
Here every rank holds a full replica (Criteo-scale tables fit 288 GB), takes a contiguous row
slice of each global batch, and one exchange step per iteration makes the replicas' gradients
identical:

  * dense weights (CIN, DNN, heads, attention: a few MB): ONE flat all-reduce SUM;
  * embedding tables: never all-reduced as dense [V, D] tensors.  The per-rank row gradients
    (B_local x (m x D + X row + 1) floats, 7.7 MB at B=4096, m=26, D=16) are all-gathered in ONE
    collective and every rank runs the same single scatter launch over all ranks' rows, so the
    dense table gradients come out identical everywhere without moving table-sized data over the
    point-to-point links;
  * the L2 gradient is applied locally after the reduce (it is the same on every replica).
"""
import torch
import torch.distributed as dist

_ctx = None


def current():
    """The process-wide context, or None when torch.distributed is not initialised / world size 1."""
    global _ctx
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return None
    if _ctx is None or _ctx.world != dist.get_world_size():
        _ctx = RowParallel()
    return _ctx


def split_points(n, world):
    """Balanced contiguous split of n rows: rank r owns [p[r], p[r+1])."""
    return [(r * n) // world for r in range(world + 1)]


class RowParallel(object):
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self._n_global = None
        self._replicated = set()      # ids of parameters whose gradients are already identical on all ranks

    # ---------------------------------------------------------------- sharding
    def shard(self, t):
        """This rank's rows of a global-batch tensor (every rank sees the same global batch).  A global batch with
        fewer rows than ranks (the tail of an epoch) leaves some ranks without rows: such a rank takes row 0 as a
        stand-in, so that it still runs the step and joins every collective, and differentiates it with weight 0
        (`row_weight`): its data gradients are exact zeros and its rows are not part of the gathered metrics."""
        n = t.shape[0]
        if n < 1:
            raise ValueError("empty global batch")
        p = split_points(n, self.world)
        self._n_global = n
        lo, hi = p[self.rank], p[self.rank + 1]
        return t[lo:hi] if hi > lo else t[0:1]

    def local_sizes(self):
        """Rows per rank that belong to the global batch (0 for a rank that only holds a stand-in row)."""
        p = split_points(self._n_global, self.world)
        return [p[r + 1] - p[r] for r in range(self.world)]

    def step_sizes(self):
        """Rows per rank that take part in the step (a stand-in row counts)."""
        return [max(s, 1) for s in self.local_sizes()]

    def row_weight(self):
        return 1.0 if self.local_sizes()[self.rank] > 0 else 0.0

    # ---------------------------------------------------------------- collectives
    # ONE call site per collective, whatever the backend: "nccl" (RCCL) takes device tensors as they are; "gloo" has no
    # device transport on this stack, so a device tensor is staged through the host around the very same call
    # (`_stage` / `_unstage`).  The world-size-2 / -4 gloo tests therefore run the production call sequence -- the same
    # all_reduce, all_gather_into_tensor and flag exchange, on the same shapes, in the same order -- and the only code
    # the RCCL run takes alone is the identity branch of the two staging helpers.
    def _via_host(self, t):
        return self.backend == "gloo" and t.is_cuda

    def _stage(self, t):
        return t.cpu() if self._via_host(t) else t

    @staticmethod
    def _unstage(staged, like):
        return staged if staged.device == like.device else staged.to(like.device)

    def flag_device(self):
        """Where small control tensors of a collective live: the current GPU under RCCL, the host under gloo."""
        return torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")

    def all_reduce_sum(self, t):
        h = self._stage(t)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        if h is not t:
            t.copy_(h)
        return t

    def all_gather_padded(self, t, sizes):
        """Gather tensors whose dim-0 sizes differ by rank; returns the list in rank order."""
        mx = max(sizes)
        if t.shape[0] < mx:
            pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], dim=0)
        G = self.all_gather_rows(t)
        return [G[r * mx:r * mx + s] for r, s in enumerate(sizes)]

    def sum_scalar(self, t):
        v = t.detach().reshape(1).clone()
        return float(self.all_reduce_sum(v).item())

    def any_flag(self, flag):
        v = torch.tensor([1.0 if flag else 0.0], device=self.flag_device())
        return bool(self.all_reduce_sum(v).item() > 0)

    def gather_rows(self, v):
        sizes = self.local_sizes()
        if sizes[self.rank] == 0:
            v = v[:0]                          # a stand-in row is not part of the global batch
        return torch.cat(self.all_gather_padded(v, sizes), dim=0)

    # ---------------------------------------------------------------- gradient exchange
    def mark_replicated(self, params):
        """Parameters whose gradient is built from the exchanged rows (identical on every rank)."""
        for p in params:
            self._replicated.add(id(p))

    def reduce_dense_grads(self, model):
        """One flat all-reduce SUM over every gradient that is not already replicated."""
        grads = [p.grad for p in model.parameters() if p.grad is not None and id(p) not in self._replicated]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.all_reduce_sum(flat)
        views, off = [], 0
        for g in grads:
            n = g.numel()
            views.append(flat[off:off + n].view_as(g))
            off += n
        torch._foreach_copy_(grads, views)            # one multi-tensor launch instead of one copy per gradient

    def all_gather_rows(self, t):
        """[rows, ...] on every rank (same shape everywhere) -> [world * rows, ...], rank-major: one
        all_gather_into_tensor (a single contiguous receive buffer; no per-rank output list, no concatenation)."""
        src = self._stage(t.contiguous())
        out = torch.empty((self.world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src, group=self.group)
        return self._unstage(out, t)

    def exchange_rows(self, X, d_emb, d_dnn, d_lin):
        """All-gather the inputs of the embedding scatter with ONE collective.

        Each rank packs a [rows, m*D + ncols + 1] buffer -- per example: the row gradients of its m embedding rows
        (the CIN's and the DNN's contributions already summed: that halves the volume), its row of X (ids and dense
        values) and its linear-logit gradient -- 1.9 KB per example at config 2.  Ranks with fewer rows (ragged last
        batch) pad with zero rows, which scatter exact zeros.  The gathered buffer is handed to the scatter as
        strided views, so every rank runs ONE scatter launch over all ranks' rows in rank order; no transposes, no
        per-rank launches.  Returns [(X, None, row_grads, d_lin)] for EmbedGather.scatter."""
        sizes = self.step_sizes()
        B = X.shape[0]
        if B != sizes[self.rank]:
            raise RuntimeError("exchange_rows: local batch %d does not match the sharded size %d" % (B, sizes[self.rank]))
        if d_emb is None and d_dnn is None:
            raise RuntimeError("exchange_rows: no row gradients to exchange")
        ncols = X.shape[1]
        if d_emb is not None:
            m = d_emb.shape[0]
            D = d_emb.shape[1] // max(B, 1)
            mD = m * D
        else:
            mD = None
        rows = max(sizes)
        if mD is None:
            # without the FM-layout gradient the width of the sparse part is not known here: ship d_dnn whole
            mD = d_dnn.shape[1]
        Q = torch.empty((rows, mD + ncols + 1), dtype=torch.float32, device=X.device)
        if rows > B:
            Q[B:].zero_()
        if d_emb is not None:
            e = d_emb.view(m, B, D).permute(1, 0, 2)
            if d_dnn is not None:
                torch.add(d_dnn[:, :mD].view(B, m, D), e, out=Q[:B, :mD].view(B, m, D))
            else:
                Q[:B, :mD].view(B, m, D).copy_(e)
        else:
            Q[:B, :mD].copy_(d_dnn[:, :mD])
        Q[:B, mD:mD + ncols].copy_(X)
        if d_lin is not None:
            Q[:B, mD + ncols].copy_(d_lin.reshape(B))
        else:
            Q[:B, mD + ncols].zero_()
        G = self.all_gather_rows(Q)
        return [(G[:, mD:mD + ncols], None, G[:, :mD], G[:, mD + ncols])]
