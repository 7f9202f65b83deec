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
    (B_local x m x (D+1) floats, about 7.7 MB at B=4096, m=26, D=16) are all-gathered and every
    rank runs the same scatter over all ranks' rows in rank order, so the dense table gradients
    come out identical everywhere without moving table-sized data over the point-to-point links;
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
        """This rank's rows of a global-batch tensor (every rank sees the same global batch)."""
        n = t.shape[0]
        if n < self.world:
            raise ValueError("global batch of %d rows cannot be split over %d ranks" % (n, self.world))
        p = split_points(n, self.world)
        self._n_global = n
        return t[p[self.rank]:p[self.rank + 1]]

    def local_sizes(self):
        p = split_points(self._n_global, self.world)
        return [p[r + 1] - p[r] for r in range(self.world)]

    # ---------------------------------------------------------------- collectives
    def _via_host(self, t):
        return self.backend == "gloo" and t.is_cuda

    def all_reduce_sum(self, t):
        if self._via_host(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather_padded(self, t, sizes):
        """Gather tensors whose dim-0 sizes differ by rank; returns the list in rank order."""
        mx = max(sizes)
        if t.shape[0] < mx:
            pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], dim=0)
        src = t.contiguous()
        dev = src.device
        if self._via_host(src):
            src = src.cpu()
        outs = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(outs, src, group=self.group)
        return [o[:s].to(dev) for o, s in zip(outs, sizes)]

    def sum_scalar(self, t):
        v = t.detach().reshape(1).clone()
        return float(self.all_reduce_sum(v).item())

    def any_flag(self, flag):
        dev = "cuda" if self.backend == "nccl" else "cpu"
        v = torch.tensor([1.0 if flag else 0.0], device=dev)
        return bool(self.all_reduce_sum(v).item() > 0)

    def gather_rows(self, v):
        return torch.cat(self.all_gather_padded(v, self.local_sizes()), dim=0)

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
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    def exchange_rows(self, X, d_emb, d_dnn, d_lin):
        """All-gather the inputs of the embedding scatter; returns one (X, d_emb, d_dnn, d_lin) per
        rank, in rank order, so every rank accumulates the same rows in the same order."""
        sizes = self.local_sizes()
        if X.shape[0] != sizes[self.rank]:
            raise RuntimeError("exchange_rows: local batch %d does not match the sharded size %d"
                               % (X.shape[0], sizes[self.rank]))
        m = d_emb.shape[0]
        B = X.shape[0]
        D = d_emb.shape[1] // max(B, 1)
        Xs = self.all_gather_padded(X, sizes)
        # FM layout [m, B*D] -> rows-major [B, m*D] for the variable-size gather, and back
        e = d_emb.view(m, B, D).permute(1, 0, 2).reshape(B, m * D)
        es = self.all_gather_padded(e, sizes)
        ds = self.all_gather_padded(d_dnn, sizes)
        ls = self.all_gather_padded(d_lin, sizes)
        out = []
        for r in range(self.world):
            er = es[r].view(sizes[r], m, D).permute(1, 0, 2).reshape(m, sizes[r] * D).contiguous()
            out.append((Xs[r], er, ds[r], ls[r]))
        return out
