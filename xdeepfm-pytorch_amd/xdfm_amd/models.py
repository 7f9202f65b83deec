"""xDeepFM model family on the MI355X kernels, behind the reference's Keras-like API.

Public surface (names, positional order, defaults, state_dict keys) follows
deepctr/models/basemodel.py:95-527, deepctr/models/xdeepfm.py:17-107 and
deepctr/models/xdeepfm_attn.py:25-301 so that the xdftrain*.py flows run unchanged:
`Model(linear_cols, dnn_cols, ...)`, `compile`, `fit`, `evaluate`, `predict`, `state_dict`.

What differs underneath: `forward` is one fused gather launch (embeddings in FM layout, DNN input
and linear logit together), the CIN stack on MFMA kernels, and a handful of tiny torch GEMMs; with
torch.distributed initialised (one process per GPU, RCCL) `fit` shards every global batch row-wise
over the ranks (xdfm_amd/dist.py).
"""
import contextlib
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dist as xdist
from . import metrics as M
from . import ops
from .callbacks import CallbackList, History
from .inputs import (DenseFeat, SparseFeat, VarLenSparseFeat, build_input_features, create_embedding_matrix,
                     split_columns)
from .layers import CIN, DNN, CINAttention, CINAttentionV2, PredictionLayer

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    tqdm = None


def _slice(arrays, start=None, stop=None):
    """x[start:stop] for one array or each array of a list (deepctr/layers/utils.py:19-70 as fit uses it)."""
    if isinstance(arrays, (list, tuple)):
        return [None if a is None else a[start:stop] for a in arrays]
    return arrays[start:stop]


def epoch_order(n, shuffle):
    """Row order of one epoch, drawn exactly as `DataLoader(TensorDataset, shuffle=shuffle)` draws it
    (basemodel.py:213-214, :241): creating the iterator takes one int64 from the default generator (its
    base seed), then RandomSampler takes one more as the seed of a fresh generator for randperm.  The
    same primitives in the same order give the same batches as the reference for a given torch seed --
    without DataLoader's per-sample indexing and 4096-way collate (tens of ms per batch)."""
    torch.empty((), dtype=torch.int64).random_()
    if not shuffle:
        return None
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    gen = torch.Generator()
    gen.manual_seed(seed)
    return torch.randperm(n, generator=gen)


RESIDENT_LIMIT_BYTES = 64 << 30     # keep the packed fp32 input matrix on the device up to this size


def _reject_varlen(cols):
    if any(isinstance(c, VarLenSparseFeat) for c in cols):
        raise NotImplementedError("VarLenSparseFeat is outside the xDeepFM hot path (no reference script uses it)")


class Linear(nn.Module):
    """Parameters of the linear part: one [vocab,1] table per sparse field and a [n_dense,1] weight
    (deepctr/models/basemodel.py:34-92).  Inside the models its logit comes out of the fused gather;
    calling the module directly runs the same kernel for the linear part alone."""

    def __init__(self, feature_columns, feature_index, init_std=0.0001, device='cpu'):
        super().__init__()
        self.feature_index = feature_index
        self.device = device
        self.sparse_feature_columns, self.dense_feature_columns, self.varlen_sparse_feature_columns = \
            split_columns(feature_columns)
        _reject_varlen(self.varlen_sparse_feature_columns)
        # All draws happen on the CPU generator and the module moves afterwards, so a seed gives the
        # same initial weights on every device (= the reference's CPU path; on a CUDA device the
        # reference itself would take these two draws from the device generator, basemodel.py:47-61).
        self.embedding_dict = create_embedding_matrix(feature_columns, init_std, linear=True, sparse=False,
                                                      device='cpu')
        for emb in self.embedding_dict.values():          # second draw, basemodel.py:55-56
            nn.init.normal_(emb.weight, mean=0, std=init_std)
        n_dense = sum(fc.dimension for fc in self.dense_feature_columns)
        if n_dense > 0:
            self.weight = nn.Parameter(torch.Tensor(n_dense, 1))
            nn.init.normal_(self.weight, mean=0, std=init_std)
        self._plan = None
        self.to(device)

    def tables(self):
        return [self.embedding_dict[fc.embedding_name].weight for fc in self.sparse_feature_columns]

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_plan"] = None
        return state

    def forward(self, X, sparse_feat_refine_weight=None):
        if sparse_feat_refine_weight is not None:
            raise NotImplementedError("refine weights belong to IFM/DIFM, not to the xDeepFM path")
        if self._plan is None:
            dense_cols = [c for fc in self.dense_feature_columns for c in range(*self.feature_index[fc.name])]
            self._plan = ops.EmbedPlan([self.feature_index[fc.name][0] for fc in self.sparse_feature_columns],
                                       [fc.vocabulary_size for fc in self.sparse_feature_columns], dense_cols, 1)
        if not self.sparse_feature_columns:
            out = torch.zeros([X.shape[0], 1], device=X.device)
            if self.dense_feature_columns:
                cols = self._plan.dense_cols
                out = out + X[:, cols].matmul(self.weight)
            return out
        tabs = self.tables()
        w = self.weight if self.dense_feature_columns else None
        _, _, lin = ops.EmbedGather.apply(X, w, self._plan, True, *tabs, *tabs)
        return lin


class BaseModel(nn.Module):
    def __init__(self, linear_feature_columns, dnn_feature_columns, l2_reg_linear=1e-5, l2_reg_embedding=1e-5,
                 init_std=0.0001, seed=1024, task='binary', device='cpu', gpus=None):
        super().__init__()
        torch.manual_seed(seed)                              # basemodel.py:100
        if gpus:
            raise ValueError("`gpus=` (single-process nn.DataParallel, basemodel.py:206-209) is replaced by one "
                             "process per GPU: launch with torch.distributed.run and leave gpus=None")
        self.dnn_feature_columns = dnn_feature_columns
        self.linear_feature_columns = linear_feature_columns
        self.device = device
        self.gpus = gpus
        self.reg_loss = torch.zeros((1,), device=device)
        self.aux_loss = torch.zeros((1,), device=device)
        self._aux_unset = True              # aux_loss is still the initial zero: the train step need not add it
        _reject_varlen(list(linear_feature_columns) + list(dnn_feature_columns))
        self.feature_index = build_input_features(list(linear_feature_columns) + list(dnn_feature_columns))
        self.embedding_dict = create_embedding_matrix(dnn_feature_columns, init_std, sparse=False, device=device)
        self.linear_model = Linear(linear_feature_columns, self.feature_index, device=device)
        self.regularization_weight = []
        self.add_regularization_weight(self.embedding_dict.parameters(), l2=l2_reg_embedding)
        self.add_regularization_weight(self.linear_model.parameters(), l2=l2_reg_linear)
        self.out = PredictionLayer(task, )
        self.to(device)
        self._is_graph_network = True       # attributes Keras callbacks look at (basemodel.py:133-135)
        self._ckpt_saved_epoch = False
        self.history = History()
        self.stop_training = False
        self._plan = None

    # ------------------------------------------------------------------ pickling (ModelCheckpoint with
    # save_weights_only=False calls torch.save(model), deepctr/callbacks.py:41-73)
    _UNPICKLED = ("_graphed_step", "_plan", "_l2_cache", "_unit_grad_cache", "_sparse_cols", "_fused_linear", "_own_step",
                  "_row_weight")

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._UNPICKLED:           # captured graphs, ctypes descriptors, device-side plans: rebuilt on first use
            state.pop(k, None)
        state["_plan"] = None
        return state

    # The optimizer may hold table rows that are several steps behind (optim.TableAdam, deferred update): whoever reads
    # or replaces the parameters as a whole -- evaluation, a checkpoint, a reload -- gets them brought up to date first.
    def _flush_optim(self):
        opt = self.__dict__.get("optim")
        if opt is not None and hasattr(opt, "flush"):
            opt.flush()

    def train(self, mode=True):
        if not mode:
            self._flush_optim()
        return super().train(mode)

    def state_dict(self, *args, **kwargs):
        self._flush_optim()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._flush_optim()
        return super().load_state_dict(*args, **kwargs)

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault("_plan", None)
        lm = self.__dict__.get("_modules", {}).get("linear_model")
        if lm is not None:
            lm._plan = None

    def _plans(self):
        plans = [self._plan, getattr(self.linear_model, "_plan", None)]
        return [p for p in plans if p is not None]

    def _raise_on_bad_ids(self):
        """The reference's nn.Embedding raises IndexError (CPU) or device-asserts on an id outside [0, vocabulary_size)
        (basemodel.py:368-370).  K1 clamps such an id and raises a device flag instead of stopping the stream; the flag
        is read where the host synchronises anyway -- the per-epoch loss read-back of `fit`, the final copy of
        `predict` / `evaluate` -- so the error is deferred to the end of the epoch / call, never lost."""
        dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
        if dev.type != "cuda":
            return
        dp = xdist.current()
        for plan in self._plans():
            bad = plan.check_ids(dev)
            if dp is not None:
                bad = dp.any_flag(bad)       # every rank raises together: a lone IndexError would leave the others in a collective
            if bad:
                bounds = ", ".join("%d" % v for v in plan.vocab[:8]) + (" ..." if len(plan.vocab) > 8 else "")
                raise IndexError("index out of range in self: a sparse feature id lies outside [0, vocabulary_size) "
                                 "(vocabulary sizes: %s) -- check SparseFeat(vocabulary_size=max_id + 1)" % bounds)

    def _drop_table_grads(self):
        """After the model's own step the tables' `.grad` are views of the kept gradient buffer (all zeros again once K7
        has consumed them).  They are released so that code which drives autograd itself afterwards gets fresh dense
        gradients instead of accumulating into that buffer behind the optimizer's back."""
        tables = self._gather_tables()
        if tables is None or self._plan is None or not self._plan.arenas():
            return
        for t in tables:
            t.grad = None
        w = getattr(self.linear_model, "weight", None)
        if w is not None:
            w.grad = None

    # ------------------------------------------------------------------ fused input stage
    def _gather_plan(self):
        if self._plan is None:
            sparse, dense, _ = split_columns(self.dnn_feature_columns)
            dims = {fc.embedding_dim for fc in sparse}
            if len(dims) > 1:
                raise ValueError("embedding_dim of SparseFeat and VarlenSparseFeat must be same in this model!")
            lin_sparse = self.linear_model.sparse_feature_columns
            self._fused_linear = [fc.name for fc in lin_sparse] == [fc.name for fc in sparse] and \
                [fc.name for fc in self.linear_model.dense_feature_columns] == [fc.name for fc in dense]
            dense_cols = [c for fc in dense for c in range(*self.feature_index[fc.name])]
            self._plan = ops.EmbedPlan([self.feature_index[fc.name][0] for fc in sparse],
                                       [fc.vocabulary_size for fc in sparse], dense_cols,
                                       dims.pop() if dims else 1)
            self._sparse_cols = sparse
            dp = xdist.current()
            if dp is not None and self._fused_linear:
                # these gradients are built from the all-gathered rows and are identical on every rank
                self._plan.dp = dp
                dp.mark_replicated([self.embedding_dict[fc.embedding_name].weight for fc in sparse])
                dp.mark_replicated(self.linear_model.parameters())
        return self._plan

    def fused_inputs(self, X):
        """(emb_fm [m, B*D], dnn_in [B, m*D+nd], linear_logit [B,1]) from ONE gather launch
        (input_from_feature_columns + linear_model + both concatenations of the reference forward)."""
        plan = self._gather_plan()
        self._use_grad_arena(plan)
        if not self._sparse_cols:
            raise NotImplementedError("the xDeepFM path needs at least one SparseFeat")
        tabs = [self.embedding_dict[fc.embedding_name].weight for fc in self._sparse_cols]
        if self._fused_linear:
            w = getattr(self.linear_model, "weight", None)
            return ops.EmbedGather.apply(X, w, plan, True, *tabs, *self.linear_model.tables())
        emb_fm, dnn_in, _ = ops.EmbedGather.apply(X, None, plan, False, *tabs)
        return emb_fm, dnn_in, self.linear_model(X)

    def input_from_feature_columns(self, X, feature_columns, embedding_dict, support_dense=True):
        """API of basemodel.py:354-380: ([B,1,D] per sparse field, [B,k] per dense field)."""
        sparse, dense, varlen = split_columns(feature_columns)
        _reject_varlen(varlen)
        if not support_dense and len(dense) > 0:
            raise ValueError("DenseFeat is not supported in dnn_feature_columns")
        emb_list = []
        if sparse:
            plan = ops.EmbedPlan([self.feature_index[fc.name][0] for fc in sparse],
                                 [fc.vocabulary_size for fc in sparse], [], sparse[0].embedding_dim)
            tabs = [embedding_dict[fc.embedding_name].weight for fc in sparse]
            emb_fm, _, _ = ops.EmbedGather.apply(X, None, plan, False, *tabs)
            B = X.shape[0]
            emb_list = [emb_fm[j].view(B, 1, plan.D) for j in range(plan.m)]
        dense_list = [X[:, self.feature_index[fc.name][0]:self.feature_index[fc.name][1]] for fc in dense]
        return emb_list, dense_list

    def compute_input_dim(self, feature_columns, include_sparse=True, include_dense=True, feature_group=False):
        sparse, dense, varlen = split_columns(feature_columns)
        sparse = sparse + varlen
        dim = 0
        if include_sparse:
            dim += len(sparse) if feature_group else sum(fc.embedding_dim for fc in sparse)
        if include_dense:
            dim += sum(fc.dimension for fc in dense)
        return dim

    @property
    def embedding_size(self):
        sparse, _, varlen = split_columns(self.dnn_feature_columns)
        sizes = {fc.embedding_dim for fc in sparse + varlen}
        if len(sizes) > 1:
            raise ValueError("embedding_dim of SparseFeat and VarlenSparseFeat must be same in this model!")
        return list(sizes)[0]

    # ------------------------------------------------------------------ regularisation
    def add_regularization_weight(self, weight_list, l1=0.0, l2=0.0):
        weight_list = [weight_list] if isinstance(weight_list, nn.parameter.Parameter) else list(weight_list)
        self.regularization_weight.append((weight_list, l1, l2))

    def _l2_terms(self):
        """[(tensor, l2 strength)] of every regularised tensor with l2 > 0, and the summed l1 term."""
        l2_terms, l1_total = [], None
        for weight_list, l1, l2 in self.regularization_weight:
            for w in weight_list:
                p = w[1] if isinstance(w, tuple) else w
                if l1 > 0:
                    term = torch.sum(l1 * torch.abs(p))
                    l1_total = term if l1_total is None else l1_total + term
                if l2 > 0:
                    l2_terms.append((p, float(l2)))
        return l2_terms, l1_total

    def _l2_fusion(self):
        """(tensors, strengths) of the whole L2 term when the optimizer applies it itself while it streams the
        weights (xdfm_amd.optim.TableAdam, K7), else None.  Only the model's own train step uses this."""
        from .optim import TableAdam
        opt = getattr(self, "optim", None)
        if not isinstance(opt, TableAdam):
            return None
        tensors, coeffs = [], []
        for weight_list, l1, l2 in self.regularization_weight:
            for w in weight_list:
                p = w[1] if isinstance(w, tuple) else w
                if l1 > 0 or not p.is_cuda:
                    return None
                if l2 > 0:
                    tensors.append(p)
                    coeffs.append(float(l2))
        if not opt.owns(tensors):
            return None
        return tensors, coeffs

    def _gather_tables(self):
        """The tensors the fused gather reads, in its order (None before the first forward)."""
        if self._plan is None or not getattr(self, "_fused_linear", False):
            return None
        return [self.embedding_dict[fc.embedding_name].weight for fc in self._sparse_cols] + \
            self.linear_model.tables()

    def get_regularization_loss(self, _defer_tables=False, _part="all"):
        """sum over groups of l1*|w| + l2*w^2 (basemodel.py:412-428), shape [1].  All l2 groups are
        evaluated by one multi-tensor launch (K6) instead of four ATen calls per tensor.

        The private arguments are used by the model's own train step only.  `_defer_tables`: the L2
        gradient of the embedding tables is produced inside the gather's backward (as the initial value
        of the dense table gradients) instead of as table-sized tensors that autograd has to add.
        `_part`: "all", or "tables" / "rest" to split the term (row-parallel training adds the rest
        after the gradient all-reduce)."""
        l2_terms, total = self._l2_terms()
        tables = self._gather_tables()
        table_ids = {id(t) for t in tables} if tables is not None else set()
        if _part == "tables":
            l2_terms, total = [x for x in l2_terms if id(x[0]) in table_ids], None
        elif _part == "rest":
            l2_terms = [x for x in l2_terms if id(x[0]) not in table_ids]
        if l2_terms:
            cache = self.__dict__.setdefault("_l2_cache", {})
            embed_plan, n_defer = None, 0
            if _defer_tables and tables is not None and _part != "rest":
                coeff_of = {id(p): c for p, c in l2_terms}
                if all(id(t) in coeff_of for t in tables) and all(t.requires_grad for t in tables):
                    l2_terms = [(t, coeff_of[id(t)]) for t in tables] + [x for x in l2_terms if id(x[0]) not in table_ids]
                    embed_plan, n_defer = self._plan, len(tables)
            term = ops.l2_regulariser([p for p, _ in l2_terms], [c for _, c in l2_terms], cache, embed_plan, n_defer)
            total = term if total is None else total + term
        if total is None:
            total = torch.zeros((1,), device=self.device)
        return total.reshape(1)

    def train_on_batch(self, x, y):
        """One optimisation step = the body of the reference's batch loop (basemodel.py:245-262):
        forward, BCE(sum) + L2 (+ aux), backward, optimizer step.  Returns (y_pred, data_loss,
        total_loss) as device tensors -- no host synchronisation.  On a GPU, from the third step of a
        batch shape on, the step is replayed from a captured HIP graph (xdfm_amd/graphstep.py); the
        returned tensors are then the graph's static outputs, valid until the next call."""
        step = self.__dict__.get("_graphed_step")
        if step is None:
            from .graphstep import GraphedStep
            step = self.__dict__["_graphed_step"] = GraphedStep(self)
        return step(x, y)

    def _fused_head(self, x, y):
        """(y_pred, loss) through one launch (ops.Head) when the model is the binary-task xDeepFM family with the
        stock F.binary_cross_entropy loss; None otherwise."""
        if self.loss_func is not F.binary_cross_entropy or not hasattr(self, "head_inputs") or not x.is_cuda:
            return None
        out = self.out
        if getattr(out, "task", None) != "binary" or y.numel() != x.shape[0]:
            return None
        lin, cin_out, dnn_out = self.head_inputs(x)
        return ops.Head.apply(y, out.bias if out.use_bias else None, lin,
                              cin_out, self.cin_linear.weight if cin_out is not None else None,
                              dnn_out, self.dnn_linear.weight if dnn_out is not None else None)

    def _loss_forward(self, x, y):
        """(y_pred, data loss): forward + loss of the reference's batch loop (basemodel.py:250-254)."""
        head = self._fused_head(x, y)
        if head is not None:
            y_pred, loss = head
            return y_pred, loss.reshape(())
        y_pred = self(x).squeeze()
        loss_func = self.loss_func
        if isinstance(loss_func, list):
            assert len(loss_func) == self.num_tasks, "the length of `loss_func` should be equal with `self.num_tasks`"
            loss = sum(loss_func[i](y_pred[:, i], y[:, i], reduction='sum') for i in range(self.num_tasks))
        else:
            loss = loss_func(y_pred, y.squeeze(), reduction='sum')
        return y_pred, loss

    # Row-parallel step with the L2 term in K7, in two halves.  The first (forward, loss, backward down to the
    # row gradients of the gather) contains no collective, so it can be replayed from a HIP graph; the second
    # exchanges the rows, scatters, all-reduces the dense gradients and runs the optimizer, eagerly.
    def _use_grad_arena(self, plan):
        """Keep the dense table gradients across steps when the optimizer can consume them by their marks -- inside the
        model's OWN train step only (`_own_step`): there nothing touches a gradient between the scatter and K7.  A
        user-driven loop (model(x); loss.backward(); edit .grad; optim.step()) gets ordinary dense gradients, because
        K7 would not see what such code adds outside the marked chunks."""
        from .optim import TableAdam
        plan.arena_on = bool(self.__dict__.get("_own_step")) and isinstance(getattr(self, "optim", None), TableAdam)
        if plan.arena_on and plan not in self.optim.grad_sources:
            self.optim.grad_sources.append(plan)

    @contextlib.contextmanager
    def _own_step_scope(self):
        prev = self.__dict__.get("_own_step", False)
        self.__dict__["_own_step"] = True
        if self._plan is not None:
            self._use_grad_arena(self._plan)
        try:
            yield
        finally:
            self.__dict__["_own_step"] = prev
            if self._plan is not None and not prev:
                self._plan.arena_on = False

    def _with_step_extra(self, total):
        """A model's `_loss_forward` may leave a further differentiable term of the step's objective in `_step_extra`
        (xDeepFMPro: sfg_weight * sfg_loss, basemodel_sfg.py:343) and a value for the epoch log in `_step_log`."""
        extra = self.__dict__.get("_step_extra")
        return total if extra is None else total + extra.reshape(total.shape)

    def _unit_grad(self, loss):
        """Root gradient of a backward pass, cached: autograd would otherwise fill a fresh ones tensor every step.
        A rank that holds only a stand-in row (a global batch with fewer rows than ranks, dist.RowParallel.shard)
        differentiates with a zero root: all its data gradients are exact zeros."""
        if self.__dict__.get("_row_weight", 1.0) == 0.0:
            return torch.zeros_like(loss)
        key = (tuple(loss.shape), loss.device, loss.dtype)
        hit = self.__dict__.get("_unit_grad_cache")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_unit_grad_cache"] = (key, torch.ones_like(loss))
        return hit[1]

    def _split_step_first(self, x, y):
        with self._own_step_scope():
            return self._split_step_first_body(x, y)

    def _split_step_first_body(self, x, y):
        self.optim.zero_grad()
        plan = self._gather_plan()
        plan.stash = []
        try:
            y_pred, loss = self._loss_forward(x, y)
            root = loss if self._aux_unset else loss + self.aux_loss
            root = self._with_step_extra(root)
            root.backward(self._unit_grad(root))
            stash = plan.stash
        finally:
            plan.stash = None
        return y_pred.detach(), loss.detach(), stash

    def _split_step_second(self, y_pred, loss, stash, fuse):
        with self._own_step_scope():
            out = self._split_step_second_body(y_pred, loss, stash, fuse)
        self._drop_table_grads()
        return out

    def _split_step_second_body(self, y_pred, loss, stash, fuse):
        dp = xdist.current()
        dense_w = self.linear_model.weight if getattr(self.linear_model, "dense_feature_columns", None) else None
        ops.apply_stashed_scatter(stash, dense_w, self._gather_tables())
        if dp is not None:
            dp.reduce_dense_grads(self)
        self.optim.arm_l2(*fuse)
        self.optim.step()
        total_loss = loss if self._aux_unset else loss + self.aux_loss.detach()
        if self.__dict__.get("_step_extra") is not None:
            total_loss = total_loss + self._step_extra.detach().reshape(total_loss.shape)
        if self.optim.l2_value is not None:
            total_loss = total_loss + self.optim.l2_value
        return y_pred, loss, total_loss.detach()

    def _can_split_step(self, dp, fuse):
        if dp is None or fuse is None:
            return False
        self._gather_plan()
        return bool(getattr(self, "_fused_linear", False))

    def _train_step_eager(self, x, y):
        with self._own_step_scope():
            out = self._train_step_eager_body(x, y)
        self._drop_table_grads()
        return out

    def _train_step_eager_body(self, x, y):
        dp = xdist.current()
        fuse = self._l2_fusion()
        if self._can_split_step(dp, fuse):
            y_pred, loss, stash = self._split_step_first(x, y)
            return self._split_step_second(y_pred, loss, stash, fuse)
        self.optim.zero_grad()
        y_pred, loss = self._loss_forward(x, y)
        if fuse is not None:
            # gradient and value of the L2 term come from K7 (added after the gradient all-reduce when row-parallel:
            # the term is identical on every replica and must count once)
            self.optim.arm_l2(*fuse)
            total_loss = loss if self._aux_unset else loss + self.aux_loss
            total_loss = self._with_step_extra(total_loss)
            total_loss.backward(self._unit_grad(total_loss))
            if dp is not None:
                dp.reduce_dense_grads(self)
        elif dp is None:
            reg_loss = self.get_regularization_loss(_defer_tables=True)
            total_loss = self._with_step_extra(loss + reg_loss + self.aux_loss)
            total_loss.backward()
        else:
            # Data-loss gradients are SUMMED over ranks (the loss is a sum over the global batch); the L2
            # term is identical on every replica and must be applied once.  Tables: their gradient is
            # built from the all-gathered rows (identical on all ranks) with the L2 term folded in.
            # Dense weights: all-reduce the data gradients, then add the L2 gradient locally.
            reg_t = self.get_regularization_loss(_defer_tables=True, _part="tables")
            reg_d = self.get_regularization_loss(_part="rest")
            (loss * self.__dict__.get("_row_weight", 1.0) + reg_t).backward()
            dp.reduce_dense_grads(self)
            (reg_d + self.aux_loss).backward()
            total_loss = loss.detach() + reg_t.detach() + reg_d.detach() + self.aux_loss
        self.optim.step()
        if fuse is not None and self.optim.l2_value is not None:
            total_loss = total_loss.detach() + self.optim.l2_value
        # detached: a caller that keeps these alive must not keep the step's autograd graph (and with it the
        # parameters' AccumulateGrad nodes and their stream) alive into the next step
        return y_pred.detach(), loss.detach(), total_loss.detach()

    def add_auxiliary_loss(self, aux_loss, alpha):
        self.aux_loss = aux_loss * alpha
        self._aux_unset = False

    # ------------------------------------------------------------------ compile
    def compile(self, optimizer, loss=None, metrics=None):
        self.metrics_names = ["loss"]
        self._optim_capturable = False
        self._flush_optim()                # a previous optimizer may still owe table rows their latest steps
        self.optim = self._get_optim(optimizer)
        from .optim import TableAdam
        if isinstance(self.optim, TableAdam):          # also when handed in as an object, e.g. TableAdam(..., lazy_rows=True)
            self._optim_capturable = all(p.is_cuda for g in self.optim.param_groups for p in g["params"])
        self.loss_func = self._get_loss_func(loss)
        self.metrics = self._get_metrics(metrics)

    def _get_optim(self, optimizer):
        if not isinstance(optimizer, str):
            return optimizer
        def adam(params):
            # same update rule as basemodel.py:452; on a GPU use torch's single-pass multi-tensor kernel
            params = list(params)
            on_gpu = len(params) > 0 and all(p.is_cuda for p in params)
            # capturable: the step counters live on the device, so the step can be replayed from a HIP graph
            self._optim_capturable = on_gpu
            if on_gpu:
                from .optim import TableAdam
                return TableAdam(params)
            return torch.optim.Adam(params)
        table = {"sgd": lambda p: torch.optim.SGD(p, lr=0.01), "adam": adam,
                 "adagrad": torch.optim.Adagrad, "rmsprop": torch.optim.RMSprop}
        if optimizer not in table:
            raise NotImplementedError
        return table[optimizer](self.parameters())

    def _get_loss_func(self, loss):
        names = {"binary_crossentropy": F.binary_cross_entropy, "mse": F.mse_loss, "mae": F.l1_loss}

        def one(name):
            if name not in names:
                raise NotImplementedError
            return names[name]
        if isinstance(loss, str):
            return one(loss)
        if isinstance(loss, list):
            return [one(l) for l in loss]
        return loss

    @staticmethod
    def _accuracy_score(y_true, y_pred):
        return M.accuracy_score(y_true, np.where(y_pred > 0.5, 1, 0))

    def _get_metrics(self, metrics, set_eps=False):
        out = {}
        for name in (metrics or []):
            if name in ("binary_crossentropy", "logloss"):
                out[name] = M.log_loss
            if name == "auc":
                out[name] = M.roc_auc_score
            if name == "mse":
                out[name] = M.mean_squared_error
            if name in ("accuracy", "acc"):
                out[name] = self._accuracy_score
            self.metrics_names.append(name)
        return out

    def _in_multi_worker_mode(self):
        return None

    # ------------------------------------------------------------------ data plumbing
    def _as_matrix(self, x):
        """dict / list of per-feature arrays -> one [N, n_cols] array in feature_index order
        (basemodel.py:155-156,191-197)."""
        if isinstance(x, dict):
            x = [x[name] for name in self.feature_index]
        x = [np.expand_dims(a, axis=1) if len(a.shape) == 1 else a for a in x]
        return np.concatenate(x, axis=-1)

    def _resident(self, x, y=None):
        """The packed inputs as float32 tensors -- what `x.to(device).float()` (basemodel.py:242-243) yields
        batch by batch -- placed on the model's device once when they fit, otherwise kept on the host."""
        X = torch.from_numpy(np.ascontiguousarray(self._as_matrix(x))).float()
        Y = None if y is None else torch.from_numpy(np.asarray(y)).float()
        if X.numel() * 4 <= RESIDENT_LIMIT_BYTES:
            X = X.to(self.device)
            Y = None if Y is None else Y.to(self.device)
        return X, Y

    @staticmethod
    def _rows(t, order, start, stop):
        if order is None:
            return t[start:stop]
        return t.index_select(0, order[start:stop])

    # ------------------------------------------------------------------ fit / evaluate / predict
    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, initial_epoch=0, validation_split=0.,
            validation_data=None, shuffle=True, callbacks=None):
        """Training loop with the observable behaviour of basemodel.py:137-309: same batch order for a
        given torch seed (the stock DataLoader draws it), BCE(sum) + L2, History keys, callbacks."""
        if isinstance(x, dict):
            x = [x[name] for name in self.feature_index]
        do_validation = False
        val_x, val_y = [], []
        if validation_data:
            do_validation = True
            if len(validation_data) == 2:
                val_x, val_y = validation_data
            elif len(validation_data) == 3:
                val_x, val_y, _ = validation_data
            else:
                raise ValueError('When passing a `validation_data` argument, it must contain either 2 items '
                                 '(x_val, y_val), or 3 items (x_val, y_val, val_sample_weights). '
                                 'However we received `validation_data=%s`' % (validation_data,))
            if isinstance(val_x, dict):
                val_x = [val_x[name] for name in self.feature_index]
        elif validation_split and 0. < validation_split < 1.:
            do_validation = True
            n0 = x[0].shape[0] if hasattr(x[0], 'shape') else len(x[0])
            split_at = int(n0 * (1. - validation_split))
            x, val_x = _slice(x, 0, split_at), _slice(x, split_at)
            y, val_y = _slice(y, 0, split_at), _slice(y, split_at)

        X_all, Y_all = self._resident(x, y)
        if batch_size is None:
            batch_size = 256
        self.train()
        dp = xdist.current()
        if dp is not None and verbose > 0:
            print("row-parallel on %d ranks (RCCL), global batch %d" % (dp.world, batch_size * dp.world))
        if dp is None:
            print(self.device)
        global_bs = batch_size * (dp.world if dp is not None else 1)   # `batch_size` is per GPU, as basemodel.py:209
        sample_num = X_all.shape[0]
        steps_per_epoch = (sample_num - 1) // global_bs + 1

        cbs = CallbackList((callbacks or []) + [self.history])
        cbs.set_model(self)
        cbs.on_train_begin()
        cbs.set_model(self)
        self.stop_training = False
        print("Train on {0} samples, validate on {1} samples, {2} steps per epoch".format(
            sample_num, len(val_y), steps_per_epoch))
        loss_log = metric_log = None
        for epoch in range(initial_epoch, epochs):
            cbs.on_epoch_begin(epoch)
            epoch_logs, train_result = {}, {}
            t_epoch = time.time()
            total_loss_epoch = 0.0
            step_no = 0
            logged = {}
            order = epoch_order(sample_num, shuffle)
            if order is not None:
                order = order.to(X_all.device)
            it = range(0, sample_num, global_bs)
            bar = tqdm(it, disable=verbose != 1) if tqdm is not None else None
            try:
                for start in (bar if bar is not None else it):
                    xb = self._rows(X_all, order, start, start + global_bs)
                    yb = self._rows(Y_all, order, start, start + global_bs)
                    if dp is not None:
                        xb, yb = dp.shard(xb), dp.shard(yb)
                        self.__dict__["_row_weight"] = dp.row_weight()
                    xd = xb.to(self.device)
                    yd = yb.to(self.device)
                    y_pred, loss, total_loss = self.train_on_batch(xd, yd)
                    if dp is not None and dp.row_weight() == 0.0:
                        # stand-in row of a rank without rows in this (tail) batch: contributes nothing
                        total_loss = total_loss - loss
                        loss = loss * 0.0
                    # The reference reads the loss back every step (`total_loss.item()`, basemodel.py:262): a host
                    # sync per step that leaves the GPU idle while the next step is being enqueued.  The values are
                    # parked on the device instead and read once per epoch, summed in the same order in double.
                    if loss_log is None or loss_log.device != total_loss.device:
                        loss_log = torch.empty((steps_per_epoch + 1, 3), dtype=torch.float32, device=total_loss.device)
                    loss_log[step_no, 0:1].copy_(loss.detach().reshape(1))
                    loss_log[step_no, 1:2].copy_(total_loss.detach().reshape(1))
                    step_log = self.__dict__.get("_step_log")           # (name, value) a model wants summed per epoch
                    if step_log is not None:
                        loss_log[step_no, 2:3].copy_(step_log[1].reshape(1))
                    step_no += 1
                    if verbose > 0:
                        yt, yp = yd, y_pred
                        if dp is not None:
                            yt, yp = dp.gather_rows(yd.reshape(-1)), dp.gather_rows(y_pred.detach().reshape(-1))
                        for k, (name, fn) in enumerate(self.metrics.items()):
                            dev_fn = M.DEVICE.get(fn) if yt.is_cuda else None
                            if dev_fn is None:
                                train_result.setdefault(name, []).append(
                                    fn(yt.cpu().data.numpy(), yp.cpu().data.numpy().astype("float64")))
                                continue
                            # same metric on the device, parked beside the losses and read once per epoch
                            if metric_log is None or metric_log.device != yt.device:
                                metric_log = torch.empty((steps_per_epoch + 1, max(len(self.metrics), 1)),
                                                         dtype=torch.float64, device=yt.device)
                            metric_log[step_no - 1, k:k + 1].copy_(dev_fn(yt, yp.detach()).reshape(1))
                            logged[name] = k
            except KeyboardInterrupt:
                if bar is not None:
                    bar.close()
                raise
            if bar is not None:
                bar.close()
            self.__dict__["_row_weight"] = 1.0
            if step_no:
                data_l, total_l = loss_log[:step_no, 0], loss_log[:step_no, 1]
                if dp is not None:
                    # the data loss is a SUM over the global batch; the L2 part (total - data) is the same on every rank
                    data_sum = dp.all_reduce_sum(data_l.clone())
                    vals = [a + (t - d) for a, t, d in zip(data_sum.tolist(), total_l.tolist(), data_l.tolist())]
                else:
                    vals = total_l.tolist()
                for v in vals:
                    total_loss_epoch += v
            self._raise_on_bad_ids()          # deferred IndexError of the epoch's gathers (host is in sync here anyway)
            if hasattr(self.optim, "take_backlog"):
                # deferred table update: the L2 value of the steps a row was updated late for belongs to this epoch's sum
                self.optim.flush()
                total_loss_epoch += self.optim.take_backlog()
            epoch_logs["loss"] = total_loss_epoch / sample_num
            step_log = self.__dict__.get("_step_log")
            if step_log is not None and step_no:     # e.g. "sfg_loss": sum of the steps' values / sample_num (basemodel_sfg.py:365-366)
                extra_sum = 0.0
                for v in loss_log[:step_no, 2].tolist():
                    extra_sum += v
                epoch_logs[step_log[0]] = extra_sum / sample_num
            for name, k in logged.items():
                vals = metric_log[:step_no, k].tolist()
                if name == "auc" and any(v != v for v in vals):
                    raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
                train_result[name] = vals
            for name in self.metrics:                              # History keys in the order `compile` was given
                if name in train_result:
                    epoch_logs[name] = np.sum(train_result[name]) / steps_per_epoch
            if do_validation:
                for name, val in self.evaluate(val_x, val_y, batch_size).items():
                    epoch_logs["val_" + name] = val
            if verbose > 0 and (dp is None or dp.rank == 0):
                msg = "{0}s - loss: {1: .4f}".format(int(time.time() - t_epoch), epoch_logs["loss"])
                if "sfg_loss" in epoch_logs:
                    msg += " - sfg_loss: {0: .4f}".format(epoch_logs["sfg_loss"])
                for name in self.metrics:
                    msg += " - " + name + ": {0: .4f}".format(epoch_logs[name])
                if do_validation:
                    for name in self.metrics:
                        msg += " - val_" + name + ": {0: .4f}".format(epoch_logs["val_" + name])
                print('Epoch {0}/{1}'.format(epoch + 1, epochs))
                print(msg)
            if dp is None or dp.rank == 0:
                cbs.on_epoch_end(epoch, epoch_logs)
            else:
                self.history.on_epoch_end(epoch, epoch_logs)
            if dp is not None:
                self.stop_training = dp.any_flag(self.stop_training)
            if self.stop_training:
                break
        cbs.on_train_end()
        return self.history

    def evaluate(self, x, y, batch_size=256):
        pred = self.predict(x, batch_size)
        return {name: fn(y, pred) for name, fn in self.metrics.items()}

    def predict(self, x, batch_size=256):
        """float64 [N, 1] predictions (basemodel.py:325-352)."""
        self.eval()
        X_all, _ = self._resident(x)
        chunks = []
        with torch.no_grad():
            for start in range(0, X_all.shape[0], batch_size):
                chunks.append(self(X_all[start:start + batch_size].to(self.device)))
        if not chunks:
            return np.zeros((0, 1), dtype="float64")
        out = torch.cat(chunks).cpu().data.numpy().astype("float64")      # one device-to-host copy
        self._raise_on_bad_ids()
        return out


# ------------------------------------------------------------------------------------------------- #
class _XDeepFMBase(BaseModel):
    """Shared wiring of the three variants (deepctr/models/xdeepfm.py:42-107)."""

    def _build_dnn(self, dnn_feature_columns, dnn_hidden_units, dnn_activation, l2_reg_dnn, dnn_dropout, dnn_use_bn,
                   init_std, device):
        self.dnn_hidden_units = dnn_hidden_units
        self.use_dnn = len(dnn_feature_columns) > 0 and len(dnn_hidden_units) > 0
        if self.use_dnn:
            self.dnn = DNN(self.compute_input_dim(dnn_feature_columns), dnn_hidden_units, activation=dnn_activation,
                           l2_reg=l2_reg_dnn, dropout_rate=dnn_dropout, use_bn=dnn_use_bn, init_std=init_std,
                           device=device)
            self.dnn_linear = nn.Linear(dnn_hidden_units[-1], 1, bias=False).to(device)
            self.add_regularization_weight(
                filter(lambda x: 'weight' in x[0] and 'bn' not in x[0], self.dnn.named_parameters()), l2=l2_reg_dnn)
            self.add_regularization_weight(self.dnn_linear.weight, l2=l2_reg_dnn)

    def _finish_cin(self, cin_out_dim, l2_reg_cin, device):
        self.cin_linear = nn.Linear(cin_out_dim, 1, bias=False).to(device)
        self.add_regularization_weight(filter(lambda x: 'weight' in x[0], self.cin.named_parameters()), l2=l2_reg_cin)

    def head_inputs(self, X):
        """(linear logit [B,1], CIN output [B, featuremap_num] or None, DNN output [B, hidden] or None): what the
        last stage (cin_linear, dnn_linear, sum, PredictionLayer; deepctr/models/xdeepfm.py:95-107) consumes."""
        emb_fm, dnn_in, logit = self.fused_inputs(X)
        B = X.shape[0]
        cin_out = self.cin.forward_fm(emb_fm, B, self._plan.D) if self.use_cin else None
        dnn_out = self.dnn(dnn_in) if self.use_dnn else None
        return logit, cin_out, dnn_out

    def forward(self, X):
        logit, cin_out, dnn_out = self.head_inputs(X)
        if cin_out is not None:
            logit = logit + self.cin_linear(cin_out)
        if dnn_out is not None:
            logit = logit + self.dnn_linear(dnn_out)
        return self.out(logit)


class xDeepFM(_XDeepFMBase):
    """xDeepFM (deepctr/models/xdeepfm.py:17-107)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 256),
                 cin_layer_size=(256, 128,), cin_split_half=True, cin_activation='relu', l2_reg_linear=0.00001,
                 l2_reg_embedding=0.00001, l2_reg_dnn=0, l2_reg_cin=0, init_std=0.0001, seed=1024, dnn_dropout=0,
                 dnn_activation='relu', dnn_use_bn=False, task='binary', device='cpu', gpus=None):
        super().__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                         l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task, device=device,
                         gpus=gpus)
        self._build_dnn(dnn_feature_columns, dnn_hidden_units, dnn_activation, l2_reg_dnn, dnn_dropout, dnn_use_bn,
                        init_std, device)
        self.cin_layer_size = cin_layer_size
        self.use_cin = len(cin_layer_size) > 0 and len(dnn_feature_columns) > 0
        if self.use_cin:
            self.featuremap_num = (sum(cin_layer_size[:-1]) // 2 + cin_layer_size[-1]) if cin_split_half \
                else sum(cin_layer_size)
            self.cin = CIN(len(self.embedding_dict), cin_layer_size, cin_activation, cin_split_half, l2_reg_cin, seed,
                           device=device)
            self._finish_cin(self.featuremap_num, l2_reg_cin, device)
        self.to(device)


class xDeepFMAttention(_XDeepFMBase):
    """xDeepFM whose CIN ends in attention pooling (deepctr/models/xdeepfm_attn.py:25-173)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 256),
                 cin_layer_size=(256, 128,), cin_split_half=True, cin_activation='relu', cin_num_heads=4,
                 cin_attn_dropout=0.0, cin_use_layer_norm=True, cin_use_residual=True, l2_reg_linear=0.00001,
                 l2_reg_embedding=0.00001, l2_reg_dnn=0, l2_reg_cin=0, init_std=0.0001, seed=1024, dnn_dropout=0,
                 dnn_activation='relu', dnn_use_bn=False, task='binary', device='cpu', gpus=None):
        super().__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                         l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task, device=device,
                         gpus=gpus)
        self._build_dnn(dnn_feature_columns, dnn_hidden_units, dnn_activation, l2_reg_dnn, dnn_dropout, dnn_use_bn,
                        init_std, device)
        self.cin_layer_size = cin_layer_size
        self.use_cin = len(cin_layer_size) > 0 and len(dnn_feature_columns) > 0
        if self.use_cin:
            emb = next(fc.embedding_dim for fc in dnn_feature_columns if isinstance(fc, SparseFeat))
            self.featuremap_num = (sum(cin_layer_size[:-1]) // 2 + cin_layer_size[-1]) if cin_split_half \
                else sum(cin_layer_size)
            self.cin = CINAttention(field_size=len(self.embedding_dict), embedding_size=emb,
                                    layer_size=cin_layer_size, activation=cin_activation, split_half=cin_split_half,
                                    num_heads=cin_num_heads, attn_dropout=cin_attn_dropout,
                                    use_layer_norm=cin_use_layer_norm, use_residual=cin_use_residual,
                                    l2_reg=l2_reg_cin, seed=seed, device=device)
            self._finish_cin(self.featuremap_num, l2_reg_cin, device)
        self.to(device)


class xDeepFMAttentionV2(_XDeepFMBase):
    """Variant without the output projection: the CIN block returns [B, D]
    (deepctr/models/xdeepfm_attn.py:176-301)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 256),
                 cin_layer_size=(256, 128,), cin_split_half=True, cin_activation='relu', cin_num_heads=4,
                 cin_attn_dropout=0.0, cin_use_layer_norm=True, cin_use_residual=True, cin_num_attn_layers=1,
                 l2_reg_linear=0.00001, l2_reg_embedding=0.00001, l2_reg_dnn=0, l2_reg_cin=0, init_std=0.0001,
                 seed=1024, dnn_dropout=0, dnn_activation='relu', dnn_use_bn=False, task='binary', device='cpu',
                 gpus=None):
        super().__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                         l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task, device=device,
                         gpus=gpus)
        self._build_dnn(dnn_feature_columns, dnn_hidden_units, dnn_activation, l2_reg_dnn, dnn_dropout, dnn_use_bn,
                        init_std, device)
        self.cin_layer_size = cin_layer_size
        self.use_cin = len(cin_layer_size) > 0 and len(dnn_feature_columns) > 0
        if self.use_cin:
            emb = next(fc.embedding_dim for fc in dnn_feature_columns if isinstance(fc, SparseFeat))
            self.cin = CINAttentionV2(field_size=len(self.embedding_dict), embedding_size=emb,
                                      layer_size=cin_layer_size, activation=cin_activation,
                                      split_half=cin_split_half, num_heads=cin_num_heads,
                                      attn_dropout=cin_attn_dropout, use_layer_norm=cin_use_layer_norm,
                                      use_residual=cin_use_residual, num_attn_layers=cin_num_attn_layers,
                                      l2_reg=l2_reg_cin, seed=seed, device=device)
            self._finish_cin(emb, l2_reg_cin, device)
        self.to(device)
