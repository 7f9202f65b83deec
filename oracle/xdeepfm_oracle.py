"""CPU oracle for the xDeepFM hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a plain torch-CPU restatement of the arithmetic of the reference
(Syclus123/xDeepFM-pytorch, a vendored DeepCTR-Torch 0.2.9) for ONE path:

    sparse embedding gather -> CIN (outer product x 1x1 Conv1d stack)
    [-> multi-head self-attention pooling] (+ linear, DNN, sigmoid, BCE, L2, Adam)

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / timed CPU baseline.  The
shipped package (``xdeepfm-pytorch_amd/``) never imports anything from here and
fails loudly when its HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference from ``/root/reference`` in the build container, dumps its inputs /
weights / outputs / gradients to ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py`` checks every function below against them.

Every function cites the reference lines it restates (paths relative to
``/root/reference``).  Parameters travel in a flat ``dict`` that uses the
reference's ``state_dict`` key names, so a ``.pth`` written by either side can be
fed to the other.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# model description                                                            #
# --------------------------------------------------------------------------- #
@dataclass
class Spec:
    """Shape description of one xDeepFM-family model (the ctor arguments of
    deepctr/models/xdeepfm.py:42-45 and deepctr/models/xdeepfm_attn.py:55-61,185-192)."""
    sparse_names: List[str]
    vocab_sizes: List[int]
    dense_names: List[str]
    emb_dim: int
    cin_layer_size: Tuple[int, ...] = (256, 128)
    cin_split_half: bool = True
    cin_activation: str = "relu"
    dnn_hidden_units: Tuple[int, ...] = (256, 256)
    variant: str = "sum"            # "sum" (xDeepFM) | "attn" (xDeepFMAttention) | "attn_v2"
    num_heads: int = 4
    use_layer_norm: bool = True
    use_residual: bool = True
    num_attn_layers: int = 1
    l2_reg_linear: float = 1e-5
    l2_reg_embedding: float = 1e-5
    l2_reg_dnn: float = 0.0
    l2_reg_cin: float = 0.0

    @property
    def n_sparse(self) -> int:
        return len(self.sparse_names)

    @property
    def n_dense(self) -> int:
        return len(self.dense_names)

    @property
    def featuremap_num(self) -> int:
        # deepctr/models/xdeepfm.py:66-70
        if self.cin_split_half:
            return sum(self.cin_layer_size[:-1]) // 2 + self.cin_layer_size[-1]
        return sum(self.cin_layer_size)

    def feature_index(self) -> "OrderedDict[str, Tuple[int, int]]":
        # deepctr/inputs.py:99-123 with linear_cols == dnn_cols == sparse ++ dense
        # (xdftrain.py:247-256): sparse columns first, then dense, one column each.
        idx: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
        pos = 0
        for name in list(self.sparse_names) + list(self.dense_names):
            idx[name] = (pos, pos + 1)
            pos += 1
        return idx


def valid_num_heads(embed_dim: int, num_heads: int) -> int:
    """deepctr/layers/cin_attention.py:15-23"""
    if embed_dim % num_heads == 0:
        return num_heads
    for h in range(num_heads, 0, -1):
        if embed_dim % h == 0:
            return h
    return 1


# --------------------------------------------------------------------------- #
# embedding gather / linear part                                               #
# --------------------------------------------------------------------------- #
def embed_gather(X: torch.Tensor, state: Dict[str, torch.Tensor], spec: Spec) -> torch.Tensor:
    """[B, n_cols] float32 -> [B, m, D].

    deepctr/models/basemodel.py:368-370 (slice -> .long() -> nn.Embedding per field)
    followed by torch.cat(dim=1) of deepctr/models/xdeepfm.py:86."""
    fi = spec.feature_index()
    rows = []
    for name in spec.sparse_names:
        s, e = fi[name]
        ids = X[:, s:e].long()                                  # [B,1]
        rows.append(F.embedding(ids, state["embedding_dict.%s.weight" % name]))  # [B,1,D]
    return torch.cat(rows, dim=1)


def dense_values(X: torch.Tensor, spec: Spec) -> Optional[torch.Tensor]:
    """deepctr/models/basemodel.py:377-378 -> [B, n_dense] (or None)."""
    if not spec.dense_names:
        return None
    fi = spec.feature_index()
    return torch.cat([X[:, fi[n][0]:fi[n][1]] for n in spec.dense_names], dim=-1)


def linear_logit(X: torch.Tensor, state: Dict[str, torch.Tensor], spec: Spec) -> torch.Tensor:
    """deepctr/models/basemodel.py:63-92 -> [B,1]."""
    fi = spec.feature_index()
    out = torch.zeros([X.shape[0], 1], dtype=X.dtype)
    if spec.sparse_names:
        cols = []
        for name in spec.sparse_names:
            s, e = fi[name]
            cols.append(F.embedding(X[:, s:e].long(),
                                    state["linear_model.embedding_dict.%s.weight" % name]))  # [B,1,1]
        cat = torch.cat(cols, dim=-1)                            # [B,1,m]
        out = out + torch.sum(cat, dim=-1, keepdim=False)        # [B,1]
    dv = dense_values(X, spec)
    if dv is not None:
        out = out + dv.matmul(state["linear_model.weight"])
    return out


def combined_dnn_input(emb: torch.Tensor, dv: Optional[torch.Tensor]) -> torch.Tensor:
    """deepctr/inputs.py:126-138: [B,m,D] field-major flatten ++ dense -> [B, m*D + n_dense]."""
    flat = torch.flatten(emb, start_dim=1)
    if dv is None:
        return flat
    return torch.cat([flat, dv], dim=-1)


# --------------------------------------------------------------------------- #
# CIN                                                                          #
# --------------------------------------------------------------------------- #
def _act(x: torch.Tensor, name: str) -> torch.Tensor:
    # deepctr/layers/activation.py:57-84 (only the branches the path uses)
    name = name.lower()
    if name == "relu":
        return torch.relu(x)
    if name == "linear":
        return x
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise NotImplementedError(name)


def cin_feature_maps(x0: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
                     split_half: bool = True, activation: str = "relu") -> torch.Tensor:
    """The CIN level loop, returning the concatenated direct-connect feature maps
    [B, featuremap_num, D] *before* any pooling.

    deepctr/layers/interaction.py:216-245 (identical loop at
    deepctr/layers/cin_attention.py:257-292 and :417-449).
    weights[i]: [H_i, H'_{i-1}*m, 1] (nn.Conv1d k=1), biases[i]: [H_i]."""
    if x0.dim() != 3:
        raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % x0.dim())
    B, m, D = x0.shape
    hidden = x0
    direct = []
    n_levels = len(weights)
    for i in range(n_levels):
        size = weights[i].shape[0]
        z = torch.einsum("bhd,bmd->bhmd", hidden, x0)           # interaction.py:218
        z = z.reshape(B, hidden.shape[1] * m, D)                # :221  (k = h*m + j)
        z = F.conv1d(z, weights[i], biases[i])                  # :224
        cur = _act(z, activation)                               # :226-229
        if split_half:
            if i != n_levels - 1:
                nxt, dc = torch.split(cur, 2 * [size // 2], 1)  # :233  first half -> next hidden
            else:
                dc, nxt = cur, None
        else:
            dc, nxt = cur, cur
        direct.append(dc)
        hidden = nxt
    return torch.cat(direct, dim=1)                             # :245


def cin_forward(x0, weights, biases, split_half=True, activation="relu") -> torch.Tensor:
    """deepctr/layers/interaction.py:207-248 -> [B, featuremap_num] (sum over D, :246)."""
    return torch.sum(cin_feature_maps(x0, weights, biases, split_half, activation), -1)


# --------------------------------------------------------------------------- #
# attention pooling over the CIN feature maps                                  #
# --------------------------------------------------------------------------- #
def mhsa(x: torch.Tensor, wq, wk, wv, wo, num_heads: int, keep=None, p_drop: float = 0.0) -> torch.Tensor:
    """deepctr/layers/cin_attention.py:63-97 (bias-free projections).  keep: optional [B, heads, S, S] 0/1 mask
    standing for the draw of `self.dropout(attn_weights)` (:86, training mode): kept weights are scaled by
    1/(1-p_drop) as nn.Dropout does; None = evaluation mode / p = 0."""
    B, S, E = x.shape
    nh = valid_num_heads(E, num_heads)
    hd = E // nh
    q = F.linear(x, wq).view(B, S, nh, hd).transpose(1, 2)
    k = F.linear(x, wk).view(B, S, nh, hd).transpose(1, 2)
    v = F.linear(x, wv).view(B, S, nh, hd).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(hd)   # :84
    probs = F.softmax(scores, dim=-1)                                # :85
    if keep is not None:
        probs = probs * keep.to(probs.dtype) / (1.0 - p_drop)        # :86
    o = torch.matmul(probs, v)                                       # :89
    o = o.transpose(1, 2).contiguous().view(B, S, E)                 # :92
    return F.linear(o, wo)                                           # :95


def attention_pooling(x: torch.Tensor, w1, b1, w2) -> torch.Tensor:
    """deepctr/layers/cin_attention.py:130-144: [B,S,E] -> [B,E]."""
    s = F.linear(torch.tanh(F.linear(x, w1, b1)), w2)               # [B,S,1]
    a = F.softmax(s, dim=1)
    return torch.sum(a * x, dim=1)


def cin_attention_forward(x0, state: Dict[str, torch.Tensor], prefix: str, spec: Spec, keep=None,
                          p_drop: float = 0.0) -> torch.Tensor:
    """CINAttention.forward (deepctr/layers/cin_attention.py:239-318) -> [B, featuremap_num]
    or CINAttentionV2.forward (:399-466) -> [B, D] when spec.variant == 'attn_v2'.
    keep: optional [n_layers, B, heads, S, S] attention-dropout masks (see mhsa)."""
    L = len(spec.cin_layer_size)
    W = [state["%sconv1ds.%d.weight" % (prefix, i)] for i in range(L)]
    Bs = [state["%sconv1ds.%d.bias" % (prefix, i)] for i in range(L)]
    fm = cin_feature_maps(x0, W, Bs, spec.cin_split_half, spec.cin_activation)   # [B,S,D]
    E = fm.shape[-1]
    if spec.variant == "attn":
        a = mhsa(fm, state[prefix + "mhsa.W_q.weight"], state[prefix + "mhsa.W_k.weight"],
                 state[prefix + "mhsa.W_v.weight"], state[prefix + "mhsa.W_o.weight"], spec.num_heads,
                 None if keep is None else keep[0], p_drop)
        if spec.use_residual:
            a = a + fm                                                           # :305-306
        if spec.use_layer_norm:
            a = F.layer_norm(a, (E,), state[prefix + "layer_norm.weight"],
                             state[prefix + "layer_norm.bias"], 1e-5)            # :309-310
        pooled = attention_pooling(a, state[prefix + "attn_pooling.attention.0.weight"],
                                   state[prefix + "attn_pooling.attention.0.bias"],
                                   state[prefix + "attn_pooling.attention.2.weight"])
        return F.linear(pooled, state[prefix + "output_proj.weight"])            # :316
    if spec.variant == "attn_v2":
        r = fm
        for l in range(spec.num_attn_layers):                                    # :452-461
            p = "%smhsa_layers.%d." % (prefix, l)
            a = mhsa(r, state[p + "W_q.weight"], state[p + "W_k.weight"],
                     state[p + "W_v.weight"], state[p + "W_o.weight"], spec.num_heads,
                     None if keep is None else keep[l], p_drop)
            if spec.use_residual:
                a = a + r
            if spec.use_layer_norm:
                a = F.layer_norm(a, (E,), state["%slayer_norms.%d.weight" % (prefix, l)],
                                 state["%slayer_norms.%d.bias" % (prefix, l)], 1e-5)
            r = a
        return attention_pooling(r, state[prefix + "attn_pooling.attention.0.weight"],
                                 state[prefix + "attn_pooling.attention.0.bias"],
                                 state[prefix + "attn_pooling.attention.2.weight"])
    raise ValueError(spec.variant)


# --------------------------------------------------------------------------- #
# DNN / heads / whole model                                                    #
# --------------------------------------------------------------------------- #
def dnn_forward(x: torch.Tensor, state: Dict[str, torch.Tensor], n_layers: int,
                prefix: str = "dnn.") -> torch.Tensor:
    """deepctr/layers/core.py:120-134 with use_bn=False, dropout=0, relu."""
    for i in range(n_layers):
        x = torch.relu(F.linear(x, state["%slinears.%d.weight" % (prefix, i)],
                                state["%slinears.%d.bias" % (prefix, i)]))
    return x


def model_forward(X: torch.Tensor, state: Dict[str, torch.Tensor], spec: Spec) -> torch.Tensor:
    """xDeepFM.forward (deepctr/models/xdeepfm.py:79-107) and its attention twins
    (deepctr/models/xdeepfm_attn.py:143-173, 271-301) -> y_pred [B,1]."""
    emb = embed_gather(X, state, spec)
    logit = linear_logit(X, state, spec)
    use_cin = len(spec.cin_layer_size) > 0
    use_dnn = len(spec.dnn_hidden_units) > 0
    if use_cin:
        if spec.variant == "sum":
            L = len(spec.cin_layer_size)
            cin_out = cin_forward(emb, [state["cin.conv1ds.%d.weight" % i] for i in range(L)],
                                  [state["cin.conv1ds.%d.bias" % i] for i in range(L)],
                                  spec.cin_split_half, spec.cin_activation)
        else:
            cin_out = cin_attention_forward(emb, state, "cin.", spec)
        cin_logit = F.linear(cin_out, state["cin_linear.weight"])            # xdeepfm.py:88
    if use_dnn:
        dnn_in = combined_dnn_input(emb, dense_values(X, spec))              # xdeepfm.py:90
        dnn_out = dnn_forward(dnn_in, state, len(spec.dnn_hidden_units))
        dnn_logit = F.linear(dnn_out, state["dnn_linear.weight"])            # xdeepfm.py:92
    if use_dnn and use_cin:
        logit = logit + dnn_logit + cin_logit                                 # xdeepfm.py:100-101
    elif use_cin:
        logit = logit + cin_logit
    elif use_dnn:
        logit = logit + dnn_logit
    return torch.sigmoid(logit + state["out.bias"])                          # core.py:154-160


def regularization_groups(state: Dict[str, torch.Tensor], spec: Spec) -> List[Tuple[List[str], float]]:
    """Which tensors carry which L2 strength: deepctr/models/basemodel.py:126-127,
    deepctr/models/xdeepfm.py:57-60,74-75 (same lines in xdeepfm_attn.py:114-118,153-156)."""
    emb = ["embedding_dict.%s.weight" % n for n in spec.sparse_names]
    lin = ["linear_model.embedding_dict.%s.weight" % n for n in spec.sparse_names]
    if spec.dense_names:
        lin.append("linear_model.weight")
    groups = [(emb, spec.l2_reg_embedding), (lin, spec.l2_reg_linear)]
    if spec.dnn_hidden_units:
        groups.append((["dnn.linears.%d.weight" % i for i in range(len(spec.dnn_hidden_units))],
                       spec.l2_reg_dnn))
        groups.append((["dnn_linear.weight"], spec.l2_reg_dnn))
    if spec.cin_layer_size:
        groups.append(([k for k in state if k.startswith("cin.") and "weight" in k], spec.l2_reg_cin))
    return groups


def regularization_loss(state: Dict[str, torch.Tensor], spec: Spec) -> torch.Tensor:
    """deepctr/models/basemodel.py:412-428 (l1 == 0 everywhere on this path)."""
    total = torch.zeros((1,))
    for names, l2 in regularization_groups(state, spec):
        if l2 > 0:
            for n in names:
                total = total + torch.sum(l2 * torch.square(state[n]))
    return total


def total_loss(X, y, state, spec) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """One evaluation of the training objective of deepctr/models/basemodel.py:245-257:
    BCE(reduction='sum') on probabilities + L2.  Returns (total, data_loss, y_pred)."""
    y_pred = model_forward(X, state, spec).squeeze()
    loss = F.binary_cross_entropy(y_pred, y.squeeze(), reduction="sum")
    return loss + regularization_loss(state, spec), loss, y_pred


def train_steps(batches, state: Dict[str, torch.Tensor], spec: Spec, lr: float = 1e-3):
    """basemodel.py:241-262 for a list of (X, y) batches with torch.optim.Adam (basemodel.py:452).
    `state` tensors are updated in place; returns the list of (data_loss, total_loss) floats."""
    params = [p.requires_grad_(True) for p in state.values()]
    opt = torch.optim.Adam(params, lr=lr)
    log = []
    for X, y in batches:
        tot, dl, _ = total_loss(X, y, state, spec)
        opt.zero_grad()
        tot.backward()
        opt.step()
        log.append((float(dl.item()), float(tot.item())))
    return log


# --------------------------------------------------------------------------- #
# xdeepfm_pro: SFG decoder / loss, AutoDis                                     #
# --------------------------------------------------------------------------- #
@dataclass
class ProSpec:
    """The extra constructor arguments of deepctr/xdeepfm_pro/xdeepfm_pro.py:85-95."""
    sfg_weight: float = 0.1
    sfg_hidden_units: Tuple[int, ...] = (128, 64)
    sfg_positive_only: bool = True
    sfg_use_label_attention: bool = True
    use_autodis: bool = False


def autodis_forward(dense: torch.Tensor, state: Dict[str, torch.Tensor], prefix: str = "autodis_encoder.autodis.") -> torch.Tensor:
    """deepctr/xdeepfm_pro/autodis.py:99-127: per feature Linear(1,nb) -> LeakyReLU(0.2) -> Linear(nb,nb) -> softmax(./temp)
    -> mix of the bucket embeddings; [B, nd] -> [B, nd * D]."""
    outs = []
    for i in range(dense.shape[1]):
        v = dense[:, i:i + 1]
        h = F.leaky_relu(F.linear(v, state["%sbucket_projectors.%d.0.weight" % (prefix, i)],
                                  state["%sbucket_projectors.%d.0.bias" % (prefix, i)]), 0.2)
        sc = F.linear(h, state["%sbucket_projectors.%d.2.weight" % (prefix, i)], state["%sbucket_projectors.%d.2.bias" % (prefix, i)])
        w = F.softmax(sc / state[prefix + "feature_temperatures"][i], dim=-1)
        outs.append(torch.matmul(w, state[prefix + "meta_embeddings"][i]))
    return torch.cat(outs, dim=-1)


def sfg_loss(X: torch.Tensor, y: torch.Tensor, emb: torch.Tensor, state: Dict[str, torch.Tensor], spec: Spec,
             pro: ProSpec) -> torch.Tensor:
    """compute_sfg_loss + SFGDecoder.forward + SFGLoss.forward (deepctr/xdeepfm_pro/basemodel_sfg.py:420-476,
    sfg_decoder.py:113-157, :257-311), dropout 0: label-aware gate, shared MLP, one vocabulary-wide softmax head per sparse
    field with masked cross-entropy, one regression head with masked MSE."""
    B = X.shape[0]
    dv = dense_values(X, spec)
    dec_in = torch.cat([emb.reshape(B, -1)] + ([dv] if dv is not None else []), dim=-1)      # sfg_decoder.py:113-136
    labels = y.reshape(-1)
    pre = "sfg_decoder."
    if pro.sfg_use_label_attention:                                                           # sfg_decoder.py:184-206
        lab = state[pre + "label_attention.label_embedding.weight"][labels.long()]
        a = torch.relu(F.linear(torch.cat([dec_in, lab], dim=-1), state[pre + "label_attention.attention_net.0.weight"],
                                state[pre + "label_attention.attention_net.0.bias"]))
        gate = torch.sigmoid(F.linear(a, state[pre + "label_attention.attention_net.2.weight"],
                                      state[pre + "label_attention.attention_net.2.bias"]))
        dec_in = dec_in * gate
    h = dec_in
    for i in range(len(pro.sfg_hidden_units)):                                                # Linear, ReLU, Dropout(0) triples
        h = torch.relu(F.linear(h, state["%sshared_layers.%d.weight" % (pre, 3 * i)], state["%sshared_layers.%d.bias" % (pre, 3 * i)]))
    if pro.sfg_positive_only:                                                                 # sfg_decoder.py:262-268
        mask = (labels == 1).float()
        npos = mask.sum() + 1e-8
    else:
        mask = torch.ones_like(labels)
        npos = float(B)
    fi = spec.feature_index()
    total_sparse = torch.zeros(())
    for name in spec.sparse_names:                                                            # sfg_decoder.py:275-293
        logits = F.linear(h, state["%ssparse_heads.%s.weight" % (pre, name)], state["%ssparse_heads.%s.bias" % (pre, name)])
        ce = F.cross_entropy(logits, X[:, fi[name][0]].long(), reduction="none")
        total_sparse = total_sparse + (ce * mask).sum() / npos
    total_dense = torch.zeros(())
    if spec.dense_names:                                                                      # sfg_decoder.py:295-304
        preds = F.linear(h, state[pre + "dense_head.weight"], state[pre + "dense_head.bias"])
        mse = F.mse_loss(preds, dv, reduction="none").mean(dim=-1)
        total_dense = (mse * mask).sum() / npos
    return total_sparse + total_dense


def pro_forward_with_sfg(X, y, state, spec: Spec, pro: ProSpec, training: bool = True):
    """xDeepFMPro.forward_with_sfg (deepctr/xdeepfm_pro/xdeepfm_pro.py:203-274) -> (y_pred [B,1], sfg loss or None)."""
    emb = embed_gather(X, state, spec)
    logit = linear_logit(X, state, spec)
    if spec.cin_layer_size:
        L = len(spec.cin_layer_size)
        cin_out = cin_forward(emb, [state["cin.conv1ds.%d.weight" % i] for i in range(L)],
                              [state["cin.conv1ds.%d.bias" % i] for i in range(L)], spec.cin_split_half, spec.cin_activation)
        logit = logit + F.linear(cin_out, state["cin_linear.weight"])
    if spec.dnn_hidden_units:
        dv = dense_values(X, spec)
        if pro.use_autodis and dv is not None:
            dnn_in = torch.cat([emb.reshape(X.shape[0], -1), autodis_forward(dv, state)], dim=-1)   # xdeepfm_pro.py:236-242
        else:
            dnn_in = combined_dnn_input(emb, dv)
        logit = logit + F.linear(dnn_forward(dnn_in, state, len(spec.dnn_hidden_units)), state["dnn_linear.weight"])
    y_pred = torch.sigmoid(logit + state["out.bias"])
    sfg = sfg_loss(X, y, emb, state, spec, pro) if (training and y is not None) else None
    return y_pred, sfg


def pro_total_loss(X, y, state, spec: Spec, pro: ProSpec):
    """Objective of the BaseModelSFG.fit loop body (basemodel_sfg.py:327-343): BCE(sum) + L2 + sfg_weight * sfg_loss."""
    y_pred, sfg = pro_forward_with_sfg(X, y, state, spec, pro, True)
    loss = F.binary_cross_entropy(y_pred.squeeze(), y.squeeze(), reduction="sum")
    return loss + regularization_loss(state, spec) + pro.sfg_weight * sfg, loss, sfg, y_pred


# --------------------------------------------------------------------------- #
# initialisation with the reference's RNG order                                #
# --------------------------------------------------------------------------- #
def init_state(spec: Spec, seed: int = 1024, init_std: float = 1e-4) -> Dict[str, torch.Tensor]:
    """A fresh parameter dict drawn in the reference's construction order
    (BaseModel.__init__ basemodel.py:100-129 -> xDeepFM.__init__ xdeepfm.py:50-75 /
    xdeepfm_attn.py:88-158), so that seed 1024 gives the reference's initial weights.
    Built from stock torch.nn modules because their default initialisers are part of
    that order (nn.Embedding N(0,1), nn.Linear / nn.Conv1d kaiming-uniform)."""
    import torch.nn as nn
    torch.manual_seed(seed)                                                  # basemodel.py:100
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    D = spec.emb_dim
    # inputs.py:167-178 : default init then normal_(0, init_std)
    # (all tables are constructed first, then re-drawn in a second loop)
    embs = [(n, nn.Embedding(v, D)) for n, v in zip(spec.sparse_names, spec.vocab_sizes)]
    for n, e in embs:
        nn.init.normal_(e.weight, mean=0, std=init_std)
        st["embedding_dict.%s.weight" % n] = e.weight.detach()
    # Linear: tables (inputs.py:167-178), second normal_ pass (basemodel.py:55-56), dense w (:58-61)
    lin = [(n, nn.Embedding(v, 1)) for n, v in zip(spec.sparse_names, spec.vocab_sizes)]
    for n, e in lin:
        nn.init.normal_(e.weight, mean=0, std=init_std)
    for n, e in lin:
        nn.init.normal_(e.weight, mean=0, std=init_std)
    lin_w = None
    if spec.dense_names:
        lin_w = torch.Tensor(len(spec.dense_names), 1)
        nn.init.normal_(lin_w, mean=0, std=init_std)
    # state_dict order of Linear: its own Parameter first, then the sub-module's
    if lin_w is not None:
        st["linear_model.weight"] = lin_w
    for n, e in lin:
        st["linear_model.embedding_dict.%s.weight" % n] = e.weight.detach()
    st["out.bias"] = torch.zeros((1,))                                       # core.py:152
    # DNN (core.py:104-116)
    if spec.dnn_hidden_units:
        dims = [spec.n_sparse * D + spec.n_dense] + list(spec.dnn_hidden_units)
        linears = [nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
        for l in linears:
            nn.init.normal_(l.weight, mean=0, std=init_std)
        for i, l in enumerate(linears):
            st["dnn.linears.%d.weight" % i] = l.weight.detach()
            st["dnn.linears.%d.bias" % i] = l.bias.detach()
        st["dnn_linear.weight"] = nn.Linear(dims[-1], 1, bias=False).weight.detach()   # xdeepfm.py:56
    if spec.cin_layer_size:
        m = spec.n_sparse
        field_nums = [m]
        for i, size in enumerate(spec.cin_layer_size):                       # interaction.py:190-201
            c = nn.Conv1d(field_nums[-1] * m, size, 1)
            st["cin.conv1ds.%d.weight" % i] = c.weight.detach()
            st["cin.conv1ds.%d.bias" % i] = c.bias.detach()
            field_nums.append(size // 2 if spec.cin_split_half else size)
        if spec.variant in ("attn", "attn_v2"):
            def _mhsa(prefix):                                               # cin_attention.py:47-61
                ws = [nn.Linear(D, D, bias=False) for _ in range(4)]
                for w in ws:
                    nn.init.xavier_uniform_(w.weight)
                for nm, w in zip(("W_q", "W_k", "W_v", "W_o"), ws):
                    st["%s%s.weight" % (prefix, nm)] = w.weight.detach()

            def _pool(prefix):                                               # cin_attention.py:114-128
                a0 = nn.Linear(D, D)
                a2 = nn.Linear(D, 1, bias=False)
                nn.init.xavier_uniform_(a0.weight)
                nn.init.zeros_(a0.bias)
                nn.init.xavier_uniform_(a2.weight)
                st[prefix + "attention.0.weight"] = a0.weight.detach()
                st[prefix + "attention.0.bias"] = a0.bias.detach()
                st[prefix + "attention.2.weight"] = a2.weight.detach()

            if spec.variant == "attn":                                       # cin_attention.py:214-235
                _mhsa("cin.mhsa.")
                if spec.use_layer_norm:
                    st["cin.layer_norm.weight"] = torch.ones(D)
                    st["cin.layer_norm.bias"] = torch.zeros(D)
                _pool("cin.attn_pooling.")
                op = nn.Linear(D, spec.featuremap_num, bias=False)
                nn.init.xavier_uniform_(op.weight)
                st["cin.output_proj.weight"] = op.weight.detach()
            else:                                                            # cin_attention.py:375-395
                for l in range(spec.num_attn_layers):
                    _mhsa("cin.mhsa_layers.%d." % l)
                    if spec.use_layer_norm:
                        st["cin.layer_norms.%d.weight" % l] = torch.ones(D)
                        st["cin.layer_norms.%d.bias" % l] = torch.zeros(D)
                _pool("cin.attn_pooling.")
        cin_out_dim = spec.emb_dim if spec.variant == "attn_v2" else spec.featuremap_num
        st["cin_linear.weight"] = nn.Linear(cin_out_dim, 1, bias=False).weight.detach()   # xdeepfm.py:73
    return OrderedDict((k, v.clone()) for k, v in st.items())


# --------------------------------------------------------------------------- #
# metrics (sklearn restated in numpy: the GPU box need not have sklearn)       #
# --------------------------------------------------------------------------- #
def log_loss(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """sklearn.metrics.log_loss (1.7) for binary labels: clip to [eps, 1-eps] with
    eps = float64 machine epsilon, mean negative log-likelihood."""
    y = np.asarray(y_true, dtype=np.float64).ravel()
    p = np.asarray(y_pred, dtype=np.float64).ravel()
    eps = np.finfo(np.float64).eps
    p = np.clip(p, eps, 1 - eps)
    return float(-np.mean(y * np.log(p) + (1 - y) * np.log(1 - p)))


def roc_auc(y_true: np.ndarray, y_score: np.ndarray) -> float:
    """sklearn.metrics.roc_auc_score for binary labels = Mann-Whitney U with mid-ranks."""
    y = np.asarray(y_true).ravel() > 0.5
    s = np.asarray(y_score, dtype=np.float64).ravel()
    n_pos = int(y.sum())
    n_neg = y.size - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(s, kind="mergesort")
    ss = s[order]
    ranks = np.empty(s.size, dtype=np.float64)
    i = 0
    while i < ss.size:                    # mid-ranks for ties
        j = i
        while j + 1 < ss.size and ss[j + 1] == ss[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


# --------------------------------------------------------------------------- #
# synthetic Criteo-shaped batches                       #
# --------------------------------------------------------------------------- #
def synthetic_batch(n_rows: int, vocab_sizes: Sequence[int], n_dense: int, seed: int = 2025,
                    zipf: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """X [N, m + n_dense] float32 (ids then dense in [0,1)), y [N,1] float32 ~ Bernoulli(0.25)."""
    rng = np.random.default_rng(seed)
    cols = []
    for v in vocab_sizes:
        u = rng.random(n_rows)
        ids = np.floor(v * (u ** 3 if zipf else u)).astype(np.int64)
        cols.append(np.minimum(ids, v - 1).astype(np.float32))
    for _ in range(n_dense):
        cols.append(rng.random(n_rows).astype(np.float32))
    X = np.stack(cols, axis=1).astype(np.float32)
    y = (rng.random(n_rows) < 0.25).astype(np.float32).reshape(-1, 1)
    return X, y
