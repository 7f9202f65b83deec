"""nn.Module front-ends of the HIP kernels, with the reference's constructor signatures and
state_dict keys (deepctr/layers/interaction.py:159-248, deepctr/layers/cin_attention.py,
deepctr/layers/core.py:67-160).  Parameters live in stock torch containers (nn.Conv1d,
nn.Linear, nn.LayerNorm) so that initial values under a given seed and checkpoint key names are
the reference's; their `forward` is never used -- the arithmetic runs in libxdfm_hip.so.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _valid_num_heads(embed_dim, num_heads):
    """Largest head count <= num_heads dividing embed_dim (deepctr/layers/cin_attention.py:15-23)."""
    for h in range(num_heads, 0, -1):
        if embed_dim % h == 0:
            return h
    return 1


def _cin_convs(field_size, layer_size, split_half):
    """The Conv1d parameter holders and field_nums of deepctr/layers/interaction.py:189-201."""
    if len(layer_size) == 0:
        raise ValueError("layer_size must be a list(tuple) of length greater than 1")
    convs, field_nums = nn.ModuleList(), [field_size]
    for i, size in enumerate(layer_size):
        convs.append(nn.Conv1d(field_nums[-1] * field_nums[0], size, 1))
        if split_half:
            if i != len(layer_size) - 1 and size % 2 > 0:
                raise ValueError("layer_size must be even number except for the last layer when split_half=True")
            field_nums.append(size // 2)
        else:
            field_nums.append(size)
    return convs, field_nums


def _featuremap_num(layer_size, split_half):
    return sum(layer_size[:-1]) // 2 + layer_size[-1] if split_half else sum(layer_size)


class _CINBase(nn.Module):
    def _levels(self, x0_fm, B, D, pool):
        return ops.cin_stack(x0_fm, B, D, self.layer_size, self.split_half, self.activation, pool,
                             [c.weight for c in self.conv1ds], [c.bias for c in self.conv1ds])

    @staticmethod
    def _check3d(inputs):
        if len(inputs.shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(inputs.shape)))


class CIN(_CINBase):
    """Compressed Interaction Network; [B, field_size, D] -> [B, featuremap_num].

    Drop-in for deepctr/layers/interaction.py:159-248.  `forward_fm` takes the FM-layout tensor the
    fused gather produces and skips the layout change."""

    def __init__(self, field_size, layer_size=(128, 128), activation='relu', split_half=True, l2_reg=1e-5, seed=1024,
                 device='cpu'):
        super().__init__()
        self.layer_size = tuple(layer_size)
        self.split_half = split_half
        self.activation = activation
        ops.activation_code(activation)
        self.l2_reg = l2_reg
        self.seed = seed
        self.conv1ds, self.field_nums = _cin_convs(field_size, self.layer_size, split_half)
        self.to(device)

    def forward(self, inputs):
        self._check3d(inputs)
        B, _, D = inputs.shape
        return self.forward_fm(ops.to_fm_layout(inputs), B, D)

    def forward_fm(self, x0_fm, B, D):
        return self._levels(x0_fm, B, D, "sum")


class MultiHeadSelfAttention(nn.Module):
    """Parameter holder + reference arithmetic of deepctr/layers/cin_attention.py:26-97."""

    def __init__(self, embed_dim, num_heads=4, dropout=0.0, device='cpu'):
        super().__init__()
        num_heads = _valid_num_heads(embed_dim, num_heads)
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        self.scale = math.sqrt(self.head_dim)
        self.W_q = nn.Linear(embed_dim, embed_dim, bias=False)
        self.W_k = nn.Linear(embed_dim, embed_dim, bias=False)
        self.W_v = nn.Linear(embed_dim, embed_dim, bias=False)
        self.W_o = nn.Linear(embed_dim, embed_dim, bias=False)
        self.dropout = nn.Dropout(dropout)
        for lin in (self.W_q, self.W_k, self.W_v, self.W_o):
            nn.init.xavier_uniform_(lin.weight)
        self.to(device)

    def forward(self, x):
        B, S, _ = x.shape
        heads = lambda t: t.view(B, S, self.num_heads, self.head_dim).transpose(1, 2)
        q, k, v = heads(self.W_q(x)), heads(self.W_k(x)), heads(self.W_v(x))
        p = self.dropout(F.softmax(torch.matmul(q, k.transpose(-2, -1)) / self.scale, dim=-1))
        o = torch.matmul(p, v).transpose(1, 2).contiguous().view(B, S, self.embed_dim)
        return self.W_o(o)


class AttentionPooling(nn.Module):
    """deepctr/layers/cin_attention.py:100-144: softmax_seq(w2 . tanh(W1 x + b1)) weighted sum."""

    def __init__(self, embed_dim, hidden_dim=None, device='cpu'):
        super().__init__()
        hidden_dim = hidden_dim or embed_dim
        self.attention = nn.Sequential(nn.Linear(embed_dim, hidden_dim), nn.Tanh(),
                                       nn.Linear(hidden_dim, 1, bias=False))
        for mod in self.attention:
            if isinstance(mod, nn.Linear):
                nn.init.xavier_uniform_(mod.weight)
                if mod.bias is not None:
                    nn.init.zeros_(mod.bias)
        self.to(device)

    def forward(self, x):
        w = F.softmax(self.attention(x), dim=1)
        return torch.sum(w * x, dim=1)


class CINAttention(_CINBase):
    """CIN whose sum pooling is replaced by MHSA -> (+residual) -> LayerNorm -> attention pooling ->
    projection (deepctr/layers/cin_attention.py:147-318); [B, field_size, D] -> [B, featuremap_num]."""

    def __init__(self, field_size, embedding_size, layer_size=(128, 128), activation='relu', split_half=True,
                 num_heads=4, attn_dropout=0.0, use_layer_norm=True, use_residual=True, l2_reg=1e-5, seed=1024,
                 device='cpu'):
        super().__init__()
        self.layer_size = tuple(layer_size)
        self.split_half = split_half
        self.activation = activation
        ops.activation_code(activation)
        self.l2_reg, self.seed = l2_reg, seed
        self.embedding_size = embedding_size
        self.use_layer_norm, self.use_residual = use_layer_norm, use_residual
        self.conv1ds, self.field_nums = _cin_convs(field_size, self.layer_size, split_half)
        self.featuremap_num = _featuremap_num(self.layer_size, split_half)
        self.mhsa = MultiHeadSelfAttention(embedding_size, num_heads, attn_dropout, device)
        if use_layer_norm:
            self.layer_norm = nn.LayerNorm(embedding_size)
        self.attn_pooling = AttentionPooling(embedding_size, embedding_size, device)
        self.output_proj = nn.Linear(embedding_size, self.featuremap_num, bias=False)
        nn.init.xavier_uniform_(self.output_proj.weight)
        self.to(device)

    def forward(self, inputs):
        self._check3d(inputs)
        B, _, D = inputs.shape
        return self.forward_fm(ops.to_fm_layout(inputs), B, D)

    def forward_fm(self, x0_fm, B, D):
        fm = self._levels(x0_fm, B, D, "fm")                  # [S, B*D], FM layout
        pooled = ops.attn_pool(fm, B, D, [self.mhsa], [self.layer_norm] if self.use_layer_norm else None,
                               self.attn_pooling, self.use_residual)
        return self.output_proj(pooled)


class CINAttentionV2(_CINBase):
    """deepctr/layers/cin_attention.py:321-466: N x (MHSA, residual, LayerNorm) then attention pooling;
    [B, field_size, D] -> [B, D]."""

    def __init__(self, field_size, embedding_size, layer_size=(128, 128), activation='relu', split_half=True,
                 num_heads=4, attn_dropout=0.0, use_layer_norm=True, use_residual=True, num_attn_layers=1,
                 l2_reg=1e-5, seed=1024, device='cpu'):
        super().__init__()
        self.layer_size = tuple(layer_size)
        self.split_half = split_half
        self.activation = activation
        ops.activation_code(activation)
        self.l2_reg, self.seed = l2_reg, seed
        self.embedding_size = embedding_size
        self.use_layer_norm, self.use_residual = use_layer_norm, use_residual
        self.num_attn_layers = num_attn_layers
        self.conv1ds, self.field_nums = _cin_convs(field_size, self.layer_size, split_half)
        self.featuremap_num = _featuremap_num(self.layer_size, split_half)
        self.mhsa_layers = nn.ModuleList()
        self.layer_norms = nn.ModuleList() if use_layer_norm else None
        for _ in range(num_attn_layers):
            self.mhsa_layers.append(MultiHeadSelfAttention(embedding_size, num_heads, attn_dropout, device))
            if use_layer_norm:
                self.layer_norms.append(nn.LayerNorm(embedding_size))
        self.attn_pooling = AttentionPooling(embedding_size, embedding_size, device)
        self.to(device)

    def forward(self, inputs):
        self._check3d(inputs)
        B, _, D = inputs.shape
        return self.forward_fm(ops.to_fm_layout(inputs), B, D)

    def forward_fm(self, x0_fm, B, D):
        fm = self._levels(x0_fm, B, D, "fm")
        return ops.attn_pool(fm, B, D, list(self.mhsa_layers),
                             list(self.layer_norms) if self.use_layer_norm else None, self.attn_pooling,
                             self.use_residual)


class DNN(nn.Module):
    """ReLU / linear MLP (deepctr/layers/core.py:67-134).  The GEMMs are far below 2 % of the step's
    FLOPs and go to hipBLASLt through torch (ops.Dense: bias gradient by the library's memset-free column
    sum, so that the step can be captured in a HIP graph); Dice / PReLU belong to other models of the zoo."""

    def __init__(self, inputs_dim, hidden_units, activation='relu', l2_reg=0, dropout_rate=0, use_bn=False,
                 init_std=0.0001, dice_dim=3, seed=1024, device='cpu'):
        super().__init__()
        if len(hidden_units) == 0:
            raise ValueError("hidden_units is empty!!")
        if not isinstance(activation, str) or activation.lower() not in ("relu", "linear", "sigmoid"):
            raise NotImplementedError("DNN activation %r is outside the xDeepFM path" % (activation,))
        self.activation = activation.lower()
        self.dropout_rate, self.seed, self.l2_reg, self.use_bn = dropout_rate, seed, l2_reg, use_bn
        self.dropout = nn.Dropout(dropout_rate)
        dims = [inputs_dim] + list(hidden_units)
        self.linears = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        if use_bn:
            self.bn = nn.ModuleList([nn.BatchNorm1d(dims[i + 1]) for i in range(len(dims) - 1)])
        for name, p in self.linears.named_parameters():
            if 'weight' in name:
                nn.init.normal_(p, mean=0, std=init_std)
        self.to(device)

    def forward(self, x):
        for i, lin in enumerate(self.linears):
            if self.activation == "relu" and not self.use_bn:
                x = self.dropout(ops.dense_relu(x, lin.weight, lin.bias))
                continue
            x = ops.dense(x, lin.weight, lin.bias)
            if self.use_bn:
                x = self.bn[i](x)
            if self.activation == "relu":
                x = torch.relu(x)
            elif self.activation == "sigmoid":
                x = torch.sigmoid(x)
            x = self.dropout(x)
        return x


class PredictionLayer(nn.Module):
    """logit (+bias) -> sigmoid for task 'binary' (deepctr/layers/core.py:137-160)."""

    def __init__(self, task='binary', use_bias=True, **kwargs):
        if task not in ["binary", "multiclass", "regression"]:
            raise ValueError("task must be binary,multiclass or regression")
        super().__init__()
        self.use_bias, self.task = use_bias, task
        if use_bias:
            self.bias = nn.Parameter(torch.zeros((1,)))

    def forward(self, X):
        out = X + self.bias if self.use_bias else X
        return torch.sigmoid(out) if self.task == "binary" else out
