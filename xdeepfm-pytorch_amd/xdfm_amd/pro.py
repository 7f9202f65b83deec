"""xDeepFM "Pro": xDeepFM + Supervised Feature Generation (SFG) decoder + optional AutoDis, behind the interface of
deepctr/xdeepfm_pro (reference files cited per class).  Same constructor signatures, module structure and
state_dict keys as the reference, so `.pth` files interchange and `xdftrain_pro.py`'s flow runs unchanged.

What runs where:
  * embeddings / linear logit / DNN input: the fused gather K1; CIN: the MFMA kernels K3 / K4 -- exactly as xDeepFM;
  * the SFG decoder's input IS K1's `dnn_in` output ([sparse embeddings field-major | dense values],
    sfg_decoder.py:113-136), so the decoder costs no extra gather;
  * the 26 vocabulary-wide softmax heads + masked cross-entropy (sfg_decoder.py:146-149, :277-293) never materialise
    the [B, V_f] logits: `ops.VocabSoftmaxCE` walks the vocabulary in tiles with an online log-sum-exp and recomputes the
    tiles in the backward; with `sfg_positive_only` (the default) only the rows with label 1 enter the decoder at all --
    masked rows contribute exact zeros to loss and gradients in the reference too (`ce_loss * positive_mask`).  The tile
    GEMMs are library GEMMs (hipBLASLt through torch); a hand-written MFMA kernel for them is the next step;
  * the decoder MLP, LabelAwareAttention and AutoDis are small dense layers on torch ops.
The pro train step launches eagerly (the positive-row compaction has a data-dependent shape).
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .inputs import DenseFeat, SparseFeat
from .layers import CIN, DNN
from .models import BaseModel


# ------------------------------------------------------------------------------------------------- #
class LabelAwareAttention(nn.Module):
    """sigmoid gate from [input | label embedding] (deepctr/xdeepfm_pro/sfg_decoder.py:160-206)."""

    def __init__(self, input_dim: int, hidden_dim: int = 64, device: str = 'cpu'):
        super().__init__()
        self.label_embedding = nn.Embedding(2, hidden_dim)
        self.attention_net = nn.Sequential(nn.Linear(input_dim + hidden_dim, hidden_dim), nn.ReLU(),
                                           nn.Linear(hidden_dim, input_dim), nn.Sigmoid())
        self.to(device)

    def forward(self, x: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        if len(labels.shape) > 1:
            labels = labels.squeeze(-1)
        label_emb = self.label_embedding(labels.long())
        return self.attention_net(torch.cat([x, label_emb], dim=-1))


class SFGDecoder(nn.Module):
    """MLP decoder with one vocabulary-wide head per sparse field and one regression head for the dense fields
    (deepctr/xdeepfm_pro/sfg_decoder.py:19-157)."""

    def __init__(self, embedding_dim: int, sparse_feature_dims: Dict[str, int], dense_feature_names: List[str],
                 hidden_units: Tuple[int, ...] = (128, 64), dropout_rate: float = 0.1,
                 use_label_aware_attention: bool = True, device: str = 'cpu'):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.sparse_feature_dims = sparse_feature_dims
        self.dense_feature_names = dense_feature_names
        self.use_label_aware_attention = use_label_aware_attention
        self.device = device
        self.num_sparse_features = len(sparse_feature_dims)
        self.num_dense_features = len(dense_feature_names)
        input_dim = self.num_sparse_features * embedding_dim + self.num_dense_features
        layers, prev_dim = [], input_dim
        for hidden_dim in hidden_units:
            layers += [nn.Linear(prev_dim, hidden_dim), nn.ReLU(), nn.Dropout(dropout_rate)]
            prev_dim = hidden_dim
        self.shared_layers = nn.Sequential(*layers)
        self.sparse_heads = nn.ModuleDict()
        for feat_name, vocab_size in sparse_feature_dims.items():
            self.sparse_heads[feat_name] = nn.Linear(prev_dim, vocab_size)
        self.dense_head = nn.Linear(prev_dim, self.num_dense_features) if self.num_dense_features > 0 else None
        if use_label_aware_attention:
            self.label_attention = LabelAwareAttention(input_dim=input_dim,
                                                       hidden_dim=hidden_units[0] if hidden_units else 64, device=device)
        self.to(device)

    def hidden(self, decoder_input: torch.Tensor, labels: Optional[torch.Tensor]) -> torch.Tensor:
        """decoder input [rows, m*D + nd] -> last hidden layer (sfg_decoder.py:138-143)."""
        if self.use_label_aware_attention and labels is not None:
            decoder_input = decoder_input * self.label_attention(decoder_input, labels)
        return self.shared_layers(decoder_input)

    def forward(self, sparse_embeddings, dense_values, labels=None):
        """The reference's signature (lists of [B,1,D] / [B,1] tensors) -> (dict of [B, V_f] logits, [B, nd]).  This
        materialises the logits and exists for API compatibility; the models' loss goes through `hidden` + the tiled
        cross-entropy instead."""
        sparse_concat = torch.cat([e.squeeze(1) if len(e.shape) == 3 else e for e in sparse_embeddings], dim=-1)
        dense_concat = torch.cat(dense_values, dim=-1) if len(dense_values) > 0 else \
            torch.zeros(sparse_concat.shape[0], 0, device=sparse_concat.device)
        hidden = self.hidden(torch.cat([sparse_concat, dense_concat], dim=-1), labels)
        sparse_logits = {name: self.sparse_heads[name](hidden) for name in self.sparse_feature_dims.keys()}
        dense_preds = self.dense_head(hidden) if self.dense_head is not None else \
            torch.zeros(hidden.shape[0], 0, device=hidden.device)
        return sparse_logits, dense_preds


class SFGLoss(nn.Module):
    """Masked reconstruction loss: cross-entropy per sparse field + MSE over the dense fields, summed over the
    positive rows and divided by their number (deepctr/xdeepfm_pro/sfg_decoder.py:209-311)."""

    def __init__(self, sparse_feature_names: List[str], dense_feature_names: List[str], sparse_weight: float = 1.0,
                 dense_weight: float = 1.0, positive_only: bool = True, label_smooth: float = 0.0, device: str = 'cpu'):
        super().__init__()
        self.sparse_feature_names = sparse_feature_names
        self.dense_feature_names = dense_feature_names
        self.sparse_weight, self.dense_weight = sparse_weight, dense_weight
        self.positive_only = positive_only
        self.label_smooth = label_smooth
        self.device = device

    def forward(self, sparse_logits, dense_preds, sparse_targets, dense_targets, labels):
        if len(labels.shape) > 1:
            labels = labels.squeeze(-1)
        if self.positive_only:
            positive_mask = (labels == 1).float()
            num_positive = positive_mask.sum() + 1e-8
        else:
            positive_mask = torch.ones_like(labels).float()
            num_positive = labels.shape[0]
        loss_dict = {}
        total_sparse = torch.zeros((), device=labels.device)
        total_dense = torch.zeros((), device=labels.device)
        for name in self.sparse_feature_names:
            if name in sparse_logits and name in sparse_targets:
                targets = sparse_targets[name].long()
                if len(targets.shape) > 1:
                    targets = targets.squeeze(-1)
                ce = F.cross_entropy(sparse_logits[name], targets, reduction='none')
                masked = (ce * positive_mask).sum() / num_positive
                total_sparse = total_sparse + masked
                loss_dict['sfg_sparse_%s' % name] = masked
        if len(self.dense_feature_names) > 0 and dense_preds.shape[1] > 0:
            mse = F.mse_loss(dense_preds, dense_targets, reduction='none').mean(dim=-1)
            total_dense = (mse * positive_mask).sum() / num_positive
            loss_dict['sfg_dense'] = total_dense
        total = self.sparse_weight * total_sparse + self.dense_weight * total_dense
        loss_dict['sfg_total'] = total
        return total, loss_dict


# ------------------------------------------------------------------------------------------------- #
class AutoDisLayer(nn.Module):
    """Soft discretisation of dense features into bucket-embedding mixtures (deepctr/xdeepfm_pro/autodis.py:20-149)."""

    def __init__(self, num_features: int, num_buckets: int = 16, embedding_dim: int = 8, temperature: float = 1.0,
                 keep_raw: bool = True, device: str = 'cpu'):
        super().__init__()
        self.num_features, self.num_buckets, self.embedding_dim = num_features, num_buckets, embedding_dim
        self.temperature, self.keep_raw, self.device = temperature, keep_raw, device
        if num_features > 0:
            self.meta_embeddings = nn.Parameter(torch.randn(num_features, num_buckets, embedding_dim) * 0.01)
            self.bucket_projectors = nn.ModuleList([
                nn.Sequential(nn.Linear(1, num_buckets), nn.LeakyReLU(0.2), nn.Linear(num_buckets, num_buckets))
                for _ in range(num_features)])
            self.feature_temperatures = nn.Parameter(torch.ones(num_features) * temperature)
        self.to(device)

    def forward(self, dense_values: List[torch.Tensor]):
        if self.num_features == 0 or len(dense_values) == 0:
            batch_size = dense_values[0].shape[0] if dense_values else 1
            return torch.zeros(batch_size, 0, device=self.device), []
        batch_size = dense_values[0].shape[0]
        dense_embeddings = []
        for i, dense_val in enumerate(dense_values):
            if len(dense_val.shape) == 1:
                dense_val = dense_val.unsqueeze(-1)
            scores = self.bucket_projectors[i](dense_val)
            weights = F.softmax(scores / self.feature_temperatures[i], dim=-1)
            dense_embeddings.append(torch.matmul(weights, self.meta_embeddings[i]).unsqueeze(1))
        return torch.cat(dense_embeddings, dim=1).view(batch_size, -1), dense_embeddings

    def get_bucket_indices(self, dense_values: List[torch.Tensor]) -> List[torch.Tensor]:
        out = []
        for i, dense_val in enumerate(dense_values):
            if len(dense_val.shape) == 1:
                dense_val = dense_val.unsqueeze(-1)
            out.append(self.bucket_projectors[i](dense_val).argmax(dim=-1))
        return out


class DenseFeatureEncoder(nn.Module):
    """AutoDis or raw values behind one interface (deepctr/xdeepfm_pro/autodis.py:152-238)."""

    def __init__(self, dense_feature_names: List[str], embedding_dim: int = 8, use_autodis: bool = True,
                 num_buckets: int = 16, temperature: float = 1.0, device: str = 'cpu'):
        super().__init__()
        self.dense_feature_names = dense_feature_names
        self.embedding_dim, self.use_autodis = embedding_dim, use_autodis
        self.num_features = len(dense_feature_names)
        self.device = device
        self.autodis = AutoDisLayer(self.num_features, num_buckets, embedding_dim, temperature, device=device) \
            if use_autodis and self.num_features > 0 else None
        self.to(device)

    def forward(self, dense_values: List[torch.Tensor]):
        if self.num_features == 0 or len(dense_values) == 0:
            batch_size = dense_values[0].shape[0] if dense_values else 1
            z = torch.zeros(batch_size, 0, device=self.device)
            return z, [], z
        raw_values = torch.cat(dense_values, dim=-1)
        if self.use_autodis and self.autodis is not None:
            flat, emb_list = self.autodis(dense_values)
            return flat, emb_list, raw_values
        return raw_values, [dv.unsqueeze(-1) for dv in dense_values], raw_values

    def get_output_dim(self) -> int:
        return self.num_features * self.embedding_dim if self.use_autodis else self.num_features


# ------------------------------------------------------------------------------------------------- #
class BaseModelSFG(BaseModel):
    """BaseModel + SFG decoder / loss and the `sfg_loss` History entry (deepctr/xdeepfm_pro/basemodel_sfg.py:96-476).
    The training loop is BaseModel.fit: the step adds `sfg_weight * sfg_loss` to the total (basemodel_sfg.py:343) and
    the epoch logs gain `sfg_loss` = sum of the steps' values / sample_num (:365-366)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, l2_reg_linear=1e-5, l2_reg_embedding=1e-5,
                 init_std=0.0001, seed=1024, task='binary', device='cpu', gpus=None, use_sfg=True, sfg_weight=0.1,
                 sfg_hidden_units=(128, 64), sfg_dropout=0.1, sfg_positive_only=True, sfg_use_label_attention=True):
        super().__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                         l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task, device=device,
                         gpus=gpus)
        self.use_sfg = use_sfg
        self.sfg_weight = sfg_weight
        self.sfg_positive_only = sfg_positive_only
        self.sfg_loss = torch.zeros((1,), device=device)
        self.sparse_feature_columns = [fc for fc in dnn_feature_columns if isinstance(fc, SparseFeat)] \
            if dnn_feature_columns else []
        self.dense_feature_columns = [fc for fc in dnn_feature_columns if isinstance(fc, DenseFeat)] \
            if dnn_feature_columns else []
        self.embedding_dim = self.sparse_feature_columns[0].embedding_dim if self.sparse_feature_columns else 8
        if use_sfg:
            dense_names = [fc.name for fc in self.dense_feature_columns]
            self.sfg_decoder = SFGDecoder(
                embedding_dim=self.embedding_dim,
                sparse_feature_dims={fc.name: fc.vocabulary_size for fc in self.sparse_feature_columns},
                dense_feature_names=dense_names, hidden_units=sfg_hidden_units, dropout_rate=sfg_dropout,
                use_label_aware_attention=sfg_use_label_attention, device=device)
            self.sfg_loss_fn = SFGLoss([fc.name for fc in self.sparse_feature_columns], dense_names,
                                       positive_only=sfg_positive_only, device=device)
        else:
            self.sfg_decoder = None
            self.sfg_loss_fn = None
        self.to(device)

    def compile(self, optimizer, loss=None, metrics=None):
        from . import dist as xdist
        if xdist.current() is not None and self.use_sfg:
            raise NotImplementedError("xDeepFMPro under row-parallel training: the SFG loss divides by the number of positives "
                                      "of the GLOBAL batch (sfg_decoder.py:262-268), which the per-rank step does not exchange yet")
        super().compile(optimizer, loss, metrics)
        if self.use_sfg:                                   # basemodel_sfg.py:588-590
            self.metrics_names.insert(1, "sfg_loss")
        self._optim_capturable = False                     # eager launches: the positive-row compaction is data dependent

    def forward_with_sfg(self, X, y=None):
        raise NotImplementedError("Subclass must implement forward_with_sfg")

    def forward(self, X):
        return self.forward_with_sfg(X, None)[0]

    # ------------------------------------------------------------------ SFG loss on the fused layouts
    def compute_sfg_loss_fused(self, X, dnn_in, labels):
        """sfg loss from K1's `dnn_in` rows ([sparse embeddings | dense values] = the decoder input of
        sfg_decoder.py:113-136) -- arithmetic of compute_sfg_loss (basemodel_sfg.py:420-476) + SFGLoss, with the rows
        the mask zeroes left out and the vocabulary heads evaluated tile by tile."""
        dec, fn = self.sfg_decoder, self.sfg_loss_fn
        lab = labels.reshape(-1)
        if fn.positive_only:
            rows = torch.nonzero(lab == 1).reshape(-1)      # host-visible shape: this step is not graph-captured
            num_positive = rows.numel() + 1e-8
            x_rows, d_rows, l_rows = X.index_select(0, rows), dnn_in.index_select(0, rows), lab.index_select(0, rows)
        else:
            x_rows, d_rows, l_rows, num_positive = X, dnn_in, lab, float(lab.shape[0])
        loss_dict = {}
        total_sparse = torch.zeros((), device=X.device)
        total_dense = torch.zeros((), device=X.device)
        if x_rows.shape[0] > 0:
            hidden = dec.hidden(d_rows, l_rows)
            fcs = list(self.sparse_feature_columns)
            if fcs and ops.vocab_heads_ce_supported(hidden.shape[1]):
                # all heads in one autograd node: the hidden rows are packed once, no logits in HBM (ops.VocabHeadsCE)
                cols = [self.feature_index[fc.name][0] for fc in fcs]
                targets = x_rows[:, cols].long().t().contiguous()
                heads = [dec.sparse_heads[fc.name] for fc in fcs]
                ce_all = ops.vocab_heads_ce(hidden, targets, [h.weight for h in heads], [h.bias for h in heads])
                per_field = ce_all.sum(dim=1) / num_positive
                for k, fc in enumerate(fcs):
                    total_sparse = total_sparse + per_field[k]
                    loss_dict['sfg_sparse_%s' % fc.name] = per_field[k]
            else:
                for fc in fcs:
                    col = self.feature_index[fc.name][0]
                    head = dec.sparse_heads[fc.name]
                    ce = ops.vocab_softmax_ce(hidden, head.weight, head.bias, x_rows[:, col])
                    masked = ce.sum() / num_positive
                    total_sparse = total_sparse + masked
                    loss_dict['sfg_sparse_%s' % fc.name] = masked
            if dec.dense_head is not None:
                cols = [c for fc in self.dense_feature_columns for c in range(*self.feature_index[fc.name])]
                mse = F.mse_loss(dec.dense_head(hidden), x_rows[:, cols], reduction='none').mean(dim=-1)
                total_dense = mse.sum() / num_positive
                loss_dict['sfg_dense'] = total_dense
        total = fn.sparse_weight * total_sparse + fn.dense_weight * total_dense
        loss_dict['sfg_total'] = total
        return total, {'sfg_loss': total, 'sfg_loss_dict': loss_dict}

    def compute_sfg_loss(self, X, sparse_embedding_list, dense_value_list, labels):
        """Reference signature (basemodel_sfg.py:420-476): lists of [B,1,D] embeddings and [B,1] dense values."""
        if not self.use_sfg or self.sfg_decoder is None:
            return torch.tensor(0.0, device=X.device), {}
        sparse_concat = torch.cat([e.squeeze(1) if len(e.shape) == 3 else e for e in sparse_embedding_list], dim=-1)
        dense_concat = torch.cat(dense_value_list, dim=-1) if dense_value_list else \
            torch.zeros(X.shape[0], 0, device=X.device)
        return self.compute_sfg_loss_fused(X, torch.cat([sparse_concat, dense_concat], dim=-1), labels)

    # ------------------------------------------------------------------ hooks of BaseModel's train step
    def _loss_forward(self, x, y):
        y_pred, sfg_info = self.forward_with_sfg(x, y)
        y_pred = y_pred.squeeze()
        loss = self.loss_func(y_pred, y.squeeze(), reduction='sum')
        self._step_extra = None
        # the reference logs sfg_loss every epoch when use_sfg, also when it is zero (an epoch after the first
        # validation runs in eval mode there -- fit never calls train() again, basemodel_sfg.py:276 / :493 -- so its
        # forward_with_sfg returns no sfg_info; same here, BaseModel.fit shares that behaviour)
        self._step_log = ("sfg_loss", torch.zeros((), device=loss.device)) if self.use_sfg else None
        if self.use_sfg and sfg_info is not None:
            self._step_extra = self.sfg_weight * sfg_info['sfg_loss']
            self._step_log = ("sfg_loss", sfg_info['sfg_loss'].detach())
        return y_pred, loss


class xDeepFMPro(BaseModelSFG):
    """xDeepFM + SFG (+ AutoDis) (deepctr/xdeepfm_pro/xdeepfm_pro.py:31-393)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 256), cin_layer_size=(256, 128),
                 cin_split_half=True, cin_activation='relu', l2_reg_linear=0.00001, l2_reg_embedding=0.00001, l2_reg_dnn=0,
                 l2_reg_cin=0, init_std=0.0001, seed=1024, dnn_dropout=0, dnn_activation='relu', dnn_use_bn=False,
                 task='binary', device='cpu', gpus=None, use_sfg=True, sfg_weight=0.1, sfg_hidden_units=(128, 64),
                 sfg_dropout=0.1, sfg_positive_only=True, sfg_use_label_attention=True, use_autodis=False,
                 autodis_buckets=16, autodis_temperature=1.0):
        super().__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=l2_reg_linear,
                         l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task, device=device,
                         gpus=gpus, use_sfg=use_sfg, sfg_weight=sfg_weight, sfg_hidden_units=sfg_hidden_units,
                         sfg_dropout=sfg_dropout, sfg_positive_only=sfg_positive_only,
                         sfg_use_label_attention=sfg_use_label_attention)
        self.dnn_hidden_units = dnn_hidden_units
        self.use_dnn = len(dnn_feature_columns) > 0 and len(dnn_hidden_units) > 0
        self.use_autodis = use_autodis
        if use_autodis and len(self.dense_feature_columns) > 0:
            self.autodis_encoder = DenseFeatureEncoder([fc.name for fc in self.dense_feature_columns],
                                                       embedding_dim=self.embedding_dim, use_autodis=True,
                                                       num_buckets=autodis_buckets, temperature=autodis_temperature,
                                                       device=device)
            autodis_output_dim = self.autodis_encoder.get_output_dim()
        else:
            self.autodis_encoder = None
            autodis_output_dim = 0
        if self.use_dnn:
            dnn_input_dim = self.compute_input_dim(dnn_feature_columns)
            if use_autodis and self.autodis_encoder is not None:
                dnn_input_dim += autodis_output_dim - sum(fc.dimension for fc in self.dense_feature_columns)
            self.dnn = DNN(dnn_input_dim, dnn_hidden_units, activation=dnn_activation, l2_reg=l2_reg_dnn,
                           dropout_rate=dnn_dropout, use_bn=dnn_use_bn, init_std=init_std, device=device)
            self.dnn_linear = nn.Linear(dnn_hidden_units[-1], 1, bias=False).to(device)
            self.add_regularization_weight(
                filter(lambda x: 'weight' in x[0] and 'bn' not in x[0], self.dnn.named_parameters()), l2=l2_reg_dnn)
            self.add_regularization_weight(self.dnn_linear.weight, l2=l2_reg_dnn)
        self.cin_layer_size = cin_layer_size
        self.use_cin = len(cin_layer_size) > 0 and len(dnn_feature_columns) > 0
        if self.use_cin:
            self.featuremap_num = (sum(cin_layer_size[:-1]) // 2 + cin_layer_size[-1]) if cin_split_half \
                else sum(cin_layer_size)
            self.cin = CIN(len(self.embedding_dict), cin_layer_size, cin_activation, cin_split_half, l2_reg_cin, seed,
                           device=device)
            self.cin_linear = nn.Linear(self.featuremap_num, 1, bias=False).to(device)
            self.add_regularization_weight(filter(lambda x: 'weight' in x[0], self.cin.named_parameters()), l2=l2_reg_cin)
        self.to(device)

    def forward_with_sfg(self, X, y=None):
        """(y_pred, sfg_info) (xdeepfm_pro.py:203-274) on the fused layouts: one gather launch feeds the CIN (FM
        layout), the DNN, the linear logit and the SFG decoder."""
        emb_fm, dnn_in, logit = self.fused_inputs(X)
        B = X.shape[0]
        plan = self._plan
        if self.use_cin:
            logit = logit + self.cin_linear(self.cin.forward_fm(emb_fm, B, plan.D))
        if self.use_dnn:
            dnn_input = dnn_in
            if self.use_autodis and self.autodis_encoder is not None and plan.nd > 0:
                mD = plan.m * plan.D
                dense_list = [dnn_in[:, mD + k:mD + k + 1] for k in range(plan.nd)]
                autodis_out, _, _ = self.autodis_encoder(dense_list)
                dnn_input = torch.cat([dnn_in[:, :mD], autodis_out], dim=-1)
            logit = logit + self.dnn_linear(self.dnn(dnn_input))
        y_pred = self.out(logit)
        sfg_info = None
        if self.use_sfg and y is not None and self.training:
            sfg_loss, sfg_info = self.compute_sfg_loss_fused(X, dnn_in, y)
            sfg_info['sfg_loss'] = sfg_loss
        return y_pred, sfg_info

    def get_embedding_analysis(self, X):
        """Embedding statistics (xdeepfm_pro.py:281-324)."""
        with torch.no_grad():
            emb_fm, _, _ = self.fused_inputs(X)
            B = X.shape[0]
            all_emb = ops.from_fm_layout(emb_fm, B, self._plan.D)
            flat = all_emb.reshape(B, -1)
            normalized = flat / (flat.norm(dim=1, keepdim=True) + 1e-8)
            cos = torch.mm(normalized, normalized.t())
            return {'mean_embedding': all_emb.mean(dim=0), 'std_embedding': all_emb.std(dim=0),
                    'embedding_variance': all_emb.var(dim=0).mean(),
                    'avg_sample_cosine_similarity': (cos.sum() - cos.trace()) / (cos.numel() - cos.shape[0]),
                    'num_fields': all_emb.shape[1], 'embedding_dim': all_emb.shape[2]}


class xDeepFMProLight(xDeepFMPro):
    """Smaller defaults (deepctr/xdeepfm_pro/xdeepfm_pro.py:327-393)."""

    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(128, 64), cin_layer_size=(128, 64),
                 cin_split_half=True, cin_activation='relu', l2_reg_linear=0.00001, l2_reg_embedding=0.00001, l2_reg_dnn=0,
                 l2_reg_cin=0, init_std=0.0001, seed=1024, dnn_dropout=0, dnn_activation='relu', dnn_use_bn=False,
                 task='binary', device='cpu', gpus=None, use_sfg=True, sfg_weight=0.05, sfg_hidden_units=(64, 32),
                 sfg_dropout=0.1, sfg_positive_only=True, sfg_use_label_attention=True, use_autodis=False,
                 autodis_buckets=8, autodis_temperature=1.0):
        super().__init__(linear_feature_columns, dnn_feature_columns, dnn_hidden_units=dnn_hidden_units,
                         cin_layer_size=cin_layer_size, cin_split_half=cin_split_half, cin_activation=cin_activation,
                         l2_reg_linear=l2_reg_linear, l2_reg_embedding=l2_reg_embedding, l2_reg_dnn=l2_reg_dnn,
                         l2_reg_cin=l2_reg_cin, init_std=init_std, seed=seed, dnn_dropout=dnn_dropout,
                         dnn_activation=dnn_activation, dnn_use_bn=dnn_use_bn, task=task, device=device, gpus=gpus,
                         use_sfg=use_sfg, sfg_weight=sfg_weight, sfg_hidden_units=sfg_hidden_units, sfg_dropout=sfg_dropout,
                         sfg_positive_only=sfg_positive_only, sfg_use_label_attention=sfg_use_label_attention,
                         use_autodis=use_autodis, autodis_buckets=autodis_buckets, autodis_temperature=autodis_temperature)
