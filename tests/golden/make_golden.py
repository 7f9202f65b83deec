#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only); the .npz
files it writes are data (inputs, weights, expected outputs / gradients) and are
what travels to the GPU box.  Usage:  python tests/golden/make_golden.py

Import recipe:
  * `deepctr` is pre-registered as an empty namespace package whose __path__ is
    the reference directory, so deepctr/__init__.py (which spawns a version-check
    thread that fetches a URL, deepctr/utils.py:19-44) is never executed;
  * the layer / input modules are pure torch and import unmodified;
  * deepctr/models/basemodel.py:22-25 and deepctr/callbacks.py:2-4 import four
    Keras *plumbing* classes (CallbackList, History, EarlyStopping,
    ModelCheckpoint) from tensorflow, which is not installed.  They carry no
    arithmetic; minimal stand-ins are registered so the model classes import.
"""
import importlib.machinery
import os
import sys
import types
import zlib

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _register_reference():
    pkg = types.ModuleType("deepctr")
    pkg.__path__ = [os.path.join(REF, "deepctr")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("deepctr", None, is_package=True)
    sys.modules["deepctr"] = pkg

    class _Callback(object):
        def __init__(self, *a, **k):
            self.model = None

        def set_model(self, model):
            self.model = model

        def set_params(self, params):
            self.params = params

        def on_train_begin(self, logs=None): pass
        def on_train_end(self, logs=None): pass
        def on_epoch_begin(self, epoch, logs=None): pass
        def on_epoch_end(self, epoch, logs=None): pass

    class History(_Callback):
        def on_train_begin(self, logs=None):
            self.epoch, self.history = [], {}

        def on_epoch_end(self, epoch, logs=None):
            self.epoch.append(epoch)
            for k, v in (logs or {}).items():
                self.history.setdefault(k, []).append(v)

    class CallbackList(object):
        def __init__(self, callbacks=None):
            self.callbacks = list(callbacks or [])

        def set_model(self, model):
            self.model = model
            for c in self.callbacks:
                c.set_model(model)

        def __getattr__(self, name):
            if name.startswith("on_"):
                def _fan(*a, **k):
                    for c in self.callbacks:
                        getattr(c, name)(*a, **k)
                return _fan
            raise AttributeError(name)

    class EarlyStopping(_Callback): pass
    class ModelCheckpoint(_Callback): pass

    for name in ("tensorflow", "tensorflow.python", "tensorflow.python.keras",
                 "tensorflow.python.keras.callbacks"):
        mod = types.ModuleType(name)
        mod.__path__ = []
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = mod
    cb = sys.modules["tensorflow.python.keras.callbacks"]
    cb.CallbackList, cb.History = CallbackList, History
    cb.EarlyStopping, cb.ModelCheckpoint = EarlyStopping, ModelCheckpoint


_register_reference()
from deepctr.layers.interaction import CIN                      # noqa: E402
from deepctr.layers.cin_attention import (CINAttention, CINAttentionV2,   # noqa: E402
                                          MultiHeadSelfAttention, AttentionPooling)
from deepctr.inputs import SparseFeat, DenseFeat, get_feature_names, build_input_features  # noqa: E402
from deepctr.models.xdeepfm import xDeepFM                      # noqa: E402
from deepctr.models.xdeepfm_attn import xDeepFMAttention, xDeepFMAttentionV2   # noqa: E402

from oracle import xdeepfm_oracle as orc                        # noqa: E402


def _np(t):
    return t.detach().cpu().numpy().copy()


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-34s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


# --------------------------------------------------------------------------- #
CIN_CASES = [
    # name,            B,  m,  D, layer_size,     split, act
    ("cin_b1_m3",       1,  3,  4, (4,),           True,  "relu"),
    ("cin_b3_m3_l2",    3,  3,  4, (8, 4),         True,  "relu"),
    ("cin_b3_m5_l3",    3,  5,  8, (6, 4, 4),      True,  "relu"),
    ("cin_nosplit",     3,  4,  8, (6, 5),         False, "relu"),
    ("cin_linear_act",  5,  4,  4, (8, 6),         True,  "linear"),
    ("cin_odd_last",    4,  3, 16, (8, 5),         True,  "relu"),
    ("cin_b64_m26_d16", 64, 26, 16, (32, 16, 16),  True,  "relu"),
    ("cin_b9_m22_d10",  9, 22, 10, (16, 12),       True,  "relu"),
    # layer sizes the DEFAULT arithmetic (f16x3) has kernels for in all three directions (forward H > 32, dW H > 64, dX
    # H > 16; level 0 of both runs the folded kernels): the goldens above all fall back to the fp32-MFMA kernels there
    ("cin_x3_m26_d16",  8, 26, 16, (128, 96, 72),  True,  "relu"),
    ("cin_x3_m22_d32",  6, 22, 32, (136, 96),      True,  "relu"),
]


def gen_cin():
    for name, B, m, D, ls, split, act in CIN_CASES:
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2 ** 31))
        torch.manual_seed(7)
        layer = CIN(m, ls, act, split, 0.0, 1024, device="cpu")
        x = (torch.randn(B, m, D, generator=g) * 0.5).requires_grad_(True)
        # biases of default init are O(1/sqrt(K)); keep them, they dominate at init
        out = layer(x)
        gout = torch.randn(out.shape, generator=g)
        (out * gout).sum().backward()
        arrays = dict(x=_np(x), out=_np(out), gout=_np(gout), dx=_np(x.grad),
                      layer_size=np.array(ls), split_half=np.array(split), activation=np.array(act))
        for i, c in enumerate(layer.conv1ds):
            arrays["w%d" % i], arrays["b%d" % i] = _np(c.weight), _np(c.bias)
            arrays["dw%d" % i], arrays["db%d" % i] = _np(c.weight.grad), _np(c.bias.grad)
        # cross-check the oracle right here against the live reference
        W = [c.weight.detach() for c in layer.conv1ds]
        Bs = [c.bias.detach() for c in layer.conv1ds]
        o = orc.cin_forward(x.detach(), W, Bs, split, act)
        assert torch.allclose(o, out.detach(), rtol=1e-6, atol=1e-6), name
        _save(name, **arrays)


ATTN_CASES = [
    # name,             cls,  B,  m,  D, layer_size, heads, ln,   res,  nlayers
    ("attn_v1_small",   "v1", 3,  4,  8, (8, 6),     4,     True,  True,  1),
    ("attn_v1_noln",    "v1", 2,  3,  8, (6, 4),     2,     False, True,  1),
    ("attn_v1_nores",   "v1", 2,  3,  4, (6, 4),     4,     True,  False, 1),
    ("attn_v1_heads3",  "v1", 2,  3, 10, (6, 4),     4,     True,  True,  1),   # 10 % 4 != 0 -> 2 heads
    ("attn_v1_d16",     "v1", 4, 26, 16, (16, 8),    4,     True,  True,  1),
    ("attn_v2_l1",      "v2", 3,  4,  8, (8, 6),     4,     True,  True,  1),
    ("attn_v2_l2",      "v2", 3,  4,  8, (8, 6),     2,     True,  True,  2),
]


def gen_attn():
    for name, cls, B, m, D, ls, heads, ln, res, nl in ATTN_CASES:
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2 ** 31))
        torch.manual_seed(11)
        if cls == "v1":
            layer = CINAttention(m, D, ls, "relu", True, heads, 0.0, ln, res, 0.0, 1024, "cpu")
            variant = "attn"
        else:
            layer = CINAttentionV2(m, D, ls, "relu", True, heads, 0.0, ln, res, nl, 0.0, 1024, "cpu")
            variant = "attn_v2"
        # make LayerNorm affine / pooling bias non-trivial so they are actually exercised
        with torch.no_grad():
            for k, p in layer.named_parameters():
                if "layer_norm" in k or k.endswith("attention.0.bias"):
                    p.add_(0.3 * torch.randn(p.shape, generator=g))
        x = (torch.randn(B, m, D, generator=g) * 0.7).requires_grad_(True)
        out = layer(x)
        gout = torch.randn(out.shape, generator=g)
        (out * gout).sum().backward()
        arrays = dict(x=_np(x), out=_np(out), gout=_np(gout), dx=_np(x.grad), layer_size=np.array(ls),
                      num_heads=np.array(heads), use_layer_norm=np.array(ln), use_residual=np.array(res),
                      num_attn_layers=np.array(nl), variant=np.array(variant))
        for k, p in layer.named_parameters():
            arrays["p:" + k] = _np(p)
            arrays["g:" + k] = _np(p.grad)
        spec = orc.Spec(["f%d" % i for i in range(m)], [1] * m, [], D, tuple(ls), True, "relu", (),
                        variant, heads, ln, res, nl)
        st = {"cin." + k: p.detach() for k, p in layer.named_parameters()}
        o = orc.cin_attention_forward(x.detach(), st, "cin.", spec)
        assert torch.allclose(o, out.detach(), rtol=1e-5, atol=1e-6), name
        _save(name, **arrays)


# --------------------------------------------------------------------------- #
def _columns(vocab, n_dense, D):
    sparse = ["C%d" % (i + 1) for i in range(len(vocab))]
    dense = ["I%d" % (i + 1) for i in range(n_dense)]
    cols = [SparseFeat(n, vocabulary_size=v, embedding_dim=D) for n, v in zip(sparse, vocab)]
    cols += [DenseFeat(n, 1) for n in dense]
    return sparse, dense, cols


MODEL_CASES = [
    # name,              cls,               vocab list,                   nd,  D, cin,        dnn,      B,  kw
    ("model_sum_small",  xDeepFM,           [7, 5, 11, 3, 9, 4],           3,  4, (8, 6),     (16, 8),  32, {}),
    ("model_sum_c1",     xDeepFM,           [50] * 26,                    13,  8, (32, 16),   (32, 32), 64, {}),
    ("model_attn_small", xDeepFMAttention,  [7, 5, 11, 3, 9, 4],           3,  8, (8, 6),     (16, 8),  16,
     dict(cin_num_heads=4)),
    ("model_attnv2_small", xDeepFMAttentionV2, [7, 5, 11, 3, 9, 4],        3,  8, (8, 6),     (16, 8),  16,
     dict(cin_num_heads=2, cin_num_attn_layers=2)),
    ("model_nodense",    xDeepFM,           [13] * 22,                     0, 16, (16, 8, 8), (16,),    24, {}),
    # CIN levels wide enough for the f16x3 forward / dX / dW kernels (the bench's arithmetic)
    ("model_x3_cin",     xDeepFM,           [30] * 26,                    13,  8, (96, 72),   (32, 32), 32, {}),
]


def gen_models():
    for name, cls, vocab, nd, D, cin, dnn, B, kw in MODEL_CASES:
        sparse, dense, cols = _columns(vocab, nd, D)
        model = cls(cols, cols, dnn_hidden_units=dnn, cin_layer_size=cin, l2_reg_dnn=1e-5,
                    device="cpu", **kw)
        assert list(model.feature_index.keys()) == get_feature_names(cols) == sparse + dense
        init = {k: _np(v) for k, v in model.state_dict().items()}
        # embeddings at N(0,1e-4) make every gradient tiny; goldens use a livelier copy of the weights
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if "embedding_dict" in k or k == "linear_model.weight" or "dnn" in k or k == "cin_linear.weight":
                    p.copy_(0.3 * torch.randn(p.shape, generator=g))
        X, y = orc.synthetic_batch(3 * B, vocab, nd, seed=2025)
        model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
        state0 = {k: _np(v) for k, v in model.state_dict().items()}
        Xt, yt = torch.from_numpy(X[:B]), torch.from_numpy(y[:B])
        model.train()
        y_pred = model(Xt)
        loss = torch.nn.functional.binary_cross_entropy(y_pred.squeeze(), yt.squeeze(), reduction="sum")
        reg = model.get_regularization_loss()
        model.optim.zero_grad()
        (loss + reg).backward()
        grads = {k: _np(p.grad) for k, p in model.named_parameters()}
        model.optim.zero_grad()
        # three Adam steps exactly as BaseModel.fit does them (basemodel.py:241-262)
        losses = []
        for s in range(3):
            xb = torch.from_numpy(X[s * B:(s + 1) * B]).float()
            yb = torch.from_numpy(y[s * B:(s + 1) * B]).float()
            yp = model(xb).squeeze()
            model.optim.zero_grad()
            l = torch.nn.functional.binary_cross_entropy(yp, yb.squeeze(), reduction="sum")
            tot = l + model.get_regularization_loss() + model.aux_loss
            losses.append([l.item(), tot.item()])
            tot.backward()
            model.optim.step()
        state3 = {k: _np(v) for k, v in model.state_dict().items()}
        pred_after = model.predict([X[:, i] for i in range(X.shape[1])], batch_size=B)
        arrays = dict(X=X, y=y, B=np.array(B), y_pred=_np(y_pred), loss=np.array(loss.item()),
                      reg=np.array(reg.item()), losses3=np.array(losses), pred_after=pred_after,
                      vocab=np.array(vocab), n_dense=np.array(nd), emb_dim=np.array(D),
                      cin=np.array(cin), dnn=np.array(dnn), cls=np.array(cls.__name__),
                      kw_keys=np.array(sorted(kw.keys())), kw_vals=np.array([kw[k] for k in sorted(kw)]))
        for k, v in init.items():
            arrays["init:" + k] = v if ("embedding_dict" not in k or len(vocab) < 10) else v[:2]
        for k, v in state0.items():
            arrays["s0:" + k] = v
        for k, v in grads.items():
            arrays["g:" + k] = v
        for k, v in state3.items():
            arrays["s3:" + k] = v
        _save(name, **arrays)


def gen_fit_history():
    """2-epoch BaseModel.fit with shuffle=False (basemodel.py:137-309): History keys and values."""
    vocab, nd, D = [9, 6, 12, 5], 2, 4
    sparse, dense, cols = _columns(vocab, nd, D)
    model = xDeepFM(cols, cols, dnn_hidden_units=(8,), cin_layer_size=(6, 4), l2_reg_dnn=1e-5, device="cpu")
    model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
    for pg in model.optim.param_groups:
        pg["lr"] = 1e-2
    s0 = {k: _np(v) for k, v in model.state_dict().items()}
    X, y = orc.synthetic_batch(200, vocab, nd, seed=77)
    Xv, yv = orc.synthetic_batch(80, vocab, nd, seed=78)
    names = sparse + dense
    hist = model.fit({n: X[:, i] for i, n in enumerate(names)}, y, batch_size=64, epochs=2, verbose=2,
                     validation_data=({n: Xv[:, i] for i, n in enumerate(names)}, yv), shuffle=False)
    arrays = dict(X=X, y=y, Xv=Xv, yv=yv, vocab=np.array(vocab), n_dense=np.array(nd), emb_dim=np.array(D),
                  hist_keys=np.array(sorted(hist.history.keys())),
                  hist_vals=np.array([hist.history[k] for k in sorted(hist.history.keys())]),
                  pred=model.predict({n: Xv[:, i] for i, n in enumerate(names)}, 32))
    for k, v in s0.items():
        arrays["s0:" + k] = v
    for k, v in model.state_dict().items():
        arrays["s1:" + k] = _np(v)
    _save("fit_history", **arrays)


PRO_CASES = [
    # name,          vocab,                nd, D, cin,    dnn,     sfg hidden, B,  kw
    ("pro_small",    [7, 5, 11, 3, 9, 4],  3,  4, (8, 6), (16, 8), (16, 8),    48, dict(sfg_positive_only=True, sfg_use_label_attention=True)),
    ("pro_autodis",  [7, 5, 11, 3, 9, 4],  3,  4, (8, 6), (16, 8), (16, 8),    48, dict(sfg_positive_only=False, sfg_use_label_attention=False,
                                                                                       use_autodis=True, autodis_buckets=6, sfg_weight=0.3)),
    ("pro_nodense",  [13] * 22,            0,  8, (16, 8), (16,),  (12,),      32, dict()),
    # heads of width 64 over vocabularies on both sides of the 256-row stages: the fused heads (csrc/vocab_ce_x3.hip) run
    ("pro_heads64",  [300, 57, 1000, 33, 260, 5], 3, 4, (8, 6), (16, 8), (32, 64), 96, dict()),
]


def gen_pro(only_case=None):
    """deepctr/xdeepfm_pro: xDeepFMPro forward_with_sfg, every gradient of loss + reg + sfg_weight * sfg_loss, three Adam
    steps of the BaseModelSFG.fit loop body (basemodel_sfg.py:317-349), predict; sfg_dropout = 0 (the dropout masks of two
    generators cannot agree)."""
    from deepctr.xdeepfm_pro.xdeepfm_pro import xDeepFMPro
    for name, vocab, nd, D, cin, dnn, sfgh, B, kw in PRO_CASES:
        if only_case and name != only_case:
            continue
        sparse, dense, cols = _columns(vocab, nd, D)
        model = xDeepFMPro(cols, cols, dnn_hidden_units=dnn, cin_layer_size=cin, l2_reg_dnn=1e-5, device="cpu",
                           sfg_hidden_units=sfgh, sfg_dropout=0.0, **kw)
        init = {k: _np(v) for k, v in model.state_dict().items()}
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if "embedding_dict" in k or k == "linear_model.weight" or k.startswith("dnn") or k == "cin_linear.weight" \
                        or "sfg_decoder" in k or "autodis" in k:
                    if "feature_temperatures" in k:
                        p.copy_(1.0 + 0.2 * torch.rand(p.shape, generator=g))
                    else:
                        p.copy_(0.3 * torch.randn(p.shape, generator=g))
        X, y = orc.synthetic_batch(3 * B, vocab, nd, seed=2026)
        model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
        state0 = {k: _np(v) for k, v in model.state_dict().items()}
        model.train()
        Xt, yt = torch.from_numpy(X[:B]).float(), torch.from_numpy(y[:B]).float()
        y_pred, info = model.forward_with_sfg(Xt, yt)
        loss = torch.nn.functional.binary_cross_entropy(y_pred.squeeze(), yt.squeeze(), reduction="sum")
        reg = model.get_regularization_loss()
        sfg = info["sfg_loss"]
        model.optim.zero_grad()
        (loss + reg + model.aux_loss + model.sfg_weight * sfg).backward()
        grads = {k: _np(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)
                 for k, p in model.named_parameters()}
        model.optim.zero_grad()
        losses = []
        for s in range(3):
            xb = torch.from_numpy(X[s * B:(s + 1) * B]).float()
            yb = torch.from_numpy(y[s * B:(s + 1) * B]).float()
            yp, inf = model.forward_with_sfg(xb, yb)
            model.optim.zero_grad()
            l = torch.nn.functional.binary_cross_entropy(yp.squeeze(), yb.squeeze(), reduction="sum")
            tot = l + model.get_regularization_loss() + model.aux_loss + model.sfg_weight * inf["sfg_loss"]
            losses.append([l.item(), tot.item(), inf["sfg_loss"].item()])
            tot.backward()
            model.optim.step()
        state3 = {k: _np(v) for k, v in model.state_dict().items()}
        pred_after = model.predict([X[:, i] for i in range(X.shape[1])], batch_size=B)
        arrays = dict(X=X, y=y, B=np.array(B), y_pred=_np(y_pred), loss=np.array(loss.item()), reg=np.array(reg.item()),
                      sfg=np.array(sfg.item()), losses3=np.array(losses), pred_after=pred_after, vocab=np.array(vocab),
                      n_dense=np.array(nd), emb_dim=np.array(D), cin=np.array(cin), dnn=np.array(dnn), sfg_hidden=np.array(sfgh),
                      kw_keys=np.array(sorted(kw.keys())), kw_vals=np.array([float(kw[k]) for k in sorted(kw)]))
        for k, v in init.items():
            arrays["init:" + k] = v
        for k, v in state0.items():
            arrays["s0:" + k] = v
        for k, v in grads.items():
            arrays["g:" + k] = v
        for k, v in state3.items():
            arrays["s3:" + k] = v
        _save(name, **arrays)
    if only_case:
        return
    # History of BaseModelSFG.fit (basemodel_sfg.py:224-400): keys incl. sfg_loss
    vocab, nd, D = [9, 6, 12, 5], 2, 4
    sparse, dense, cols = _columns(vocab, nd, D)
    model = xDeepFMPro(cols, cols, dnn_hidden_units=(8,), cin_layer_size=(6, 4), l2_reg_dnn=1e-5, device="cpu",
                       sfg_hidden_units=(8, 6), sfg_dropout=0.0, sfg_weight=0.2)
    model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
    for pg in model.optim.param_groups:
        pg["lr"] = 1e-2
    s0 = {k: _np(v) for k, v in model.state_dict().items()}
    X, y = orc.synthetic_batch(256, vocab, nd, seed=79)      # full batches: a one-class tail batch makes sklearn raise in the reference loop
    Xv, yv = orc.synthetic_batch(80, vocab, nd, seed=80)
    names = sparse + dense
    hist = model.fit({n: X[:, i] for i, n in enumerate(names)}, y, batch_size=64, epochs=2, verbose=2,
                     validation_data=({n: Xv[:, i] for i, n in enumerate(names)}, yv), shuffle=False)
    arrays = dict(X=X, y=y, Xv=Xv, yv=yv, vocab=np.array(vocab), n_dense=np.array(nd), emb_dim=np.array(D),
                  hist_keys=np.array(sorted(hist.history.keys())),
                  hist_vals=np.array([hist.history[k] for k in sorted(hist.history.keys())]),
                  pred=model.predict({n: Xv[:, i] for i, n in enumerate(names)}, 32))
    for k, v in s0.items():
        arrays["s0:" + k] = v
    for k, v in model.state_dict().items():
        arrays["s1:" + k] = _np(v)
    _save("pro_fit_history", **arrays)


def gen_metrics():
    from sklearn.metrics import log_loss, roc_auc_score
    rng = np.random.default_rng(3)
    y = (rng.random(500) < 0.3).astype(np.float32)
    p = np.clip(rng.random(500), 1e-4, 1 - 1e-4)
    p[::7] = p[3]                                   # ties
    _save("metrics", y=y, p=p, logloss=np.array(log_loss(y, p)), auc=np.array(roc_auc_score(y, p)))


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = sys.argv[1:]                  # e.g. `make_golden.py pro` regenerates one family, `pro:pro_heads64` one case of it
    for a in [a for a in only if a.startswith("pro:")]:
        gen_pro(a.split(":", 1)[1])
        only = [o for o in only if o != a] or ["-"]
    for fam, fn in (("cin", gen_cin), ("attn", gen_attn), ("models", gen_models), ("fit", gen_fit_history),
                    ("pro", gen_pro), ("metrics", gen_metrics)):
        if not only or fam in only:
            fn()
