"""Pins oracle/xdeepfm_oracle.py to the reference: every golden file under tests/golden/
was produced by running the real reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import xdeepfm_oracle as orc

T = torch.from_numpy


def _spec_from_model_golden(g):
    vocab = [int(v) for v in g["vocab"]]
    nd = int(g["n_dense"])
    cls = str(g["cls"])
    kw = dict(zip([str(k) for k in g["kw_keys"]], [int(v) for v in g["kw_vals"]]))
    variant = {"xDeepFM": "sum", "xDeepFMAttention": "attn", "xDeepFMAttentionV2": "attn_v2"}[cls]
    return orc.Spec(["C%d" % (i + 1) for i in range(len(vocab))], vocab,
                    ["I%d" % (i + 1) for i in range(nd)], int(g["emb_dim"]),
                    tuple(int(v) for v in g["cin"]), True, "relu", tuple(int(v) for v in g["dnn"]),
                    variant, kw.get("cin_num_heads", 4), True, True, kw.get("cin_num_attn_layers", 1),
                    l2_reg_dnn=1e-5)


@pytest.mark.parametrize("name", golden_names("cin_"))
def test_cin_forward_backward(name):
    g = load_golden(name)
    L = len(g["layer_size"])
    x = T(g["x"]).requires_grad_(True)
    W = [T(g["w%d" % i]).requires_grad_(True) for i in range(L)]
    Bs = [T(g["b%d" % i]).requires_grad_(True) for i in range(L)]
    out = orc.cin_forward(x, W, Bs, bool(g["split_half"]), str(g["activation"]))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-6, atol=1e-6)
    (out * T(g["gout"])).sum().backward()
    # gradients are long fp32 reductions whose order depends on the torch thread count
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-4, atol=1e-5)
    for i in range(L):
        np.testing.assert_allclose(W[i].grad.numpy(), g["dw%d" % i], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(Bs[i].grad.numpy(), g["db%d" % i], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", golden_names("attn_"))
def test_cin_attention_forward_backward(name):
    g = load_golden(name)
    x = T(g["x"]).requires_grad_(True)
    B, m, D = x.shape
    st = {"cin." + k[2:]: T(v).requires_grad_(True) for k, v in g.items() if k.startswith("p:")}
    spec = orc.Spec(["f%d" % i for i in range(m)], [1] * m, [], D, tuple(int(v) for v in g["layer_size"]),
                    True, "relu", (), str(g["variant"]), int(g["num_heads"]), bool(g["use_layer_norm"]),
                    bool(g["use_residual"]), int(g["num_attn_layers"]))
    out = orc.cin_attention_forward(x, st, "cin.", spec)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-5, atol=1e-6)
    (out * T(g["gout"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-4, atol=1e-6)
    for k, v in st.items():
        np.testing.assert_allclose(v.grad.numpy(), g["g:" + k[4:]], rtol=1e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("name", golden_names("model_"))
def test_model_forward_grads_adam(name):
    g = load_golden(name)
    spec = _spec_from_model_golden(g)
    B = int(g["B"])
    X, y = T(g["X"]), T(g["y"])
    st = {k[3:]: T(v.copy()).requires_grad_(True) for k, v in g.items() if k.startswith("s0:")}
    tot, dl, yp = orc.total_loss(X[:B], y[:B], st, spec)
    np.testing.assert_allclose(yp.detach().numpy(), g["y_pred"].squeeze(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dl.item(), float(g["loss"]), rtol=1e-6)
    np.testing.assert_allclose(orc.regularization_loss(st, spec).item(), float(g["reg"]), rtol=1e-5)
    tot.backward()
    for k, v in st.items():
        np.testing.assert_allclose(v.grad.numpy(), g["g:" + k], rtol=2e-4, atol=1e-5 * float(np.abs(g["g:" + k]).max()) + 1e-9, err_msg=k)
    # three Adam steps (deepctr/models/basemodel.py:241-262)
    st = {k[3:]: T(v.copy()) for k, v in g.items() if k.startswith("s0:")}
    batches = [(X[s * B:(s + 1) * B], y[s * B:(s + 1) * B]) for s in range(3)]
    log = orc.train_steps(batches, st, spec, lr=1e-3)
    np.testing.assert_allclose(np.array(log), g["losses3"], rtol=1e-5)
    for k, v in st.items():
        np.testing.assert_allclose(v.detach().numpy(), g["s3:" + k], rtol=1e-4, atol=1e-6, err_msg=k)
    with torch.no_grad():
        pred = orc.model_forward(X, {k: v.detach() for k, v in st.items()}, spec)
    np.testing.assert_allclose(pred.numpy(), g["pred_after"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", golden_names("model_"))
def test_init_matches_reference_rng_order(name):
    """seed 1024 -> the reference's own initial weights."""
    g = load_golden(name)
    spec = _spec_from_model_golden(g)
    st = orc.init_state(spec, seed=1024, init_std=1e-4)
    keys = [k[5:] for k in g if k.startswith("init:")]
    assert sorted(keys) == sorted(st.keys())
    for k in keys:
        want = g["init:" + k]
        got = st[k].numpy()[: want.shape[0]] if want.shape != tuple(st[k].shape) else st[k].numpy()
        np.testing.assert_array_equal(got, want, err_msg=k)


def test_metrics_match_sklearn_golden():
    g = load_golden("metrics")
    assert abs(orc.log_loss(g["y"], g["p"]) - float(g["logloss"])) < 1e-12
    assert abs(orc.roc_auc(g["y"], g["p"]) - float(g["auc"])) < 1e-12


def test_cin_rejects_non_3d():
    with pytest.raises(ValueError):
        orc.cin_forward(torch.zeros(3, 4), [torch.zeros(2, 16, 1)], [torch.zeros(2)])


def _pro_case(g):
    vocab, nd, D = [int(v) for v in g["vocab"]], int(g["n_dense"]), int(g["emb_dim"])
    kw = dict(zip([str(k) for k in g["kw_keys"]], [float(v) for v in g["kw_vals"]]))
    names = ["C%d" % (i + 1) for i in range(len(vocab))]
    dnames = ["I%d" % (i + 1) for i in range(nd)]
    spec = orc.Spec(names, vocab, dnames, D, tuple(int(v) for v in g["cin"]), True, "relu", tuple(int(v) for v in g["dnn"]),
                    l2_reg_dnn=1e-5)
    pro = orc.ProSpec(sfg_weight=kw.get("sfg_weight", 0.1), sfg_hidden_units=tuple(int(v) for v in g["sfg_hidden"]),
                      sfg_positive_only=bool(kw.get("sfg_positive_only", 1.0)),
                      sfg_use_label_attention=bool(kw.get("sfg_use_label_attention", 1.0)), use_autodis=bool(kw.get("use_autodis", 0.0)))
    return spec, pro


@pytest.mark.parametrize("name", golden_names("pro_"))
def test_pro_oracle_vs_reference_golden(name):
    """deepctr/xdeepfm_pro (SFG decoder, label-aware gate, masked CE / MSE, AutoDis): forward, sfg loss, every gradient of
    BCE + L2 + sfg_weight * sfg_loss against the values the reference produced (tests/golden/make_golden.py pro)."""
    if name == "pro_fit_history":
        pytest.skip("History of the fit loop: checked on the GPU path")
    g = load_golden(name)
    spec, pro = _pro_case(g)
    B = int(g["B"])
    state = {k[3:]: T(g[k]).clone().requires_grad_(True) for k in g if k.startswith("s0:")}
    X, y = T(g["X"][:B]).float(), T(g["y"][:B]).float()
    tot, loss, sfg, y_pred = orc.pro_total_loss(X, y, state, spec, pro)
    np.testing.assert_allclose(y_pred.detach().numpy(), g["y_pred"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)
    np.testing.assert_allclose(sfg.item(), float(g["sfg"]), rtol=1e-5)
    tot.backward()
    for k, v in state.items():
        want = g["g:" + k]
        got = v.grad.numpy() if v.grad is not None else np.zeros_like(want)
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6 * max(1.0, float(np.abs(want).max())), err_msg=k)
