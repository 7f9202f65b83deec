"""GPU parity tests (-m gpu) of the xdeepfm_pro path: xDeepFMPro on the HIP kernels against the golden
vectors the reference produced (tests/golden/make_golden.py pro) and the tiled vocabulary cross-entropy against
F.cross_entropy on materialised logits."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _build(g, dev):
    from deepctr.inputs import DenseFeat, SparseFeat
    from deepctr.xdeepfm_pro import xDeepFMPro
    vocab, nd, D = [int(v) for v in g["vocab"]], int(g["n_dense"]), int(g["emb_dim"])
    kw = dict(zip([str(k) for k in g["kw_keys"]], [float(v) for v in g["kw_vals"]]))
    for k in ("sfg_positive_only", "sfg_use_label_attention", "use_autodis"):
        if k in kw:
            kw[k] = bool(kw[k])
    if "autodis_buckets" in kw:
        kw["autodis_buckets"] = int(kw["autodis_buckets"])
    cols = [SparseFeat("C%d" % (i + 1), v, D) for i, v in enumerate(vocab)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(nd)]
    model = xDeepFMPro(cols, cols, dnn_hidden_units=tuple(int(v) for v in g["dnn"]), cin_layer_size=tuple(int(v) for v in g["cin"]),
                       l2_reg_dnn=1e-5, device=dev, sfg_hidden_units=tuple(int(v) for v in g["sfg_hidden"]), sfg_dropout=0.0, **kw)
    return model


@pytest.mark.parametrize("name", [n for n in golden_names("pro_") if n != "pro_fit_history"])
def test_pro_model_vs_reference_golden(name):
    """xDeepFMPro (deepctr/xdeepfm_pro/xdeepfm_pro.py:31-274): construction draws the reference's initial weights (same
    RNG order, same state_dict keys); forward_with_sfg -> y_pred, BCE, sfg loss; every gradient of
    BCE + L2 + sfg_weight * sfg_loss; three steps of the fit loop body (basemodel_sfg.py:317-349); predict."""
    from xdfm_amd import ops
    dev = _dev()
    g = load_golden(name)
    model = _build(g, dev)
    fused_before = ops.VocabHeadsCE.calls
    sd = model.state_dict()
    init = {k[5:]: g[k] for k in g if k.startswith("init:")}
    assert sorted(sd.keys()) == sorted(init.keys())
    for k, v in init.items():
        np.testing.assert_array_equal(sd[k].cpu().numpy(), v, err_msg="initial " + k)
    model.load_state_dict({k[3:]: T(g[k]) for k in g if k.startswith("s0:")})
    model.compile("adam", "binary_crossentropy", metrics=[])
    model.train()
    B = int(g["B"])
    X, y = g["X"], g["y"]
    xb, yb = T(X[:B]).float().to(dev), T(y[:B]).float().to(dev)
    y_pred, info = model.forward_with_sfg(xb, yb)
    np.testing.assert_allclose(y_pred.detach().cpu().numpy(), g["y_pred"], rtol=2e-5, atol=2e-6)
    loss = torch.nn.functional.binary_cross_entropy(y_pred.squeeze(), yb.squeeze(), reduction="sum")
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=2e-5)
    np.testing.assert_allclose(info["sfg_loss"].item(), float(g["sfg"]), rtol=2e-5)
    reg = model.get_regularization_loss()
    np.testing.assert_allclose(reg.item(), float(g["reg"]), rtol=2e-5)
    model.optim.zero_grad()
    (loss + reg + model.sfg_weight * info["sfg_loss"]).backward()
    for k, p in model.named_parameters():
        want = g["g:" + k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
        np.testing.assert_allclose(got, want, rtol=3e-4, atol=3e-5 * float(np.abs(want).max()) + 1e-9, err_msg="grad " + k)
    model.optim.zero_grad()
    # three steps of the product's train step (forward_with_sfg, BCE + L2 + sfg, backward, Adam)
    for s in range(3):
        xs, ys = T(X[s * B:(s + 1) * B]).float().to(dev), T(y[s * B:(s + 1) * B]).float().to(dev)
        _, l, tot = model.train_on_batch(xs, ys)
        np.testing.assert_allclose([l.item(), tot.item()], g["losses3"][s][:2], rtol=2e-4)
    for k, v in model.state_dict().items():
        want = g["s3:" + k]
        np.testing.assert_allclose(v.cpu().numpy(), want, rtol=2e-3, atol=3e-6 + 2e-4 * float(np.abs(want).max()) * 1e-2, err_msg="after 3 steps " + k)
    pred = model.predict([X[:, i] for i in range(X.shape[1])], batch_size=B)
    np.testing.assert_allclose(pred, g["pred_after"], rtol=2e-4, atol=2e-6)
    # heads of width 32 / 64 take the fused kernels (logits only in MFMA accumulators), the others the tiled path
    fused = ops.VocabHeadsCE.calls - fused_before
    assert fused == (4 if int(g["sfg_hidden"][-1]) in (32, 64) else 0), fused


def test_pro_fit_history_vs_reference_golden():
    """BaseModelSFG.fit (basemodel_sfg.py:224-400), 2 epochs, shuffle=False: History keys and values incl. `sfg_loss`
    (zero in the second epoch: after the first validation the reference's loop -- and this one -- stays in eval mode)."""
    from deepctr.inputs import DenseFeat, SparseFeat
    from deepctr.xdeepfm_pro import xDeepFMPro
    dev = _dev()
    g = load_golden("pro_fit_history")
    vocab, nd, D = [int(v) for v in g["vocab"]], int(g["n_dense"]), int(g["emb_dim"])
    cols = [SparseFeat("C%d" % (i + 1), v, D) for i, v in enumerate(vocab)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(nd)]
    model = xDeepFMPro(cols, cols, dnn_hidden_units=(8,), cin_layer_size=(6, 4), l2_reg_dnn=1e-5, device=dev,
                       sfg_hidden_units=(8, 6), sfg_dropout=0.0, sfg_weight=0.2)
    for k, v in model.state_dict().items():
        np.testing.assert_array_equal(v.cpu().numpy(), g["s0:" + k], err_msg=k)
    model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
    for pg in model.optim.param_groups:
        pg["lr"] = 1e-2
    names = list(model.feature_index.keys())
    hist = model.fit({n: g["X"][:, i] for i, n in enumerate(names)}, g["y"], batch_size=64, epochs=2, verbose=2,
                     validation_data=({n: g["Xv"][:, i] for i, n in enumerate(names)}, g["yv"]), shuffle=False)
    keys = [str(k) for k in g["hist_keys"]]
    assert sorted(hist.history.keys()) == keys
    got = np.array([hist.history[k] for k in keys])
    np.testing.assert_allclose(got, g["hist_vals"], rtol=5e-4, atol=1e-6)
    assert hist.history["sfg_loss"][1] == 0.0 and hist.history["sfg_loss"][0] > 0
    pred = model.predict({n: g["Xv"][:, i] for i, n in enumerate(names)}, 32)
    np.testing.assert_allclose(pred, g["pred"], rtol=5e-4, atol=1e-6)


@pytest.mark.parametrize("rows,K,V,tile_bytes", [(300, 64, 50000, 1 << 20), (17, 32, 1000, 1 << 30), (1, 8, 5, 1 << 10),
                                                  (1000, 64, 262144, 64 << 20)])
def test_vocab_softmax_ce_tiles_vs_materialised_logits(rows, K, V, tile_bytes):
    """ops.VocabSoftmaxCE: nn.Linear(K, V) + F.cross_entropy(reduction='none') (sfg_decoder.py:146-149, :277-283) without the
    [rows, V] logits -- forward and all three gradients against the materialised computation in float64."""
    from xdfm_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(rows + V)
    h = torch.randn(rows, K, generator=g).to(dev).requires_grad_(True)
    W = (torch.randn(V, K, generator=g) * 0.3).to(dev).requires_grad_(True)
    b = (torch.randn(V, generator=g) * 0.3).to(dev).requires_grad_(True)
    tgt = torch.randint(0, V, (rows,), generator=g).to(dev)
    gout = torch.rand(rows, generator=g).to(dev)
    old = ops.VOCAB_TILE_BYTES
    ops.VOCAB_TILE_BYTES = tile_bytes
    try:
        ce = ops.vocab_softmax_ce(h, W, b, tgt.float())
        (ce * gout).sum().backward()
    finally:
        ops.VOCAB_TILE_BYTES = old
    h64, W64, b64 = (t.detach().double().requires_grad_(True) for t in (h, W, b))
    want = torch.nn.functional.cross_entropy(torch.nn.functional.linear(h64, W64, b64), tgt, reduction="none")
    (want * gout.double()).sum().backward()
    np.testing.assert_allclose(ce.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
    for got, w, name in ((h.grad, h64.grad, "dh"), (W.grad, W64.grad, "dW"), (b.grad, b64.grad, "db")):
        w = w.cpu().numpy()
        np.testing.assert_allclose(got.cpu().numpy(), w, rtol=2e-4, atol=2e-5 * float(np.abs(w).max()), err_msg=name)


@pytest.mark.parametrize("rows,K,vocabs,hs,ws_", [(300, 64, (50000, 77, 1000), 1.0, 0.3), (17, 32, (1000, 33), 1.0, 0.3),
                                                  (1, 64, (5,), 1.0, 0.3), (1100, 64, (70001, 256), 1.0, 0.3),
                                                  (513, 32, (4097,), 1.0, 0.3),
                                                  (4096, 64, (3000, 129), 1.0, 0.3),          # eight row groups (no positive-only filter)
                                                  (200, 64, (5000,), 1.0e3, 3.0e-4),          # operand ranges far from 1: the power-of-two scales
                                                  (200, 64, (5000,), 1.0, 1.5)])              # logits of +-40: the log-sum-exp's range
def test_vocab_heads_ce_fused_vs_materialised_logits(rows, K, vocabs, hs, ws_):
    """ops.VocabHeadsCE (csrc/vocab_ce_x3.hip): the heads of all sparse fields over the same hidden rows with the logits
    only in MFMA accumulators (sfg_decoder.py:146-149 + :277-283) -- losses and all gradients against the materialised
    computation in float64; ragged row counts (tiles of 32, row groups of 512), vocabularies that are no multiple of the
    256-row stages, a second row group (1100 rows), one-row / one-entry edge cases.  Tolerances: the f16x3 products carry
    ~2^-22 relative error per term, the base-2 exp / log ~1 ulp; same bars as the tiled path's test above."""
    from xdfm_amd import ops
    dev = _dev()
    assert ops.vocab_heads_ce_supported(K) and not ops.vocab_heads_ce_supported(48)
    g = torch.Generator().manual_seed(rows + sum(vocabs))
    h = (torch.randn(rows, K, generator=g) * hs).to(dev).requires_grad_(True)
    Ws = [(torch.randn(V, K, generator=g) * ws_).to(dev).requires_grad_(True) for V in vocabs]
    bs = [(torch.randn(V, generator=g) * 0.3).to(dev).requires_grad_(True) for V in vocabs]
    tgt = torch.stack([torch.randint(0, V, (rows,), generator=g) for V in vocabs]).to(dev)
    gout = (torch.rand(len(vocabs), rows, generator=g) - 0.3) * 1e-3         # both signs
    gout[:, ::7] = 0.0                                                        # and exact zeros (rows the loss masks)
    gout = gout.to(dev)
    ce = ops.vocab_heads_ce(h, tgt, Ws, bs)
    (ce * gout).sum().backward()
    h64 = h.detach().double().requires_grad_(True)
    W64 = [w.detach().double().requires_grad_(True) for w in Ws]
    b64 = [b.detach().double().requires_grad_(True) for b in bs]
    want = torch.stack([torch.nn.functional.cross_entropy(torch.nn.functional.linear(h64, w, b), tgt[f], reduction="none")
                        for f, (w, b) in enumerate(zip(W64, b64))])
    (want * gout.double()).sum().backward()
    np.testing.assert_allclose(ce.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
    pairs = [(h.grad, h64.grad, "dh")] + [(w.grad, w6.grad, "dW%d" % f) for f, (w, w6) in enumerate(zip(Ws, W64))] + \
            [(b.grad, b6.grad, "db%d" % f) for f, (b, b6) in enumerate(zip(bs, b64))]
    for got, w, name in pairs:
        w = w.cpu().numpy()
        np.testing.assert_allclose(got.cpu().numpy(), w, rtol=2e-4, atol=2e-5 * float(np.abs(w).max()), err_msg=name)
    # determinism: fixed summation orders, no atomics on floats
    h2 = h.detach().clone().requires_grad_(True)
    W2 = [w.detach().clone().requires_grad_(True) for w in Ws]
    b2 = [b.detach().clone().requires_grad_(True) for b in bs]
    ce2 = ops.vocab_heads_ce(h2, tgt, W2, b2)
    (ce2 * gout).sum().backward()
    assert torch.equal(ce2, ce) and torch.equal(h2.grad, h.grad)
    assert all(torch.equal(a.grad, b_.grad) for a, b_ in zip(W2, Ws))
