"""CPU: host-side mirror of the reference interface (feature columns, column map, CIN geometry,
constructors / error behaviour, callbacks, metrics, state_dict contract)."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def test_feature_columns_and_index():
    from deepctr.inputs import DenseFeat, SparseFeat, VarLenSparseFeat, build_input_features, get_feature_names
    s = SparseFeat("C1", 100, embedding_dim=8)
    assert s.embedding_name == "C1" and s.group_name == "default_group" and s.dtype == "int32"
    assert SparseFeat("C2", 10000, "auto").embedding_dim == 6 * int(pow(10000, 0.25))
    assert hash(s) == hash("C1") and s == SparseFeat("C1", 100, 8)
    d = DenseFeat("I1")
    assert d.dimension == 1 and d.dtype == "float32"
    cols = [SparseFeat("C%d" % i, 10, 4) for i in range(3)] + [DenseFeat("I1", 1), DenseFeat("I2", 2)]
    idx = build_input_features(cols + cols)                     # linear_cols + dnn_cols, duplicates ignored
    assert list(idx.items()) == [("C0", (0, 1)), ("C1", (1, 2)), ("C2", (2, 3)), ("I1", (3, 4)), ("I2", (4, 6))]
    assert get_feature_names(cols) == ["C0", "C1", "C2", "I1", "I2"]
    v = VarLenSparseFeat(SparseFeat("h", 5, 4), maxlen=3, length_name="hl")
    assert list(build_input_features([v]).items()) == [("h", (0, 3)), ("hl", (3, 4))]
    with pytest.raises(TypeError):
        build_input_features([object()])


def test_cin_geometry_matches_reference_field_nums():
    from xdfm_amd.ops import cin_geometry
    lv, fm = cin_geometry(26, (256, 128, 128), True)
    assert fm == 320 and [(l[0], l[1]) for l in lv] == [(256, 26), (128, 128), (128, 64)]
    assert [(l[2], l[3], l[4], l[5]) for l in lv] == [(128, 128, 128, 0), (64, 64, 64, 128), (0, 0, 128, 192)]
    lv, fm = cin_geometry(4, (6, 5), False)
    assert fm == 11 and [(l[0], l[1], l[2], l[4]) for l in lv] == [(6, 4, 6, 6), (5, 6, 5, 5)]
    lv, fm = cin_geometry(3, (8, 5), True)
    assert fm == 9 and lv[1][:2] == (5, 4)


def test_constructor_contract_and_state_dict_keys():
    from deepctr.inputs import DenseFeat, SparseFeat
    from deepctr.models import xDeepFM, xDeepFMAttention, xDeepFMAttentionV2
    cols = [SparseFeat("C%d" % (i + 1), 20, 16) for i in range(26)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(13)]
    m = xDeepFM(cols, cols, cin_layer_size=(256, 128, 128))
    sd = m.state_dict()
    assert tuple(sd["cin.conv1ds.0.weight"].shape) == (256, 676, 1)
    assert tuple(sd["cin.conv1ds.1.weight"].shape) == (128, 3328, 1)
    assert tuple(sd["cin.conv1ds.2.weight"].shape) == (128, 1664, 1)
    assert tuple(sd["cin_linear.weight"].shape) == (1, 320)
    assert tuple(sd["dnn.linears.0.weight"].shape) == (256, 429)
    assert tuple(sd["linear_model.weight"].shape) == (13, 1) and tuple(sd["out.bias"].shape) == (1,)
    assert tuple(sd["linear_model.embedding_dict.C7.weight"].shape) == (20, 1)
    assert m.featuremap_num == 320 and list(m.feature_index)[:2] == ["C1", "C2"]
    a = xDeepFMAttention(cols, cols, cin_layer_size=(256, 128, 128))
    for k in ("cin.mhsa.W_q.weight", "cin.layer_norm.weight", "cin.attn_pooling.attention.0.bias",
              "cin.attn_pooling.attention.2.weight"):
        assert k in a.state_dict()
    assert tuple(a.state_dict()["cin.output_proj.weight"].shape) == (320, 16)
    v2 = xDeepFMAttentionV2(cols, cols, cin_num_attn_layers=2)
    assert "cin.mhsa_layers.1.W_o.weight" in v2.state_dict() and tuple(v2.state_dict()["cin_linear.weight"].shape) == (1, 16)
    with pytest.raises(ValueError):
        xDeepFM(cols, cols, gpus=[0])
    with pytest.raises(ValueError):
        xDeepFM(cols, cols, cin_layer_size=(7, 4))
    with pytest.raises(NotImplementedError):
        xDeepFM(cols, cols, cin_activation="prelu")
    # regularisation groups (basemodel.py:126-127, xdeepfm.py:57-60,74-75)
    groups = [(len(w), l2) for w, _, l2 in m.regularization_weight]
    assert groups == [(26, 1e-5), (27, 1e-5), (2, 0), (1, 0), (3, 0)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.get_regularization_loss()                 # K6 runs on the GPU only


def test_metrics_vs_sklearn_golden():
    from xdfm_amd import metrics as M
    g = load_golden("metrics")
    assert abs(M.log_loss(g["y"], g["p"]) - float(g["logloss"])) < 1e-12
    assert abs(M.roc_auc_score(g["y"], g["p"]) - float(g["auc"])) < 1e-12
    with pytest.raises(ValueError):
        M.roc_auc_score(np.ones(4), np.arange(4))


def test_device_metrics_equal_the_numpy_metrics():
    """xdfm_amd.metrics.*_device (what `fit` logs per step without a host sync) against the numpy versions that are
    pinned to sklearn above: AUC bit for bit (ties, repeated scores, tiny batches), logloss / mse to 1e-12; a
    single-class batch gives NaN where the host version raises."""
    import torch
    from xdfm_amd import metrics as M
    rng = np.random.default_rng(7)
    for n, ties in ((4096, False), (4096, True), (100, True), (2, False)):
        y = (rng.random(n) < 0.3).astype(np.float32)
        y[0], y[-1] = 1.0, 0.0
        p = rng.random(n).astype(np.float32)
        if ties:
            p = np.round(p * 20) / 20                      # many equal scores, incl. 0 and 1
        yt, pt = torch.from_numpy(y), torch.from_numpy(p)
        assert float(M.roc_auc_score_device(yt, pt)) == M.roc_auc_score(y, p.astype(np.float64))
        assert abs(float(M.log_loss_device(yt, pt)) - M.log_loss(y, p.astype(np.float64))) < 1e-12
        assert abs(float(M.mean_squared_error_device(yt, pt)) - M.mean_squared_error(y, p.astype(np.float64))) < 1e-12
    ones = torch.ones(8)
    assert torch.isnan(M.roc_auc_score_device(ones, torch.rand(8)))
    assert set(M.DEVICE) == {M.log_loss, M.roc_auc_score, M.mean_squared_error}


def test_callbacks_keras_semantics(tmp_path):
    from deepctr.callbacks import CallbackList, EarlyStopping, History, ModelCheckpoint

    class Dummy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(2))
            self.stop_training = False
    model = Dummy()
    es = EarlyStopping(monitor="val_auc", patience=2, mode="max", verbose=0)
    ck = ModelCheckpoint(str(tmp_path / "best.pth"), monitor="val_auc", save_best_only=True, save_weights_only=True,
                         mode="max", verbose=0)
    hist = History()
    seen = []

    class Duck(object):                         # the scripts' callbacks are duck-typed (xdftrain.py:31-97)
        def on_epoch_end(self, epoch, logs=None):
            seen.append(epoch)
    cbs = CallbackList([es, ck, Duck(), hist])
    cbs.set_model(model)
    cbs.on_train_begin()
    aucs = [0.6, 0.7, 0.65, 0.69, 0.5]
    for e, a in enumerate(aucs):
        with torch.no_grad():
            model.w.fill_(float(e))
        cbs.on_epoch_begin(e)
        cbs.on_epoch_end(e, {"loss": 1.0 - a, "val_auc": a})
        if model.stop_training:
            break
    cbs.on_train_end()
    assert model.stop_training and es.stopped_epoch == 3 and seen == [0, 1, 2, 3]
    assert hist.history["val_auc"] == aucs[:4] and hist.epoch == [0, 1, 2, 3]
    best = torch.load(str(tmp_path / "best.pth"), weights_only=True)
    assert float(best["w"][0]) == 1.0           # epoch index 1 had the best val_auc
    assert EarlyStopping(monitor="val_loss").monitor_op == np.less
    assert EarlyStopping(monitor="val_auc").monitor_op == np.greater


def test_fm_layout_roundtrip():
    from xdfm_amd import ops
    x = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 3, 4)
    fm = ops.to_fm_layout(x)
    assert fm.shape == (3, 8) and float(fm[1, 4 + 2]) == float(x[1, 1, 2])
    assert torch.equal(ops.from_fm_layout(fm, 2, 4), x)


def test_epoch_order_reproduces_dataloader_batches():
    """fit's vectorised batching must visit the rows in the order DataLoader(shuffle=...) of the
    reference's fit would (deepctr/models/basemodel.py:213-214,241) for the same torch seed, epoch after
    epoch, and leave the default generator in the same state."""
    import torch.utils.data as Data
    from xdfm_amd.models import epoch_order
    n, bs = 103, 16
    X = torch.arange(n, dtype=torch.float64).view(n, 1)
    for shuffle in (True, False):
        torch.manual_seed(1024)
        loader = Data.DataLoader(Data.TensorDataset(X, X), shuffle=shuffle, batch_size=bs)
        want = [[b[0].view(-1).tolist() for b in loader] for _ in range(3)]
        tail_ref = torch.rand(3)
        torch.manual_seed(1024)
        got = []
        for _ in range(3):
            order = epoch_order(n, shuffle)
            rows = X if order is None else X.index_select(0, order)
            got.append([rows[i:i + bs].view(-1).tolist() for i in range(0, n, bs)])
        assert got == want
        assert torch.equal(torch.rand(3), tail_ref)


def test_entry_point_reader_and_preprocessor(tmp_path):
    """xdftrain_amd.py host pieces: Criteo-format reader (tab / comma, header or not, blanks) and the
    unknown->0 / first-appearance->1..N encoder + min-max scaler (xdftrain.py:165-237 semantics)."""
    import importlib.util
    import os
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("xdftrain_amd", os.path.join(PKG, "xdftrain_amd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = [["1"] + [str(i) for i in range(13)] + ["a%d" % (k % 3) for k in range(26)],
            ["0"] + ["" if i == 2 else str(2 * i) for i in range(13)] + ["b"] * 25 + [""],
            ["1"] + [str(i + 5) for i in range(13)] + ["a0"] * 26]
    p = tmp_path / "train.txt"
    p.write_text("\n".join("\t".join(r) for r in rows) + "\n")
    t = mod.read_table(str(p))
    assert t["label"].tolist() == [1.0, 0.0, 1.0] and t["I3"][1] == 0.0 and t["C26"][1] == "-1"
    hdr = tmp_path / "hdr.csv"
    hdr.write_text(",".join(["label"] + mod.DENSE + mod.SPARSE) + "\n" + "\n".join(",".join(r) for r in rows) + "\n")
    t2 = mod.read_table(str(hdr))
    assert t2["label"].tolist() == t["label"].tolist() and list(t2["C1"]) == list(t["C1"])
    prep = mod.Preprocessor().fit(mod.take(t, np.array([0, 1])))
    x = prep.transform(t)
    assert x["C1"].tolist() == [1, 2, 1] and x["C2"].tolist() == [1, 2, 0]      # "a0" unseen for C2 -> 0
    assert prep.vocab("C1") == 3 and x["I2"].dtype == np.float32
    assert x["I2"][0] == 0.0 and x["I2"][1] == 1.0 and abs(x["I2"][2] - 5.0) < 1e-6   # (6 - 1) / (2 - 1): scaled with the fit range


def test_entry_point_flags_follow_the_three_reference_scripts():
    """xdftrain_amd.py accepts the command lines of xdftrain.py / xdftrain_attn.py / xdftrain_pro.py (run.bash,
    run_attn.bash, run_sfg.bash flows): --mode eval|final, --cin_attn_dropout, the SFG / AutoDis flags, --stratify,
    and each script's own defaults for epochs / batch sizes / out_dir (xdftrain.py:725-727, xdftrain_attn.py:747-749,
    xdftrain_pro.py:795-797)."""
    import importlib.util
    import os
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("xdftrain_amd", os.path.join(PKG, "xdftrain_amd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = mod.parse_args(["--data_path", "x.txt"])
    assert (a.mode, a.model, a.epochs, a.batch_size, a.pred_batch_size, a.out_dir) == \
        ("eval", "xdeepfm", 3, 4096, 8192, "./outputs_xdeepfm")
    a = mod.parse_args(["--data_path", "x.txt", "--mode", "final", "--model_version", "v2", "--cin_attn_dropout", "0.1",
                        "--cin_no_residual", "--cin_num_attn_layers", "2", "--stratify"], model="attn")
    assert (a.mode, a.model, a.epochs, a.cin_attn_dropout, a.cin_use_residual, a.cin_use_layer_norm, a.stratify) == \
        ("final", "attn", 50, 0.1, False, True, True)
    a = mod.parse_args(["--data_path", "x.txt", "--sfg_hidden_units", "64", "32", "--sfg_all_samples", "--use_autodis",
                        "--autodis_buckets", "8", "--use_light_version", "--epochs", "5"], model="pro")
    assert (a.model, a.epochs, a.batch_size, a.pred_batch_size, a.sfg_hidden_units, a.sfg_positive_only, a.use_sfg,
            a.use_autodis, a.autodis_buckets, a.use_light_version, a.out_dir) == \
        ("pro", 5, 2048, 4096, [64, 32], False, True, True, 8, True, "./outputs_xdeepfm_pro")
    assert mod.parse_args(["--no_sfg"], model="pro").use_sfg is False
    a = mod.parse_args(["--data_path", "x.txt"], model="xdeepfm", script="v1")           # xdftrain_v1.py:629-653
    assert (a.epochs, a.test_size, a.val_size, a.patience, a.use_early_stopping, a.out_dir) == \
        (20, 0.2, 0.2, 2, True, "./outputs_xdeepfm")
    a = mod.parse_args(["--data_path", "x.txt"], model="xdeepfm")
    assert (a.epochs, a.test_size, a.val_size, a.patience, a.use_early_stopping) == (3, None, 0.1, 50, False)
    for shim in ("xdftrain.py", "xdftrain_attn.py", "xdftrain_pro.py", "xdftrain_v1.py"):
        assert os.path.exists(os.path.join(PKG, shim))
    # split: plain and stratified (class ratio kept in both parts, nothing lost, nothing shared)
    y = (np.arange(1000) % 10 == 0).astype(np.float64)
    for strat in (False, True):
        tr, va = mod.split_rows(y, 0.2, 7, strat)
        assert len(va) == 200 and sorted(np.concatenate([tr, va]).tolist()) == list(range(1000))
    assert y[va].sum() == 20 and y[tr].sum() == 80
    tr2, va2 = mod.split_rows(y, 0.2, 7, True)
    assert np.array_equal(tr, tr2) and np.array_equal(va, va2)


def test_table_adam_deferred_switch(monkeypatch):
    """TableAdam's `deferred`: True / False / "auto" (default), from the argument or XDFM_ADAM_DEFERRED = 1 / 0 / auto.
    "auto" takes the deferred table update when the gathers' tables hold at least DEFER_MIN_NUMEL parameters and the
    streaming sweep below (same bits either way, tests/test_gpu_host.py; DESIGN 4.3b item 4 has the crossover)."""
    from xdfm_amd import optim
    p = [torch.nn.Parameter(torch.zeros(4, 4))]
    monkeypatch.delenv("XDFM_ADAM_DEFERRED", raising=False)
    assert optim.TableAdam(p).deferred == "auto"
    for env, want in (("0", False), ("1", True), ("auto", "auto")):
        monkeypatch.setenv("XDFM_ADAM_DEFERRED", env)
        assert optim.TableAdam(p).deferred == want
        assert optim.TableAdam(p, deferred=True).deferred is True          # the argument wins over the environment
        assert optim.TableAdam(p, deferred=False).deferred is False
        assert optim.TableAdam(p, deferred="auto").deferred == "auto"
    assert optim.DEFER_MIN_NUMEL == 1 << 26
