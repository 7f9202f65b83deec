"""GPU tests (-m gpu) of the host logic around the captured train step and the drop-in behaviour of fit / predict:
optimizer state reloads, learning-rate schedules, deferred IndexError for bad ids, pickling of a trained model, and
the capture-after-collection incident of round 1 (tools/graph_crash_probe.py)."""
import copy
import gc
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy
VOCAB, ND, D = [50, 31, 77, 12, 9, 40], 3, 8



def _needs_default_env(feature):
    """Tests that assert a feature is ACTIVE skip when the environment switches it off (XDFM_GRAD_ARENA=0 / XDFM_HIP_GRAPH=0 /
    XDFM_ADAM_DEFERRED=0 are supported ways to run the product; the rest of the suite passes under them)."""
    import os
    env = {"arena": "XDFM_GRAD_ARENA", "graph": "XDFM_HIP_GRAPH", "deferred": "XDFM_ADAM_DEFERRED"}[feature]
    if os.environ.get(env, "1") == "0":
        pytest.skip("%s=0" % env)

def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model(dev, use_graph=True, cls=None, **kw):
    from deepctr.inputs import DenseFeat, SparseFeat
    from deepctr.models import xDeepFM
    from xdfm_amd import graphstep
    cols = [SparseFeat("C%d" % (i + 1), v, D) for i, v in enumerate(VOCAB)] + \
        [DenseFeat("I%d" % (i + 1), 1) for i in range(ND)]
    model = (cls or xDeepFM)(cols, cols, dnn_hidden_units=(32, 16), cin_layer_size=(16, 8), l2_reg_dnn=1e-5, device=dev, **kw)
    model.compile("adam", "binary_crossentropy", metrics=[])
    model.train()
    step = graphstep.GraphedStep(model)
    step.disabled = not use_graph
    model.__dict__["_graphed_step"] = step
    return model, step


def _batch(s, rows=256):
    from oracle import xdeepfm_oracle as orc
    X, y = orc.synthetic_batch(rows, VOCAB, ND, seed=300 + s)
    return T(X), T(y)


def test_optimizer_state_reload_is_honoured_by_the_captured_step():
    """the captured K7 launches bake the addresses of exp_avg / exp_avg_sq / step.  After
    `optim.load_state_dict` (resume) those tensors are new ones; the graph key carries TableAdam.generation, so the
    step is captured again instead of updating the old, freed buffers.  Graph run == eager twin throughout."""
    dev = _dev()

    def run(use_graph):
        model, step = _model(dev, use_graph)
        saved = None
        for s in range(9):
            if s == 2:
                saved = copy.deepcopy(model.optim.state_dict())
            if s == 5:
                model.optim.load_state_dict(saved)            # moments and step counters go back to step 2
            X, y = _batch(s)
            model.train_on_batch(X.to(dev), y.to(dev))
        return model, step

    m_g, st_g = run(True)
    m_e, st_e = run(False)
    assert st_g.replays >= 3 and st_e.replays == 0 and not st_g.disabled
    assert len([e for e in st_g.entries.values() if e.graph is not None]) == 2     # before and after the reload
    for (k, a), (_, b) in zip(m_g.state_dict().items(), m_e.state_dict().items()):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
    for pa, pb in zip(m_g.optim.param_groups[0]["params"], m_e.optim.param_groups[0]["params"]):
        np.testing.assert_allclose(m_g.optim.state[pa]["exp_avg"].cpu().numpy(),
                                   m_e.optim.state[pb]["exp_avg"].cpu().numpy(), rtol=1e-3, atol=1e-6)
        assert float(m_g.optim.state[pa]["step"]) == float(m_e.optim.state[pb]["step"]) == 6.0


def test_fit_and_predict_raise_index_error_for_ids_outside_the_vocabulary():
    """The reference's nn.Embedding raises IndexError on an id >= vocabulary_size (basemodel.py:368-370).  K1 clamps and
    raises a device flag; `fit` reads it with the epoch's losses, `predict` with its final copy."""
    from oracle import xdeepfm_oracle as orc
    dev = _dev()
    model, _ = _model(dev)
    names = list(model.feature_index.keys())
    X, y = orc.synthetic_batch(200, VOCAB, ND, seed=1)
    ok = {n: X[:, i] for i, n in enumerate(names)}
    model.fit(ok, y, batch_size=64, epochs=1, verbose=0)
    model.predict(ok, 64)
    Xb = X.copy()
    Xb[137, 2] = VOCAB[2]                                       # one id == vocabulary_size (nunique instead of max + 1)
    bad = {n: Xb[:, i] for i, n in enumerate(names)}
    with pytest.raises(IndexError, match="vocabulary_size"):
        model.predict(bad, 64)
    model.predict(ok, 64)                                        # the flag was cleared
    with pytest.raises(IndexError):
        model.fit(bad, y, batch_size=64, epochs=1, verbose=0)
    Xn = X.copy()
    Xn[3, 0] = -1.0
    with pytest.raises(IndexError):
        model.predict({n: Xn[:, i] for i, n in enumerate(names)}, 64)


def test_torch_save_of_the_whole_model_after_fit():
    """ModelCheckpoint(save_weights_only=False) calls torch.save(model) (deepctr/callbacks.py:41-73): after a fit the
    model holds a captured graph, ctypes descriptors and device-side plans -- they are dropped from the pickle and
    rebuilt on first use."""
    from oracle import xdeepfm_oracle as orc
    dev = _dev()
    model, _ = _model(dev)
    names = list(model.feature_index.keys())
    X, y = orc.synthetic_batch(512, VOCAB, ND, seed=2)
    data = {n: X[:, i] for i, n in enumerate(names)}
    model.fit(data, y, batch_size=64, epochs=2, verbose=0)
    assert model.__dict__["_graphed_step"].replays > 0
    want = model.predict(data, 128)
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    twin = torch.load(buf, weights_only=False)                   # our own file
    np.testing.assert_array_equal(twin.predict(data, 128), want)
    twin.fit(data, y, batch_size=64, epochs=1, verbose=0, shuffle=False)     # and it keeps training (plans / graph rebuilt)
    model.fit(data, y, batch_size=64, epochs=1, verbose=0, shuffle=False)
    np.testing.assert_allclose(twin.predict(data, 128), model.predict(data, 128), rtol=1e-4, atol=1e-6)


def test_tables_have_no_gradient_views_left_after_the_own_step():
    """After the model's own step the tables' .grad must not stay views of the kept gradient buffer: a user loop
    (backward -> step -> zero_grad) afterwards would accumulate into it unmarked."""
    _needs_default_env('arena')
    dev = _dev()
    model, _ = _model(dev)
    for s in range(4):
        X, y = _batch(s)
        model.train_on_batch(X.to(dev), y.to(dev))
        assert all(p.grad is None for p in model.embedding_dict.parameters())
        assert all(p.grad is None for p in model.linear_model.parameters())
    arena, = model._plan.arenas()
    torch.cuda.synchronize()
    assert not arena.pending and float(arena.flat.abs().max()) == 0.0 and int(arena.marks.max()) == 0


def test_capture_while_an_older_models_graphs_are_being_collected():
    """Round-1 incident (gpurun_out/gputest2.log: segfault in capture_end): a graph owned by a dropped model was
    destroyed by the cycle collector in the middle of a later capture.  Two models alive, the first one (with a
    captured graph) dropped without an explicit collection, then the second one captures: the step object holds its
    model weakly, unreachable graphs are destroyed before the capture begins and the collector is paused during it."""
    dev = _dev()
    m1, s1 = _model(dev)
    m2, s2 = _model(dev)
    for s in range(5):
        X, y = _batch(s)
        m1.train_on_batch(X.to(dev), y.to(dev))
    assert s1.replays >= 2
    for s in range(2):                                           # the two eager steps that precede m2's capture
        X, y = _batch(s)
        m2.train_on_batch(X.to(dev), y.to(dev))
    cyc = [m1]                                                   # a reference cycle: only the collector can free m1
    cyc.append(cyc)
    del m1, s1, cyc
    gc.enable()
    out = None
    for s in range(2, 8):
        X, y = _batch(s)
        junk = [[i] for i in range(2000)]                        # allocation pressure: generation-0 collections
        out = m2.train_on_batch(X.to(dev), y.to(dev))
        del junk
    torch.cuda.synchronize()
    assert s2.replays >= 4 and not s2.disabled
    assert np.isfinite(float(out[2]))


def test_lazy_rows_opt_in_updates_only_touched_rows():
    """TableAdam(lazy_rows=True) is an OPT-IN deviation: rows a batch does not touch keep weight and
    moments; touched rows and all dense weights follow the reference's update.  One step from fresh state: touched
    rows == dense Adam's, untouched rows == initial values (dense Adam moves them by the L2 term)."""
    _needs_default_env('arena')
    from xdfm_amd.optim import TableAdam
    from deepctr.inputs import DenseFeat, SparseFeat
    from deepctr.models import xDeepFM
    dev = _dev()
    vocab = [500, 400, 300]
    cols = [SparseFeat("C%d" % (i + 1), v, D) for i, v in enumerate(vocab)] + [DenseFeat("I1", 1)]

    def make(lazy):
        m = xDeepFM(cols, cols, dnn_hidden_units=(16,), cin_layer_size=(8, 4), l2_reg_dnn=1e-5, l2_reg_embedding=1e-2,
                    init_std=0.1, device=dev)
        m.compile(TableAdam(m.parameters(), lazy_rows=lazy), "binary_crossentropy", metrics=[])
        m.train()
        return m

    rng = np.random.default_rng(3)
    X = np.concatenate([np.stack([rng.integers(0, 40, 64) for _ in vocab], 1), rng.random((64, 1))], 1).astype(np.float32)
    y = (rng.random((64, 1)) < 0.5).astype(np.float32)
    m_lazy, m_dense = make(True), make(False)
    init = {k: v.detach().cpu().numpy().copy() for k, v in m_lazy.state_dict().items()}
    for m in (m_lazy, m_dense):
        m.train_on_batch(T(X).to(dev), T(y).to(dev))
    torch.cuda.synchronize()
    sl, sd = m_lazy.state_dict(), m_dense.state_dict()
    for j, v in enumerate(vocab):
        rows = np.zeros(v, dtype=bool)
        rows[X[:, j].astype(int)] = True
        for key in ("embedding_dict.C%d.weight" % (j + 1), "linear_model.embedding_dict.C%d.weight" % (j + 1)):
            a, b = sl[key].cpu().numpy(), sd[key].cpu().numpy()
            touched = rows
            if a.shape[1] == 1:         # the unit is the 16-byte chunk: 4 neighbouring rows of a [V, 1] table move together
                touched = np.repeat(rows[:v // 4 * 4].reshape(-1, 4).any(1), 4)
                touched = np.concatenate([touched, np.full(v - touched.size, rows[v // 4 * 4:].any())])
            np.testing.assert_allclose(a[touched], b[touched], rtol=1e-5, atol=1e-7, err_msg=key)
            np.testing.assert_array_equal(a[~touched], init[key][~touched], err_msg=key + " (untouched rows moved)")
            assert np.abs(b[~touched] - init[key][~touched]).max() > 0       # the reference's dense Adam does move them
    for key in sl:
        if "embedding_dict" not in key:
            np.testing.assert_allclose(sl[key].cpu().numpy(), sd[key].cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=key)


def test_attention_dropout_draws_a_new_mask_on_every_graph_replay():
    """cin_attn_dropout > 0 under the captured train step: the seed of K5's mask is a device scalar drawn from torch's
    generator inside the capture, so a replay advances it (a host-side seed would be baked into the graph and every
    step would drop the same weights).  Same batch, zero learning rate: the losses of consecutive replays differ, and
    re-seeding torch repeats the sequence.  Evaluation ignores the rate."""
    from deepctr.models import xDeepFMAttention
    dev = _dev()
    torch.manual_seed(5)
    model, step = _model(dev, True, cls=xDeepFMAttention, cin_attn_dropout=0.5, cin_num_heads=2)
    with torch.no_grad():                       # the default init (std 1e-4) makes the attention branch vanish in the loss
        for k, p in model.named_parameters():
            if "cin." in k or "cin_linear" in k:
                p.mul_(3000.0 if p.abs().max() < 1e-2 else 1.0)
    for g in model.optim.param_groups:
        g["lr"] = 0.0
    X, y = _batch(0)
    X, y = X.to(dev), y.to(dev)

    def losses(n):
        return [float(model.train_on_batch(X, y)[1]) for _ in range(n)]
    torch.manual_seed(11)
    a = losses(6)
    assert step.replays >= 3 and not step.disabled
    assert len(set(a[2:])) == len(a[2:]), a          # replayed steps: all different masks
    model.eval()
    with torch.no_grad():
        e1 = model(X).clone()
        e2 = model(X)
    assert torch.equal(e1, e2)
    model.train()


@pytest.mark.parametrize("argv,files", [
    (["--model", "attn", "--mode", "final", "--cin_attn_dropout", "0.1", "--cin_layer_size", "16,8",
      "--dnn_hidden_units", "32,16"], ["xdeepfm_attn_full_weights.pth", "history_full.json", "training_log_full.json"]),
    (["--model", "attn", "--model_version", "v2", "--cin_num_attn_layers", "2", "--cin_attn_dropout", "0.2",
      "--cin_layer_size", "16,8", "--dnn_hidden_units", "32,16", "--stratify"],
     ["xdeepfm_attn_weights.pth", "history.json", "training_log.json", "best_model.pth"]),
    (["--model", "pro", "--use_light_version", "--sfg_hidden_units", "32", "16", "--cin_layer_size", "16,8",
      "--dnn_hidden_units", "32,16"], ["xdeepfm_pro_weights.pth", "history.json"]),
    (["--model", "xdeepfm", "--mode", "final", "--cin_layer_size", "16,8", "--dnn_hidden_units", "32,16"],
     ["xdeepfm_full_weights.pth", "history_full.json"]),
    # xdftrain_v1.py's flow: a held-out test split, reported at the end (training_log.json: test_logloss / test_auc)
    (["--model", "xdeepfm", "--mode", "eval", "--cin_layer_size", "16,8", "--dnn_hidden_units", "32,16",
      "--test_size", "0.2", "--val_size", "0.2", "--stratify"],
     ["best_model.pth", "xdeepfm_weights.pth", "training_log.json"]),
])
def test_entry_point_modes_run_end_to_end(tmp_path, argv, files):
    """xdftrain_amd.py through its `--mode eval` and `--mode final` flows (xdftrain.py:302-704) for the three model
    families on synthetic Criteo-shaped rows: runs, writes the artefacts, the logged loss is finite and decreasing."""
    import importlib.util
    import json
    import os
    from conftest import PKG
    _dev()
    spec = importlib.util.spec_from_file_location("xdftrain_amd", os.path.join(PKG, "xdftrain_amd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "out")
    mod.main(argv + ["--synthetic", "6000", "--epochs", "3", "--batch_size", "512", "--embedding_dim", "8",
                     "--out_dir", out, "--verbose", "0", "--learning_rate", "0.01"])
    for f in files + ["preprocess.json"]:
        assert os.path.exists(os.path.join(out, f)), f
    hist = json.load(open(os.path.join(out, "history_full.json" if "final" in argv else "history.json")))
    assert np.all(np.isfinite(hist["loss"])) and hist["loss"][-1] < hist["loss"][0]
    if "final" in argv:
        assert not any(k.startswith("val_") for k in hist) and "auc" not in hist
    else:
        assert 0.5 < hist["val_auc"][-1] <= 1.0
    if "--test_size" in argv:
        log = json.load(open(os.path.join(out, "training_log.json")))
        assert 0.5 < log["test_auc"] <= 1.0 and np.isfinite(log["test_logloss"])


def _big_vocab_model(dev, deferred, use_graph, flush_every=5, emb_dim=D):
    from deepctr.inputs import DenseFeat, SparseFeat
    from deepctr.models import xDeepFM
    from xdfm_amd import graphstep
    vocab = [5000, 31, 20003, 3, 9, 402]            # 20003 % 4 != 0: the linear table has tail rows the sweep always updates; 3 rows: a table that is all tail
    cols = [SparseFeat("C%d" % (i + 1), v, emb_dim) for i, v in enumerate(vocab)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(ND)]
    torch.manual_seed(4)
    model = xDeepFM(cols, cols, dnn_hidden_units=(32, 16), cin_layer_size=(16, 8), l2_reg_dnn=1e-5, device=dev)
    with torch.no_grad():                           # weights large enough for the L2 pull to move bits every step
        for k, p in model.named_parameters():
            if "embedding_dict" in k:
                p.mul_(2000.0)
    model.compile("adam", "binary_crossentropy", metrics=[])
    model.optim.deferred = bool(deferred)
    model.optim.flush_every = flush_every
    model.train()
    step = graphstep.GraphedStep(model)
    step.disabled = not use_graph
    model.__dict__["_graphed_step"] = step
    return model, step, vocab


def test_table_adam_auto_picks_the_update_by_table_size(monkeypatch):
    """TableAdam(deferred="auto"), the default: tables of at least optim.DEFER_MIN_NUMEL parameters in total get the
    deferred update, smaller ones the dense sweep (which is the faster way to the same bits there)."""
    _needs_default_env('arena')
    from oracle import xdeepfm_oracle as orc
    from xdfm_amd import optim
    dev = _dev()
    seen = {}
    for name, floor in (("deferred", 1000), ("sweep", 1 << 40)):
        monkeypatch.setattr(optim, "DEFER_MIN_NUMEL", floor)
        model, step, vocab = _big_vocab_model(dev, True, False)
        model.optim.deferred = "auto"
        for s in range(3):
            X, y = orc.synthetic_batch(256, vocab, ND, seed=700 + s)
            model.train_on_batch(T(X).to(dev), T(y).to(dev))
        seen[name] = model.optim.__dict__.get("_def") is not None
        model.optim.flush()
    assert seen == {"deferred": True, "sweep": False}, seen


def test_deferred_table_update_keyed_by_the_batch_rows_is_bit_identical_too(monkeypatch):
    """The step's update of the BIG tables (>= 1 M elements by default; lowered here): chunks with a gradient are enumerated
    from the batch's rows instead of by scanning the mark bytes (xdfm_adam_apply_rows); small tables, where an id occurs
    hundreds of times per batch, stay with the scan."""
    _needs_default_env('arena')
    monkeypatch.setenv("XDFM_ADAM_ROWS_MIN_NUMEL", "30000")      # the 5000- and 20003-row tables by rows, the small ones by the scan
    test_deferred_table_update_is_bit_identical_to_the_dense_sweep(True, 10, expect_path="rows")


@pytest.mark.parametrize("use_graph,emb_dim", [(False, D), (True, D), (True, 10)], ids=["eager", "graph", "graph-D10"])
def test_deferred_table_update_is_bit_identical_to_the_dense_sweep(use_graph, emb_dim, expect_path="scan"):
    """K7d (include/xdfm.h): rows are updated when gathered / when a gradient arrives / every `flush_every` steps instead
    of every step.  Parameters, both moments and the step counters must equal the dense sweep's BIT FOR BIT -- after 23
    steps with cold and hot rows, a learning-rate change, a prediction in the middle (flush), flushes at steps that are
    not multiples of the period -- and the epoch's loss (data + L2 value incl. the backlog of replayed steps) must
    agree to fp32 summation noise.  Eager launches and HIP-graph replay."""
    _needs_default_env('arena')
    from oracle import xdeepfm_oracle as orc
    dev = _dev()

    def run(deferred):
        model, step, vocab = _big_vocab_model(dev, deferred, use_graph, emb_dim=emb_dim)   # D = 10 (the scripts' default): rows straddle 16-byte chunks
        total = 0.0
        for s in range(23):
            if s == 9:
                for g in model.optim.param_groups:
                    g["lr"] = 3e-3
            X, y = orc.synthetic_batch(256, vocab, ND, seed=500 + s)
            out = model.train_on_batch(T(X).to(dev), T(y).to(dev))
            total += float(out[2])
            if s == 12:
                model.eval()
                with torch.no_grad():
                    pred = model(T(X).to(dev)).clone()
                model.train()
        if hasattr(model.optim, "take_backlog"):
            model.optim.flush()
            total += model.optim.take_backlog()
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        moments = [(model.optim.state[p]["exp_avg"].clone(), model.optim.state[p]["exp_avg_sq"].clone(),
                    float(model.optim.state[p]["step"])) for p in model.optim.param_groups[0]["params"]]
        return model, step, sd, moments, total, pred

    m_d, st_d, sd_d, mo_d, tot_d, pred_d = run(False)
    m_l, st_l, sd_l, mo_l, tot_l, pred_l = run(True)
    assert m_l.optim._def is not None and m_d.optim._def is None
    assert m_l.optim.path_counts[expect_path] >= 2 and len(m_l.optim._def["tensors"]) == 12      # 6 embedding + 6 linear tables
    if use_graph:
        assert st_l.replays >= 15 and not st_l.disabled
    assert torch.equal(pred_d, pred_l)
    for k in sd_d:
        assert torch.equal(sd_d[k], sd_l[k]), k
    for (a, b, sa), (c, d, sb) in zip(mo_d, mo_l):
        assert torch.equal(a, c) and torch.equal(b, d) and sa == sb
    assert abs(tot_d - tot_l) <= 2e-6 * abs(tot_d), (tot_d, tot_l)


def test_deferred_table_update_with_two_param_groups_is_bit_identical_and_stays_deferred():
    """with a second param group (here the DNN at its own learning rate) the deferred tables of group 0 must
    neither be flushed by that group's step nor have their clock ticked twice per step -- the bias corrections of the
    replayed steps would be one step off.  Bits equal to `deferred=False`, and the deferral survives (one flush per
    `flush_every` steps, not one per step)."""
    _needs_default_env('arena')
    from oracle import xdeepfm_oracle as orc
    from xdfm_amd import graphstep
    from xdfm_amd.optim import TableAdam
    dev = _dev()

    def run(deferred):
        model, _, vocab = _big_vocab_model(dev, deferred, False)
        dnn = [p for k, p in model.named_parameters() if k.startswith("dnn.")]
        rest = [p for k, p in model.named_parameters() if not k.startswith("dnn.")]
        opt = TableAdam([{"params": rest}, {"params": dnn, "lr": 2e-3}], deferred=bool(deferred), flush_every=5)
        model.compile(opt, "binary_crossentropy", metrics=[])
        model.optim.deferred = bool(deferred)
        model.optim.flush_every = 5
        model.train()
        step = graphstep.GraphedStep(model)
        step.disabled = True
        model.__dict__["_graphed_step"] = step
        flushes = [0]
        real_flush = model.optim.flush

        def counting_flush():
            if model.optim.__dict__.get("_def") is not None and model.optim._since:
                flushes[0] += 1
            return real_flush()
        model.optim.flush = counting_flush
        for s in range(13):
            X, y = orc.synthetic_batch(256, vocab, ND, seed=800 + s)
            model.train_on_batch(T(X).to(dev), T(y).to(dev))
        n_flush = flushes[0]
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        mo = [(model.optim.state[p]["exp_avg"].clone(), model.optim.state[p]["exp_avg_sq"].clone(), float(model.optim.state[p]["step"]))
              for g in model.optim.param_groups for p in g["params"]]
        return model, sd, mo, n_flush

    m_d, sd_d, mo_d, _ = run(False)
    m_l, sd_l, mo_l, n_flush = run(True)
    assert len(m_l.optim.param_groups) == 2 and m_l.optim._def is not None and m_d.optim._def is None
    assert n_flush <= 3, "13 steps at flush_every=5 flushed %d times: the second group is flushing the tables" % n_flush
    for k in sd_d:
        assert torch.equal(sd_d[k], sd_l[k]), k
    for (a, b, sa), (c, d, sb) in zip(mo_d, mo_l):
        assert torch.equal(a, c) and torch.equal(b, d) and sa == sb


def test_deferred_table_update_fit_history_and_checkpoint_match_the_dense_sweep(tmp_path):
    """`fit` with the deferred update: the History (loss incl. the L2 term, validation metrics) equals the dense sweep's,
    a checkpoint written in the middle holds current rows, and a user-driven step (dense gradients without marks) after
    deferred steps is taken densely on rows that were brought up to date first."""
    from oracle import xdeepfm_oracle as orc
    dev = _dev()

    def run(deferred):
        model, _, vocab = _big_vocab_model(dev, deferred, True, flush_every=7)
        model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
        model.optim.deferred = bool(deferred)
        model.optim.flush_every = 7
        X, y = orc.synthetic_batch(3000, vocab, ND, seed=77)
        names = list(model.feature_index.keys())
        xd = {n: X[:, model.feature_index[n][0]] for n in names}
        hist = model.fit(xd, y, batch_size=256, epochs=2, verbose=0, validation_split=0.1, shuffle=False)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        # one user-driven step: ordinary dense gradients
        model.train()
        model.optim.zero_grad()
        out = model(T(X[:64]).to(dev))
        torch.nn.functional.binary_cross_entropy(out.squeeze(), T(y[:64]).to(dev).squeeze(), reduction="sum").backward()
        model.optim.step()
        sd2 = {k: v.clone() for k, v in model.state_dict().items()}
        return hist.history, sd, sd2

    h_d, sd_d, sd2_d = run(False)
    h_l, sd_l, sd2_l = run(True)
    for k in h_d:
        np.testing.assert_allclose(h_l[k], h_d[k], rtol=2e-6, atol=1e-9, err_msg=k)
    for k in sd_d:
        assert torch.equal(sd_d[k], sd_l[k]), k
        assert torch.equal(sd2_d[k], sd2_l[k]), "after a user-driven step: " + k


def test_deferred_table_update_soak_400_replayed_steps_with_the_default_period():
    """400 graph-replayed steps at the default flush period (32) on tables of 100 k rows (most rows are never touched, hot
    rows every step), ragged last batches of an 'epoch' every 50 steps, then bit-equality with the dense sweep."""
    _needs_default_env('graph')
    _needs_default_env('arena')
    from deepctr.inputs import DenseFeat, SparseFeat
    from deepctr.models import xDeepFM
    from oracle import xdeepfm_oracle as orc
    dev = _dev()
    vocab = [100000, 57, 100003, 1000, 9, 31337]

    def run(deferred):
        cols = [SparseFeat("C%d" % (i + 1), v, 16) for i, v in enumerate(vocab)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(ND)]
        torch.manual_seed(8)
        model = xDeepFM(cols, cols, dnn_hidden_units=(32, 16), cin_layer_size=(16, 8), l2_reg_dnn=1e-5, device=dev)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if "embedding_dict" in k:
                    p.mul_(2000.0)
        model.compile("adam", "binary_crossentropy", metrics=[])
        model.optim.deferred = bool(deferred)
        model.train()
        for s in range(400):
            rows = 100 if s % 50 == 49 else 512
            X, y = orc.synthetic_batch(rows, vocab, ND, seed=9000 + s)
            model.train_on_batch(T(X).to(dev), T(y).to(dev))
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        mo = [(model.optim.state[p]["exp_avg"].clone(), model.optim.state[p]["exp_avg_sq"].clone()) for p in model.optim.param_groups[0]["params"]]
        return model, sd, mo

    m_d, sd_d, mo_d = run(False)
    m_l, sd_l, mo_l = run(True)
    assert m_l.optim._def is not None and m_l.__dict__["_graphed_step"].replays >= 300
    for k in sd_d:
        assert torch.equal(sd_d[k], sd_l[k]), k
    for (a, b), (c, d) in zip(mo_d, mo_l):
        assert torch.equal(a, c) and torch.equal(b, d)


@pytest.mark.parametrize("mode,n,what", [
    (0, 1 << 32, "sqrt: every fp32 bit pattern (compared for 0 < x < 2^62, the replay's range)"),
    (1, 300 << 24, "division by sqrt(1 - beta2^t): 2^24 numerators x (256 real steps + 44 random constants)"),
    (2, 1 << 31, "general division: 2^31 random operand pairs inside the guard + its corners"),
    (3, 1 << 22, "whole replayed steps vs adam_one on 4 M random chunks x 6 steps (zeros, denormals, huge values)"),
], ids=["sqrt", "div_const", "div", "replay"])
def test_fast_adam_replay_primitives_are_correctly_rounded(mode, n, what):
    """csrc/adam_math.h: the deferred update replays missed steps with short forms of sqrt and the two divisions (packed
    fp32, one v_rsq_f32 + one v_rcp_f32 per element, no v_div_scale / v_div_fixup).  They must give the bits of the
    reference spelling (IEEE sqrtf and division as hipcc expands them) -- checked here on the device, the square root
    exhaustively."""
    import ctypes
    from xdfm_amd import _lib
    dev = _dev()
    lib = _lib.load()
    out = torch.zeros(3, dtype=torch.int64, device=dev)
    _lib.check(lib.xdfm_adam_selftest(mode, ctypes.c_ulonglong(n), ctypes.c_ulonglong(0x5eed + mode), 1e-3, 0.9, 0.999, 1e-8,
                                      out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "adam_selftest")
    torch.cuda.synchronize()
    done, bad, case = (int(v) for v in out.tolist())
    assert done >= n // 4, (what, done)
    assert bad == 0, "%s: %d of %d cases differ from the reference spelling (one of them: 0x%x)" % (what, bad, done, case)
