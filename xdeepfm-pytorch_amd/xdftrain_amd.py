#!/usr/bin/env python3
"""Criteo-format training entry point on the MI355X path.

Takes the command-line flags of the reference's xdftrain.py / xdftrain_attn.py / xdftrain_pro.py
(`xdftrain.py:707-738`, `xdftrain_attn.py:716-760`, `xdftrain_pro.py:774-834`) and follows their two flows.
`--mode eval` (`xdftrain.py:302-550`): read label + I1..I13 + C1..C26, encode categories (unknown -> 0, known ->
1..N in order of first appearance), min-max scale the dense columns, build `SparseFeat` / `DenseFeat`
columns, `compile`, `fit` with optional EarlyStopping / ModelCheckpoint, `predict`, and write
`history.json`, `<model>_weights.pth`, `test_predictions.csv`, `training_log.json`, `preprocess.json`.
`--mode final` (`xdftrain.py:553-704`): preprocessors, vocabulary and model fitted on ALL rows, no validation and no
metrics; writes `<model>_full_weights.pth`, `history_full.json`, `training_log_full.json`, `preprocess.json`.
`--model xdeepfm | attn | pro` picks the script being stood in for (the shims xdftrain.py / xdftrain_attn.py /
xdftrain_pro.py next to this file preselect it together with that script's default epochs / batch sizes / out_dir).
Not reproduced: the TensorBoard event files and the joblib dump of sklearn encoders (preprocess.json carries the same
category lists and ranges as plain JSON).  The reference's
own scripts also run unchanged against the `deepctr` package next to this file (they additionally
need pandas / sklearn / tensorboard); this script has no such dependencies beyond numpy + torch and
adds `--synthetic N` (no data file needed) and multi-process launch:

    python xdftrain_amd.py --synthetic 200000 --epochs 2 --embedding_dim 16
    python -m torch.distributed.run --nproc-per-node 8 xdftrain_amd.py --data_path train.txt ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from deepctr.callbacks import EarlyStopping, ModelCheckpoint  # noqa: E402
from deepctr.inputs import DenseFeat, SparseFeat, get_feature_names  # noqa: E402
from deepctr import models  # noqa: E402
from xdfm_amd import metrics as M  # noqa: E402

SPARSE = ["C%d" % i for i in range(1, 27)]
DENSE = ["I%d" % i for i in range(1, 14)]


# per-script defaults of the reference's three entry points (xdftrain.py:707-738, xdftrain_attn.py:716-760,
# xdftrain_pro.py:774-834); the shims xdftrain.py / xdftrain_attn.py / xdftrain_pro.py next to this file select them
SCRIPT_DEFAULTS = {
    "xdeepfm": dict(out_dir="./outputs_xdeepfm", epochs=3, batch_size=4096, pred_batch_size=8192),
    "attn": dict(out_dir="./outputs_xdeepfm_attn", epochs=50, batch_size=4096, pred_batch_size=8192),
    "pro": dict(out_dir="./outputs_xdeepfm_pro", epochs=20, batch_size=2048, pred_batch_size=4096),
    # xdftrain_v1.py:629-653: the plain model with a held-out test split (--test_size), early stopping always on
    "v1": dict(out_dir="./outputs_xdeepfm", epochs=20, batch_size=4096, pred_batch_size=8192, test_size=0.2, val_size=0.2,
               patience=2, use_early_stopping=True),
}


def parse_args(argv=None, model=None, script=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--data_path", type=str, default=None)
    p.add_argument("--eval_path", type=str, default=None)
    p.add_argument("--test_path", type=str, default=None)
    p.add_argument("--synthetic", type=int, default=0, help="generate this many Criteo-shaped rows instead of reading a file")
    p.add_argument("--out_dir", type=str, default=None)
    p.add_argument("--mode", type=str, choices=["eval", "final"], default="eval",
                   help="eval: train/validation split, metrics, best checkpoint, predictions; "
                        "final: fit the preprocessors and the model on ALL rows, no validation, no metrics")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--seed", type=int, default=2025)
    p.add_argument("--model", type=str, default=model or "xdeepfm", choices=["xdeepfm", "attn", "pro"])
    p.add_argument("--model_version", type=str, default="v1", choices=["v1", "v2"])
    p.add_argument("--embedding_dim", type=int, default=10)
    p.add_argument("--cin_layer_size", type=str, default=None, help="default: the model class's own (256,128; pro light 128,64)")
    p.add_argument("--dnn_hidden_units", type=str, default=None, help="default: the model class's own (256,256; pro light 128,64)")
    p.add_argument("--cin_num_heads", type=int, default=4)
    p.add_argument("--cin_attn_dropout", type=float, default=0.0)
    p.add_argument("--cin_use_layer_norm", action="store_true", default=True)
    p.add_argument("--cin_no_layer_norm", action="store_false", dest="cin_use_layer_norm")
    p.add_argument("--cin_use_residual", action="store_true", default=True)
    p.add_argument("--cin_no_residual", action="store_false", dest="cin_use_residual")
    p.add_argument("--cin_num_attn_layers", type=int, default=1)
    p.add_argument("--l2_reg_embedding", type=float, default=1e-5)
    p.add_argument("--l2_reg_dnn", type=float, default=1e-5)
    p.add_argument("--dnn_dropout", type=float, default=0.0)
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--optimizer", type=str, default="adam", choices=["adam", "adagrad", "sgd"])
    p.add_argument("--epochs", type=int, default=None)
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--pred_batch_size", type=int, default=None)
    p.add_argument("--val_size", type=float, default=None)
    p.add_argument("--test_size", type=float, default=None,
                   help="eval mode: hold this share of the rows out FIRST and report logloss / AUC on it at the end "
                        "(xdftrain_v1.py:323-329, :402-404); the validation split is then taken from the rest")
    p.add_argument("--use_early_stopping", action="store_true", default=None)
    p.add_argument("--patience", type=int, default=None)
    p.add_argument("--stratify", action="store_true", help="split train/validation per label class")
    p.add_argument("--verbose", type=int, default=1, choices=[0, 1, 2])
    # xdftrain_pro.py:805-832
    p.add_argument("--use_sfg", action="store_true", default=True)
    p.add_argument("--no_sfg", action="store_false", dest="use_sfg")
    p.add_argument("--sfg_weight", type=float, default=0.1)
    p.add_argument("--sfg_hidden_units", type=int, nargs="+", default=[128, 64])
    p.add_argument("--sfg_dropout", type=float, default=0.1)
    p.add_argument("--sfg_positive_only", action="store_true", default=True)
    p.add_argument("--sfg_all_samples", action="store_false", dest="sfg_positive_only")
    p.add_argument("--sfg_use_label_attention", action="store_true", default=True)
    p.add_argument("--use_autodis", action="store_true", default=False)
    p.add_argument("--autodis_buckets", type=int, default=16)
    p.add_argument("--use_light_version", action="store_true")
    args = p.parse_args(argv)
    for k, v in SCRIPT_DEFAULTS[script or args.model].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    for k, v in dict(val_size=0.1, patience=50, use_early_stopping=False).items():     # xdftrain.py / _attn / _pro
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


def read_table(path, with_label=True):
    """label, I1..I13, C1..C26 separated by tab or comma, optional header row; '' -> missing."""
    names = (["label"] if with_label else []) + DENSE + SPARSE
    with open(path, "r", encoding="utf-8", errors="ignore") as f:
        first = f.readline()
        second = f.readline()
    sep = "\t" if "\t" in (second or first) else ","
    has_header = not first.split("\t" if "\t" in first else ",")[0].strip().lstrip("-").replace(".", "").isdigit()
    cols = {n: [] for n in names}
    with open(path, "r", encoding="utf-8", errors="ignore") as f:
        if has_header:
            f.readline()
        for line in f:
            parts = line.rstrip("\n").split(sep)
            if len(parts) < len(names):
                parts += [""] * (len(names) - len(parts))
            for n, v in zip(names, parts):
                cols[n].append(v)
    out = {}
    for n in names:
        if n in SPARSE:
            out[n] = np.array([v if v != "" else "-1" for v in cols[n]], dtype=object)
        else:
            out[n] = np.array([float(v) if v != "" else 0.0 for v in cols[n]], dtype=np.float64)
    return out


def synthetic_table(n, seed):
    """Criteo-shaped rows whose label depends on a few categorical ids and dense values (learnable)."""
    rng = np.random.default_rng(seed)
    out, logit = {}, np.full(n, -1.6)
    for k, name in enumerate(DENSE):
        out[name] = np.floor(rng.gamma(1.0 + 0.3 * k, 20.0, n))
        if k < 4:
            logit += 0.25 * rng.normal() * np.log1p(out[name]) / 3.0
    for k, name in enumerate(SPARSE):
        vocab = 50 + 4000 * (k % 7)
        ids = np.floor(vocab * rng.random(n) ** 3).astype(np.int64)
        out[name] = np.array(["%x" % v for v in ids], dtype=object)
        if k % 7 == 0:                                   # the small-vocabulary fields carry signal
            logit += rng.normal(0.0, 0.8, vocab)[ids]
    out["label"] = (rng.random(n) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float64)
    return out


class Preprocessor(object):
    """Category ids: 0 = unseen, 1..N in order of first appearance in the training split; dense columns
    min-max scaled to [0, 1] with the training split's range (the reference's SafeLabelEncoder + MinMaxScaler)."""

    def fit(self, table):
        self.maps, self.lo, self.hi = {}, {}, {}
        for n in SPARSE:
            seen = {}
            for v in table[n]:
                if v not in seen:
                    seen[v] = len(seen) + 1
            self.maps[n] = seen
        for n in DENSE:
            self.lo[n], self.hi[n] = float(np.min(table[n])), float(np.max(table[n]))
        return self

    def transform(self, table):
        out = {}
        for n in SPARSE:
            mp = self.maps[n]
            out[n] = np.fromiter((mp.get(v, 0) for v in table[n]), dtype=np.int64, count=len(table[n]))
        for n in DENSE:
            span = self.hi[n] - self.lo[n]
            out[n] = ((table[n] - self.lo[n]) / (span if span > 0 else 1.0)).astype(np.float32)
        return out

    def vocab(self, n):
        return len(self.maps[n]) + 1


def take(table, idx):
    return {k: v[idx] for k, v in table.items()}


def split_rows(labels, val_size, seed, stratify):
    """(train indices, validation indices): a seeded permutation cut at val_size; per label class with --stratify
    (sklearn's train_test_split(stratify=y) keeps the class ratio, xdftrain.py:330-340)."""
    rng = np.random.default_rng(seed)
    n = len(labels)
    if not stratify:
        perm = rng.permutation(n)
        n_val = int(round(n * val_size))
        return perm[n_val:], perm[:n_val]
    tr, va = [], []
    for cls in np.unique(labels):
        idx = rng.permutation(np.nonzero(labels == cls)[0])
        n_val = int(round(len(idx) * val_size))
        va.append(idx[:n_val])
        tr.append(idx[n_val:])
    tr, va = np.concatenate(tr), np.concatenate(va)
    return rng.permutation(tr), rng.permutation(va)


def build_model(args, cols):
    """The model the chosen reference script builds (xdftrain.py:421-430, xdftrain_attn.py:394-425,
    xdftrain_pro.py:305-327), from this package's drop-in classes."""
    common = dict(task="binary", l2_reg_embedding=args.l2_reg_embedding, l2_reg_dnn=args.l2_reg_dnn,
                  dnn_dropout=args.dnn_dropout, device=args.device)
    if args.cin_layer_size:
        common["cin_layer_size"] = tuple(int(v) for v in args.cin_layer_size.split(","))
    if args.dnn_hidden_units:
        common["dnn_hidden_units"] = tuple(int(v) for v in args.dnn_hidden_units.split(","))
    if args.model == "xdeepfm":
        return models.xDeepFM(cols, cols, **common)
    if args.model == "attn":
        kw = dict(cin_num_heads=args.cin_num_heads, cin_attn_dropout=args.cin_attn_dropout,
                  cin_use_layer_norm=args.cin_use_layer_norm, cin_use_residual=args.cin_use_residual)
        if args.model_version == "v1":
            return models.xDeepFMAttention(cols, cols, **common, **kw)
        return models.xDeepFMAttentionV2(cols, cols, cin_num_attn_layers=args.cin_num_attn_layers, **common, **kw)
    from deepctr.xdeepfm_pro import xDeepFMPro, xDeepFMProLight
    cls = xDeepFMProLight if args.use_light_version else xDeepFMPro
    return cls(cols, cols, use_sfg=args.use_sfg, sfg_weight=args.sfg_weight, sfg_hidden_units=tuple(args.sfg_hidden_units),
               sfg_dropout=args.sfg_dropout, sfg_positive_only=args.sfg_positive_only,
               sfg_use_label_attention=args.sfg_use_label_attention, use_autodis=args.use_autodis,
               autodis_buckets=args.autodis_buckets, **common)


def main(argv=None, model=None, script=None):
    args = parse_args(argv, model, script)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        args.device = "cuda:%d" % local
        dist.init_process_group("nccl", device_id=torch.device(args.device))
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    os.makedirs(args.out_dir, exist_ok=True)
    t0 = time.time()
    final = args.mode == "final"

    if args.synthetic > 0:
        table = synthetic_table(args.synthetic, args.seed)
    elif args.data_path:
        table = read_table(args.data_path)
    else:
        raise SystemExit("give --data_path or --synthetic N")
    n = len(table["label"])
    pos = float(np.mean(table["label"] == 1))
    if rank == 0 and pos in (0.0, 1.0):
        print("[ERROR] all labels are %d: check the data file format" % int(pos))
    test_t = None
    if not final and args.test_size:                     # xdftrain_v1.py:323-329: test rows first, validation from the rest
        rest, te = split_rows(table["label"], args.test_size, args.seed, args.stratify)
        table, test_t = take(table, rest), take(table, te)
    if final:                                            # xdftrain.py:590-600: preprocessors and vocabulary from ALL rows
        train_t, val_t = table, None
    elif args.eval_path:
        train_t, val_t = table, read_table(args.eval_path)
    else:
        tr, va = split_rows(table["label"], args.val_size, args.seed, args.stratify)
        train_t, val_t = take(table, tr), take(table, va)
    prep = Preprocessor().fit(train_t)
    xtr = prep.transform(train_t)
    xva = prep.transform(val_t) if val_t is not None else None

    cols = [SparseFeat(f, vocabulary_size=prep.vocab(f), embedding_dim=args.embedding_dim) for f in SPARSE]
    cols += [DenseFeat(f, 1) for f in DENSE]
    names = get_feature_names(cols + cols)
    model = build_model(args, cols)
    # final mode compiles without metrics: a single-class batch would make AUC undefined (xdftrain.py:607-621)
    model.compile(optimizer=args.optimizer, loss="binary_crossentropy",
                  metrics=[] if final else ["binary_crossentropy", "auc"])
    for pg in model.optim.param_groups:
        pg["lr"] = args.learning_rate

    stem = {"xdeepfm": "xdeepfm", "attn": "xdeepfm_attn", "pro": "xdeepfm_pro"}[args.model]
    config = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(args).items()}
    if final:
        hist = model.fit({k: xtr[k] for k in names}, train_t["label"].reshape(-1, 1), batch_size=args.batch_size,
                         epochs=args.epochs, verbose=args.verbose, validation_split=0.0, shuffle=True)
        if rank == 0:
            history = {k: [float(v) for v in vals] for k, vals in hist.history.items()}
            torch.save(model.state_dict(), os.path.join(args.out_dir, stem + "_full_weights.pth"))
            json.dump(history, open(os.path.join(args.out_dir, "history_full.json"), "w"), indent=1)
            json.dump({"mode": "final", "training_time_seconds": time.time() - t0,
                       "data_info": {"data_path": args.data_path, "total_samples": n, "positive_ratio": pos},
                       "config": config, "history": history},
                      open(os.path.join(args.out_dir, "training_log_full.json"), "w"), indent=1)
            print("[Final] %d rows, %d epochs, %.1f s; last loss %.6f" % (
                n, args.epochs, time.time() - t0, history["loss"][-1] if history.get("loss") else float("nan")))
    else:
        best = os.path.join(args.out_dir, "best_model.pth")
        cbs = [ModelCheckpoint(best, monitor="val_auc", save_best_only=True, save_weights_only=True, mode="max", verbose=0)]
        if args.use_early_stopping:
            cbs.append(EarlyStopping(monitor="val_auc", patience=args.patience, mode="max", verbose=1))
        hist = model.fit({k: xtr[k] for k in names}, train_t["label"], batch_size=args.batch_size, epochs=args.epochs,
                         verbose=args.verbose, validation_data=({k: xva[k] for k in names}, val_t["label"]),
                         shuffle=True, callbacks=cbs)
        if rank == 0:
            if os.path.exists(best):
                model.load_state_dict(torch.load(best, weights_only=True))
            pred = model.predict({k: xva[k] for k in names}, args.pred_batch_size)
            ll, auc = M.log_loss(val_t["label"], pred), M.roc_auc_score(val_t["label"], pred)
            print("[RESULT] val logloss %.6f  val AUC %.6f  (%.1f s)" % (ll, auc, time.time() - t0))
            test_metrics = {}
            if test_t is not None:
                xts = prep.transform(test_t)
                tpred = model.predict({k: xts[k] for k in names}, args.pred_batch_size)
                test_metrics = {"test_logloss": M.log_loss(test_t["label"], tpred), "test_auc": M.roc_auc_score(test_t["label"], tpred)}
                print("[RESULT] test logloss %.6f  test AUC %.6f" % (test_metrics["test_logloss"], test_metrics["test_auc"]))
            if args.test_path:
                xte = prep.transform(read_table(args.test_path, with_label=False))
                tp = model.predict({k: xte[k] for k in names}, args.pred_batch_size)
                np.savetxt(os.path.join(args.out_dir, "test_predictions.csv"), tp, header="prediction", comments="")
            history = {k: [float(v) for v in vals] for k, vals in hist.history.items()}
            torch.save(model.state_dict(), os.path.join(args.out_dir, stem + "_weights.pth"))
            json.dump(history, open(os.path.join(args.out_dir, "history.json"), "w"), indent=1)
            json.dump({"mode": "eval", "training_time_seconds": time.time() - t0, "val_logloss": ll, "val_auc": auc, **test_metrics,
                       "data_info": {"data_path": args.data_path, "total_samples": n, "positive_ratio": pos},
                       "config": config, "history": history},
                      open(os.path.join(args.out_dir, "training_log.json"), "w"), indent=1)
    if rank == 0:
        json.dump({"vocab": {f: prep.vocab(f) for f in SPARSE}, "dense_min": prep.lo, "dense_max": prep.hi,
                   "categories": {f: list(prep.maps[f].keys()) for f in SPARSE}},
                  open(os.path.join(args.out_dir, "preprocess.json"), "w"))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
