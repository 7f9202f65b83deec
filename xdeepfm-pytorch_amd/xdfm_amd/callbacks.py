"""TensorFlow-free callbacks with the Keras semantics the reference relies on
(deepctr/callbacks.py:1-73 re-exports tf.keras EarlyStopping / History and subclasses
ModelCheckpoint; deepctr/models/basemodel.py:220-227,303-307 drives them through CallbackList).
Keras itself is not importable here, so the base-class behaviour is restated from its documented
contract (monitor / mode / patience / min_delta / baseline / restore_best_weights, `period`):
parity for these plumbing classes is unpinned by any reference test.  User callbacks only need
the duck-typed hooks (xdftrain.py:31-97)."""
import copy

import numpy as np
import torch


class Callback(object):
    def __init__(self):
        self.model = None
        self.params = None

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass
    def on_batch_begin(self, batch, logs=None): pass
    def on_batch_end(self, batch, logs=None): pass


class CallbackList(object):
    """Fans every hook out to the callbacks that implement it."""

    def __init__(self, callbacks=None):
        self.callbacks = list(callbacks or [])
        self.model = None

    def append(self, cb):
        self.callbacks.append(cb)

    def set_model(self, model):
        self.model = model
        for cb in self.callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(model)
            else:
                cb.model = model

    def set_params(self, params):
        for cb in self.callbacks:
            if hasattr(cb, "set_params"):
                cb.set_params(params)

    def _fan(self, hook, *args):
        for cb in self.callbacks:
            fn = getattr(cb, hook, None)
            if fn is not None:
                fn(*args)

    def on_train_begin(self, logs=None): self._fan("on_train_begin", logs)
    def on_train_end(self, logs=None): self._fan("on_train_end", logs)
    def on_epoch_begin(self, epoch, logs=None): self._fan("on_epoch_begin", epoch, logs)
    def on_epoch_end(self, epoch, logs=None): self._fan("on_epoch_end", epoch, logs)
    def on_batch_begin(self, batch, logs=None): self._fan("on_batch_begin", batch, logs)
    def on_batch_end(self, batch, logs=None): self._fan("on_batch_end", batch, logs)


class History(Callback):
    """`.epoch` list and `.history` dict of per-epoch log values."""

    def __init__(self):
        super().__init__()
        self.epoch, self.history = [], {}

    def on_train_begin(self, logs=None):
        self.epoch, self.history = [], {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for key, val in (logs or {}).items():
            self.history.setdefault(key, []).append(val)


def _monitor_op(mode, monitor):
    if mode not in ("auto", "min", "max"):
        mode = "auto"
    if mode == "min":
        return np.less
    if mode == "max":
        return np.greater
    return np.greater if ("acc" in monitor or "auc" in monitor or monitor.startswith("fmeasure")) else np.less


class EarlyStopping(Callback):
    """Stop when `monitor` has not improved by `min_delta` for `patience` epochs."""

    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto", baseline=None,
                 restore_best_weights=False):
        super().__init__()
        self.monitor, self.patience, self.verbose = monitor, patience, verbose
        self.baseline, self.restore_best_weights = baseline, restore_best_weights
        self.monitor_op = _monitor_op(mode, monitor)
        self.min_delta = abs(min_delta) * (1 if self.monitor_op == np.greater else -1)
        self.wait = self.stopped_epoch = 0
        self.best, self.best_weights = None, None

    def on_train_begin(self, logs=None):
        self.wait = self.stopped_epoch = 0
        if self.baseline is not None:
            self.best = self.baseline
        else:
            self.best = np.inf if self.monitor_op == np.less else -np.inf
        self.best_weights = None

    def on_epoch_end(self, epoch, logs=None):
        current = (logs or {}).get(self.monitor)
        if current is None:
            print("Early stopping conditioned on metric `%s` which is not available. Available metrics are: %s"
                  % (self.monitor, ",".join(list((logs or {}).keys()))))
            return
        if self.monitor_op(current - self.min_delta, self.best):
            self.best, self.wait = current, 0
            if self.restore_best_weights:
                self.best_weights = copy.deepcopy(self.model.state_dict())
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True
                if self.restore_best_weights and self.best_weights is not None:
                    if self.verbose > 0:
                        print("Restoring model weights from the end of the best epoch.")
                    self.model.load_state_dict(self.best_weights)

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose > 0:
            print("Epoch %05d: early stopping" % (self.stopped_epoch + 1))


class ModelCheckpoint(Callback):
    """torch.save of the state_dict (or the module) every `period` epochs, optionally only when
    `monitor` improves -- the on_epoch_end of deepctr/callbacks.py:41-73."""

    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=False,
                 mode="auto", period=1):
        super().__init__()
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.save_weights_only, self.period = save_best_only, save_weights_only, period
        self.epochs_since_last_save = 0
        self.monitor_op = _monitor_op(mode, monitor)
        self.best = np.inf if self.monitor_op == np.less else -np.inf

    def _save(self, path):
        torch.save(self.model.state_dict() if self.save_weights_only else self.model, path)

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save < self.period:
            return
        self.epochs_since_last_save = 0
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if not self.save_best_only:
            if self.verbose > 0:
                print("Epoch %05d: saving model to %s" % (epoch + 1, path))
            self._save(path)
            return
        current = logs.get(self.monitor)
        if current is None:
            print("Can save best model only with %s available, skipping." % self.monitor)
        elif self.monitor_op(current, self.best):
            if self.verbose > 0:
                print("Epoch %05d: %s improved from %0.5f to %0.5f, saving model to %s"
                      % (epoch + 1, self.monitor, self.best, current, path))
            self.best = current
            self._save(path)
        elif self.verbose > 0:
            print("Epoch %05d: %s did not improve from %0.5f" % (epoch + 1, self.monitor, self.best))
