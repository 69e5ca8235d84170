"""The Keras callbacks the reference installs (train.py:22-26), re-implemented for the
Model.fit in model.py: same constructor arguments, same decisions."""
import os
import time

import numpy as np


class Callback:
    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params


class LambdaCallback(Callback):
    def __init__(self, on_epoch_begin=None, on_epoch_end=None, on_batch_begin=None, on_batch_end=None,
                 on_train_begin=None, on_train_end=None):
        for name, fn in dict(on_epoch_begin=on_epoch_begin, on_epoch_end=on_epoch_end, on_batch_begin=on_batch_begin,
                             on_batch_end=on_batch_end, on_train_begin=on_train_begin,
                             on_train_end=on_train_end).items():
            if fn is not None:
                setattr(self, name, fn)


def _better(mode, monitor):
    if mode == "max" or (mode == "auto" and ("acc" in monitor or monitor.startswith("fmeasure"))):
        return np.greater, -np.inf
    return np.less, np.inf


class ModelCheckpoint(Callback):
    """ModelCheckpoint(MODEL_FILE, monitor='loss', save_best_only=True, save_weights_only=True)
    (train.py:23): writes the weights when the monitored value improves."""

    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=False,
                 mode="auto", period=1):
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.save_weights_only, self.period = save_best_only, save_weights_only, period
        self.op, self.best = _better(mode, monitor)
        self.since = 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.since += 1
        if self.since < self.period:
            return
        self.since = 0
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not self.op(cur, self.best):
                return
            self.best = cur
        if _rank() == 0:
            self.model.save_weights(path, overwrite=True)     # weights only: optimizer state is not saved
            if self.verbose:
                print("Epoch %05d: saving model to %s" % (epoch + 1, path))


class EarlyStopping(Callback):
    """EarlyStopping(monitor='loss', patience=5) (train.py:24)."""

    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto"):
        self.monitor, self.patience, self.verbose = monitor, patience, verbose
        self.op, self.best0 = _better(mode, monitor)
        self.min_delta = -abs(min_delta) if self.op is np.less else abs(min_delta)

    def on_train_begin(self, logs=None):
        self.wait, self.stopped_epoch, self.best = 0, 0, self.best0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.op(cur - self.min_delta, self.best):
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose:
            print("Epoch %05d: early stopping" % (self.stopped_epoch + 1))


class TensorBoard(Callback):
    """Stand-in for keras.callbacks.TensorBoard(log_dir='out/logs', histogram_freq=1)
    (train.py:25): TensorFlow is absent, so scalars go to <log_dir>/scalars.csv."""

    def __init__(self, log_dir="./logs", histogram_freq=0, **kw):
        self.log_dir = log_dir

    def on_train_begin(self, logs=None):
        if _rank() == 0:
            os.makedirs(self.log_dir, exist_ok=True)
            self._f = open(os.path.join(self.log_dir, "scalars.csv"), "a")
            self._f.write("# wall_time,epoch,tag,value\n")

    def on_epoch_end(self, epoch, logs=None):
        if _rank() == 0 and getattr(self, "_f", None):
            for k, v in (logs or {}).items():
                self._f.write("%f,%d,%s,%r\n" % (time.time(), epoch, k, float(v)))
            self._f.flush()

    def on_train_end(self, logs=None):
        if getattr(self, "_f", None):
            self._f.close()
            self._f = None


def _rank():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except Exception:
        pass
    return int(os.environ.get("RANK", "0"))
