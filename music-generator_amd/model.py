"""DeepJ model surface -- drop-in for the reference's model.py.

`build_models(time_steps, input_dropout, dropout)` returns `(model, time_model,
note_model)` exactly like /root/reference/model.py:128-169.  The three objects share
one flat fp32 parameter vector in HBM and duck-type the Keras `Model` methods the
reference calls: fit / predict / summary / load_weights /
save_weights / get_layer.  All arithmetic runs in libdeepj_hip.so; this file is host
logic only (batching, shuffling, callbacks, weight files, data-parallel sharding).
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import constants as K
from .constants import *  # noqa: F401,F403  (reference modules star-import constants through model)
from .engine import DeepJConfig


def primary_loss(y_true, y_pred):
    """Reference model.py:14-20 on host arrays (the training loss itself is computed on
    the GPU inside dj_train_fwd_bwd; this is for callers that evaluate predictions).
    Keras/TF1 binary_crossentropy on probabilities with the 1e-7 clip."""
    y_true = np.asarray(y_true, np.float64)
    y_pred = np.asarray(y_pred, np.float64)

    def bce(t, p):
        p = np.clip(p, 1e-7, 1 - 1e-7)
        z = np.log(p / (1 - p))
        return np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))

    played = y_true[..., 0]
    note = bce(y_true[..., 0], y_pred[..., 0]).mean(-1)
    replay = bce(y_true[..., 1], played * y_pred[..., 1] + (1 - played) * y_true[..., 1]).mean(-1)
    vol = ((y_true[..., 2] - (played * y_pred[..., 2] + (1 - played) * y_true[..., 2])) ** 2).mean(-1)
    return note + replay + vol


class KerasHDF5Error(RuntimeError):
    """The weight file is a Keras HDF5 checkpoint (the reference's out/model.h5).  Not swallowed by
    util.build_or_load: generating from random weights because a real checkpoint could not be read would be a
    silent wrong answer."""


# Tensor name of this build -> (Keras layer group, weight name) inside a weights-only HDF5 file written by the
# reference (train.py:23 ModelCheckpoint(save_weights_only=True)) from a fresh process: Keras 2 auto-names layers
# <class>_<k> in creation order (model.py:51-169), TimeDistributed wrappers own their inner layer's weights.
# Layouts are identical (Dense [in, out]; Conv1D [k, c_in, c_out]; LSTM kernel [in, 4H], recurrent_kernel [H, 4H],
# bias [4H], gate blocks i, f, c, o), so conversion is a pure rename: tools/convert_keras_h5.py.
def keras_name_map(cfg: DeepJConfig):
    m = {"style/kernel": ("style", "style/kernel:0"), "style/bias": ("style", "style/bias:0"),
         "conv/kernel": ("time_distributed_1", "time_distributed_1/kernel:0"),
         "conv/bias": ("time_distributed_1", "time_distributed_1/bias:0")}
    dense, td = 0, 2                      # time_distributed_2 = RepeatVector over the beat input (no weights)
    for axis, layers in (("time", cfg.time_axis_layers), ("note", cfg.note_axis_layers)):
        for l in range(layers):
            dense += 1
            td += 2                       # one wrapper around RepeatVector (no weights), one around the LSTM
            m["%s_dense%d/kernel" % (axis, l)] = ("dense_%d" % dense, "dense_%d/kernel:0" % dense)
            m["%s_dense%d/bias" % (axis, l)] = ("dense_%d" % dense, "dense_%d/bias:0" % dense)
            for w in ("kernel", "recurrent_kernel", "bias"):
                m["%s_lstm%d/%s" % (axis, l, w)] = ("time_distributed_%d" % td, "time_distributed_%d/%s:0" % (td, w))
    for n in ("note_dense", "volume_dense"):
        m[n + "/kernel"] = (n, n + "/kernel:0")
        m[n + "/bias"] = (n, n + "/bias:0")
    return m


class HipBackend:
    """The product backend: engines and optimizer state on one MI355X."""

    name = "hip"

    def __init__(self, device=None):
        import torch
        from . import _lib
        _lib.load()                                   # raises if libdeepj_hip.so is missing
        if not torch.cuda.is_available():
            raise _lib.DeepJError("no HIP device visible: DeepJ models need an MI355X (there is no CPU path)")
        if device is None:
            device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
        self.torch = torch
        self.device = torch.device(device)

    def tensor(self, a):
        t = self.torch
        return t.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def numpy(self, t):
        return t.detach().cpu().numpy()

    def resident(self, arrays):
        """Upload whole training arrays to HBM once (fp32) so that fit() slices batches on the device
        instead of gathering, converting and copying ~75 MB of float64 per batch on the host.  Returns a
        list of device tensors (identical arrays share one tensor), or None when the set exceeds
        DEEPJ_RESIDENT_GB (default 64; an MI355X has 288 GB)."""
        t = self.torch
        cap = float(os.environ.get("DEEPJ_RESIDENT_GB", "64")) * 2 ** 30
        uniq = {}
        for a in arrays:
            uniq.setdefault(id(a), a)
        if sum(a.size * 4 for a in uniq.values()) > cap:
            return None
        dev = {}
        for k, a in uniq.items():
            out = t.empty(a.shape, dtype=t.float32, device=self.device)
            step = max(1, (256 << 20) // max(1, a[0].size * 4))          # ~256 MB host staging at a time
            for i in range(0, a.shape[0], step):
                out[i:i + step] = t.as_tensor(np.ascontiguousarray(a[i:i + step], dtype=np.float32)).to(self.device)
            dev[k] = out
        return [dev[id(a)] for a in arrays]

    def take(self, tensors, ids):
        idx = self.torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int64)).to(self.device)
        return [x.index_select(0, idx) for x in tensors]

    def engine(self, cfg, batch, time_steps, input_dropout, dropout, kernel_flags=0):
        from .engine import Engine
        return Engine(cfg, batch, time_steps, device=self.device, input_dropout=input_dropout, dropout=dropout,
                      kernel_flags=kernel_flags)

    def dense_layer(self, engine, params, name, x):
        """get_layer(name)(x) on the device: only the `style` embedding (visualize.py:13-17) is a layer of its own."""
        if name != "style":
            raise NotImplementedError("only the 'style' layer is callable on its own")
        return self.numpy(engine.style_embedding(params, self.tensor(x)))

    def optimizer(self, n, **kw):
        from .engine import Nadam
        return Nadam(n, self.device, **kw)

    def init_params(self, cfg, seed):
        from .engine import init_params_numpy
        return init_params_numpy(cfg, seed)

    def layout(self, cfg):
        from .engine import param_layout
        return param_layout(cfg)

    def sync(self):
        self.torch.cuda.synchronize(self.device)


class History:
    def __init__(self):
        self.history = {}
        self.epoch = []


class _Shared:
    """State shared by model / time_model / note_model (weights are shared in the
    reference too: model.py:92-93,163-165)."""

    def __init__(self, cfg, backend, input_dropout, dropout, seed):
        self.cfg = cfg
        self.backend = backend
        self.input_dropout = float(input_dropout)
        self.dropout = float(dropout)
        self.layout = backend.layout(cfg)
        self.nparams = sum(int(np.prod(s)) for _, _, s in self.layout)
        self.params = backend.tensor(backend.init_params(cfg, seed))
        self.grads = None              # view of grads_ext[:nparams]
        self.grads_ext = None
        self.optimizer = None
        self.engines = {}
        self.kernel_flags = 0          # DJ_KF_* of every engine of these models (set after a cluster fault)
        self.step = 0
        self.seed = seed

    def engine(self, batch, time_steps, train):
        key = (int(batch), int(time_steps), bool(train))
        if key not in self.engines:
            pin, pdr = (self.input_dropout, self.dropout) if train else (0.0, 0.0)
            # keep at most a handful of workspaces alive (each can be GBs)
            if len(self.engines) >= 6:
                old = self.engines.pop(next(iter(self.engines)))
                if hasattr(old, "close"):
                    old.close()                     # its fault census, then its workspace: here, not in a finalizer
            kw = {"kernel_flags": self.kernel_flags} if self.kernel_flags else {}
            self.engines[key] = self.backend.engine(self.cfg, batch, time_steps, pin, pdr, **kw)
        return self.engines[key]

    def close(self):
        """Release every engine (workspace) of these models; each takes its last fault census on the way (Engine.close)."""
        for e in list(self.engines.values()):
            if hasattr(e, "close"):
                e.close()
        self.engines.clear()

    def add_kernel_flags(self, flags):
        """OR `flags` (DJ_KF_*) into every present and future engine of these models -- and nowhere else."""
        self.kernel_flags |= int(flags)
        for e in self.engines.values():
            if hasattr(e, "set_kernel_flags"):
                e.set_kernel_flags(self.kernel_flags)


class _Layer:
    """Minimal stand-in for `model.get_layer(name)` (reference visualize.py:13-17):
    callable on a host array and exposes get_weights()."""

    def __init__(self, shared, name):
        self.name = name
        self._s = shared

    def get_weights(self):
        flat = self._s.backend.numpy(self._s.params)
        out = []
        for n, off, shape in self._s.layout:
            if n.split("/")[0] == self.name:
                out.append(flat[off:off + int(np.prod(shape))].reshape(shape).copy())
        if not out:
            raise ValueError("No such layer: " + self.name)
        return out

    def __call__(self, x):
        """The layer applied to a host array, computed by the backend (HIP: dj_style_embedding)."""
        x = np.ascontiguousarray(x, np.float32)
        s = self._s
        return s.backend.dense_layer(s.engine(1, 1, train=False), s.params, self.name, x)


class Model:
    """Common Keras-like surface.  `kind` in {"model", "time", "note"}."""

    def __init__(self, shared, kind, time_steps):
        self._s = shared
        self._kind = kind
        self.time_steps = int(time_steps)
        self.stop_training = False

    # ------------------------------------------------------------------ weights
    def close(self):
        """Not part of the Keras protocol: releases the HIP workspaces of these models (shared by model / time_model /
        note_model) after their last fault census.  Optional -- an engine that is simply dropped hands its workspace to
        engine.drain_pending() -- but it is the deterministic way."""
        self._s.close()

    def count_params(self):
        return self._s.nparams

    def get_weights(self):
        flat = self._s.backend.numpy(self._s.params)
        return [flat[off:off + int(np.prod(s))].reshape(s).copy() for _, off, s in self._s.layout]

    def set_weights(self, weights):
        flat = np.concatenate([np.asarray(w, np.float32).ravel() for w in weights])
        if flat.size != self._s.nparams:
            raise ValueError("expected %d parameters, got %d" % (self._s.nparams, flat.size))
        self._s.params.copy_(self._s.backend.tensor(flat))

    def save_weights(self, path, overwrite=True):
        """Weights-only checkpoint (reference train.py:23 ModelCheckpoint(save_weights_only=True)).
        Stored as .npz with Keras tensor names and layouts (h5py is not available here)."""
        if not overwrite and os.path.exists(path):
            return
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        arrays = {n: w for (n, _, _), w in zip(self._s.layout, self.get_weights())}
        tmp = path + ".tmp.npz"
        np.savez(tmp, **arrays)
        os.replace(tmp, path)

    def load_weights(self, path):
        """Raises on any problem -- util.build_or_load (reference util.py:18-22) catches it, except for a Keras
        HDF5 file (KerasHDF5Error): h5py is not available here, convert it with tools/convert_keras_h5.py."""
        with open(path, "rb") as f:
            if f.read(8) == b"\x89HDF\r\n\x1a\n":
                raise KerasHDF5Error(
                    "%s is a Keras HDF5 checkpoint; this build reads .npz weight files with the same tensor layouts. "
                    "Convert it once with `python tools/convert_keras_h5.py %s %s` (needs h5py)."
                    % (path, path, os.path.splitext(path)[0] + ".npz"))
        with np.load(path) as z:
            ws = []
            for n, _, shape in self._s.layout:
                if n not in z:
                    raise KeyError("weight file has no tensor " + n)
                w = z[n]
                if tuple(w.shape) != tuple(shape):
                    raise ValueError("%s: expected shape %s, file has %s" % (n, shape, w.shape))
                ws.append(w)
        self.set_weights(ws)

    def get_layer(self, name):
        return _Layer(self._s, name)

    def summary(self, print_fn=print):
        """Parameter table in creation order (reference util.py:16 calls models[0].summary())."""
        print_fn("_" * 65)
        print_fn("%-38s %-16s %10s" % ("Tensor", "Shape", "Param #"))
        print_fn("=" * 65)
        for n, _, shape in self._s.layout:
            print_fn("%-38s %-16s %10d" % (n, str(tuple(shape)), int(np.prod(shape))))
        print_fn("=" * 65)
        print_fn("Total params: {:,}".format(self._s.nparams))
        print_fn("Trainable params: {:,}".format(self._s.nparams))
        print_fn("Backend: %s, compute dtype %s, time_steps %d" % (self._s.backend.name, self._s.cfg.dtype,
                                                                   self.time_steps))
        print_fn("_" * 65)

    # ------------------------------------------------------------------ inference
    def _predict_batch(self, arrays):
        s, be = self._s, self._s.backend
        B = arrays[0].shape[0]
        T = arrays[0].shape[1]
        eng = s.engine(B, T, train=False)
        t = [be.tensor(a) for a in arrays]
        if self._kind == "model":
            out = eng.predict(s.params, *t)
        elif self._kind == "time":
            out = eng.time_model_predict(s.params, *t)
        else:
            out = eng.note_model_predict(s.params, *t)
        out = be.numpy(out)
        if hasattr(eng, "raise_on_cluster_faults"):
            eng.raise_on_cluster_faults("predict")
        return out

    def predict(self, x, batch_size=32, verbose=0):
        """Keras Model.predict: inference mode, processed in chunks of `batch_size` (Keras
        default 32 -- it matters: the reference's pitch_bins feature depends on the batch
        a sample is evaluated in, SURVEY.md finding 2)."""
        arrays = [np.asarray(a) for a in (x if isinstance(x, (list, tuple)) else [x])]
        need = {"model": 4, "time": 3, "note": 3}[self._kind]
        if len(arrays) != need:
            raise ValueError("%s.predict expects %d input arrays, got %d" % (self._kind, need, len(arrays)))
        n = arrays[0].shape[0]
        outs = [self._predict_batch([a[i:i + batch_size] for a in arrays]) for i in range(0, n, batch_size)]
        return outs[0] if len(outs) == 1 else np.concatenate(outs, axis=0)


class TrainableModel(Model):
    """`model` of build_models: adds fit / train_on_batch / evaluate (reference train.py:29)."""

    def __init__(self, shared, time_steps):
        super().__init__(shared, "model", time_steps)
        self.optimizer_config = dict(lr=0.002, beta_1=0.9, beta_2=0.999, epsilon=1e-8, schedule_decay=0.004)

    def compile(self, optimizer="nadam", loss=None, **kw):
        """Kept for API parity (model.py:152): only 'nadam' + primary_loss are implemented."""
        if optimizer != "nadam":
            raise NotImplementedError("only optimizer='nadam' is implemented")

    def _dist(self):
        """torch.distributed when it is initialised with more than one rank (DEEPJ_DIST_WORLD1=1: also with ONE rank,
        so that the collective branch of the step can be exercised on a single GPU -- tests/test_dist_gpu.py)."""
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and (
                    dist.get_world_size() > 1 or os.environ.get("DEEPJ_DIST_WORLD1") == "1"):
                return dist
        except Exception:
            pass
        return None

    def train_on_batch(self, x, y=None):
        """One optimizer step on the given (global) batch; returns the mean loss.
        With torch.distributed initialised, every rank passes ITS shard and the
        gradients are summed over ranks weighted by shard size (one all-reduce)."""
        s, be = self._s, self._s.backend
        return self._train_step([np.asarray(a) for a in x],
                                np.asarray(y[0] if isinstance(y, (list, tuple)) else y) if y is not None else None)

    def _train_step(self, x, target, on_device=False, weight=None, total_weight=None, shard=None):
        """x = [notes, chosen, beat, style] (+ target) as host arrays, or as device fp32 tensors when
        `on_device` (fit() with a device-resident data set).  weight: this rank's share of the global batch
        (default: its sample count; 0 for a rank that only keeps the collective pattern alive);
        total_weight: the global batch size when the caller knows it (fit does), else it is all-reduced;
        shard = (first sample of this rank's shard within the global batch, shard stride): fit's contiguous shards.

        Data parallel, DEEPJ_DDP_EXACT=1: the step is the reference's step on the GLOBAL batch, exactly.  By default
        each rank evaluates the reference on its own shard (the pitch_bins reshape, model.py:43-49, couples the
        samples of a batch, and the dropout masks are indexed by batch rows); in exact mode every rank computes its
        part of the global pitch_bins table (dj_pitch_bins with its batch offset), the parts are all-gathered -- one
        small extra collective per step, [octave, shard, T] floats per rank -- and the shard runs against the global
        table with the global batch's dropout masks of its rows (dj_train_fwd_bwd_mb; one seed for all ranks)."""
        s, be = self._s, self._s.backend
        notes, chosen, beat, style = x
        if target is None:
            target = chosen
        B, T = notes.shape[0], notes.shape[1]
        # micro-batches (DEEPJ_MICRO_BATCH=k): a batch that is a multiple of k runs as B/k passes of k sequences
        # with gradient accumulation -- one workspace of k sequences, one optimizer step (how the scaled
        # 3x1024 model takes its global batch of 128).  EXACTLY the step on the whole batch: pitch_bins
        # (model.py:43-49 couples the samples of a batch) is taken from the whole batch's table and the dropout
        # masks are the whole batch's masks of each micro-batch's rows (dj_train_fwd_bwd_mb).
        micro = int(os.environ.get("DEEPJ_MICRO_BATCH", "0") or 0)
        parts = B // micro if (micro > 0 and B > micro and B % micro == 0 and hasattr(be, "resident")) else 1
        n = s.nparams
        if s.grads is None:
            # flat gradient + 4 tail words [weight, weight * loss, cluster faults, 0]: under data parallelism the
            # whole buffer travels in ONE all-reduce per step
            s.grads_ext = be.tensor(np.zeros(n + 4, np.float32))
            s.grads = s.grads_ext[:n]
        if s.optimizer is None:
            s.optimizer = be.optimizer(n, **self.optimizer_config)
        dist = self._dist()
        rank = dist.get_rank() if dist else 0
        world = dist.get_world_size() if dist else 1
        exact = dist is not None and os.environ.get("DEEPJ_DDP_EXACT") == "1"
        seed = (s.seed * 1000003 + s.step * (1 if exact else world) + (0 if exact else rank)) & 0xFFFFFFFF
        t = [notes, chosen, beat, style, target] if on_device else [be.tensor(a) for a in (notes, chosen, beat, style, target)]
        weight = float(B) if weight is None else float(weight)

        def global_bins(eng):
            """(global batch size, this shard's first sample in it, pitch_bins table of the global batch)."""
            import torch
            b0, per = shard if shard is not None else (rank * B, B)          # train_on_batch: equal shards, rank order
            total = int(total_weight) if total_weight is not None else world * B
            part = eng.pitch_bins(t[0].contiguous(), seed=seed, batch_offset=b0)          # [octave, B, T]
            pad = torch.zeros((part.shape[0], per, part.shape[2]), dtype=part.dtype, device=part.device)
            if weight > 0:
                pad[:, :B] = part                       # (a weight-0 rank holds a stand-in sample: nothing to add)
            parts_ = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts_, pad)
            return total, b0, torch.cat(parts_, dim=1)[:, :total].contiguous()

        def fwd_bwd():
            eng = s.engine(B // parts, T, train=True)
            if exact:
                full, off, bins = global_bins(eng)
            elif parts > 1:
                full, off, bins = B, 0, eng.pitch_bins(t[0].contiguous(), seed=seed)
            if parts == 1:
                if exact:
                    loss = eng.train_fwd_bwd(s.params, s.grads, *t, seed=seed, full_batch=full, batch_offset=off,
                                             bins_full=bins)
                else:
                    loss = eng.train_fwd_bwd(s.params, s.grads, *t, seed=seed)
            else:
                k, loss = B // parts, None
                for i in range(parts):
                    mb = [x_[i * k:(i + 1) * k].contiguous() for x_ in t]
                    li = eng.train_fwd_bwd(s.params, s.grads, *mb, seed=seed, accumulate=i > 0, full_batch=full,
                                           batch_offset=off + i * k, bins_full=bins)
                    loss = li.clone() if loss is None else loss + li     # [sum of means, faults so far (cumulative), ..]
                    if loss.numel() > 1:
                        loss[1:] = li[1:]
                loss[:1] /= parts                                # equal parts: mean of the micro-batch means
                s.grads.mul_(1.0 / parts)
            return eng, loss

        eng, loss = fwd_bwd()
        # loss = [mean loss, cluster faults of this step] on the device (dj_workspace_faults_async): ONE read-back
        has_faults = loss.numel() > 1
        if dist:
            # shard-size weighting, then ONE sum over ranks of [gradient | weight, weight * loss, faults]
            tail = s.grads_ext[n:]
            s.grads.mul_(weight)
            tail[0] = weight
            tail[1:2].copy_(loss.detach()[:1] * weight)
            if has_faults:
                tail[2:3].copy_(loss.detach()[1:2])
            else:
                tail[2] = 0.0
            dist.all_reduce(s.grads_ext)
            w_sum, wl_sum, faults = [float(v) for v in be.numpy(tail)[:3]]     # the step's one host read-back
            total = float(total_weight) if total_weight is not None else w_sum
            loss_value = wl_sum / total
            scale = 1.0 / total
            local_faults = None
        else:
            host = be.numpy(loss)                                              # the step's one host read-back
            loss_value, faults = float(host[0]), (float(host[1]) if has_faults else 0.0)
            scale = 1.0
            local_faults = faults
        if faults:
            # a cluster wait expired or a cluster was spread over several XCDs (include/deepj_hip.h): this step's
            # numbers are NaN.  Nothing has been applied yet: every rank switches ITS engines of this model family to
            # the per-tile kernel (a dj_config flag; no environment variable, no other model is touched) and the
            # step is repeated.
            from ._lib import KF_NO_CLUSTER
            if s.kernel_flags & KF_NO_CLUSTER:
                raise RuntimeError("cluster faults reported with the cluster kernel disabled")
            if hasattr(eng, "take_async_faults"):
                eng.take_async_faults(local_faults)       # reset the device-side census; the event goes to engine.FAULT_LOG
            # describe only what THIS call's census took from THIS rank's engine; a rank whose own count was zero got
            # the number through the all-reduce (an older `last_fault` would blame a wait that did not expire now)
            mine = getattr(eng, "fault_of_last_take", None)
            if mine is not None:
                from .engine import describe_fault_report
                what = " (this rank: %s)" % describe_fault_report(mine)
            else:
                what = " (reported by another rank)" if dist else ""
            print("[deepj] rank %d: %d cluster faults in the recurrent forward kernel%s: falling back to the per-tile "
                  "kernel for this model (DJ_KF_NO_CLUSTER) and repeating the step" % (rank, int(faults), what),
                  flush=True)
            s.add_kernel_flags(KF_NO_CLUSTER)
            return self._train_step(x, target, on_device, weight, total_weight, shard)
        if loss_value != loss_value:
            raise FloatingPointError("training loss is NaN at step %d" % s.step)
        s.optimizer.step(s.params, s.grads, scale)
        s.step += 1
        return loss_value

    def evaluate(self, x, y=None, batch_size=32, verbose=0):
        notes, chosen, beat, style = [np.asarray(a) for a in x]
        target = np.asarray(y[0] if isinstance(y, (list, tuple)) else y) if y is not None else chosen
        tot, n = 0.0, notes.shape[0]
        for i in range(0, n, batch_size):
            sl = slice(i, i + batch_size)
            b = notes[sl].shape[0]
            eng = self._s.engine(b, notes.shape[1], train=False)
            be = self._s.backend
            _, loss = eng.predict(self._s.params, be.tensor(notes[sl]), be.tensor(chosen[sl]), be.tensor(beat[sl]),
                                  be.tensor(style[sl]), be.tensor(target[sl]))
            tot += float(be.numpy(loss)[0]) * b
            if hasattr(eng, "raise_on_cluster_faults"):
                eng.raise_on_cluster_faults("evaluate")
        return tot / n

    def fit(self, x=None, y=None, batch_size=32, epochs=1, verbose=1, callbacks=None, shuffle=True,
            initial_epoch=0, **unused):
        """Keras-2 Model.fit semantics used by the reference (train.py:29): per epoch the
        sample order is reshuffled with np.random.shuffle, batches are consecutive slices
        (last partial batch kept), the epoch loss is the sample-weighted mean of the batch
        losses, callbacks see on_epoch_end(epoch, logs={'loss': ...}).
        Data parallel: if torch.distributed is initialised, each global batch is split into
        contiguous per-rank shards (every rank must call fit with the same data and the
        same NumPy seed)."""
        x = [np.asarray(a) for a in x]
        target = np.asarray(y[0] if isinstance(y, (list, tuple)) else y) if y is not None else x[1]
        n = x[0].shape[0]
        callbacks = list(callbacks or [])
        hist = History()
        callbacks.append(_HistoryCallback(hist))
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            if hasattr(cb, "set_params"):
                cb.set_params(dict(batch_size=batch_size, epochs=epochs, samples=n, verbose=verbose,
                                   metrics=["loss"]))
        self.stop_training = False
        _call(callbacks, "on_train_begin", {})
        dist = self._dist()
        rank = dist.get_rank() if dist else 0
        world = dist.get_world_size() if dist else 1
        index = np.arange(n)
        be = self._s.backend
        resident = be.resident(x + [target]) if hasattr(be, "resident") else None      # whole data set in HBM
        for epoch in range(initial_epoch, epochs):
            _call(callbacks, "on_epoch_begin", epoch, {})
            if shuffle:
                np.random.shuffle(index)
            t0 = time.time()
            tot, seen = 0.0, 0
            nb = (n + batch_size - 1) // batch_size
            for bi in range(nb):
                ids = index[bi * batch_size:(bi + 1) * batch_size]
                _call(callbacks, "on_batch_begin", bi, {"batch": bi, "size": len(ids)})
                weight, shard = None, None
                if world > 1:
                    per = (len(ids) + world - 1) // world
                    mine = ids[rank * per:(rank + 1) * per]
                    weight = float(len(mine))
                    shard = (rank * per if len(mine) else 0, per)     # first sample of the shard in the batch, shard stride
                    if len(mine) == 0:                    # keep the collective pattern identical: a sample with
                        mine = ids[:1]                    # weight 0 adds nothing to the gradient or the loss
                else:
                    mine = ids
                if resident is not None:
                    dev = be.take(resident, mine)
                    loss = self._train_step(dev[:4], dev[4], on_device=True, weight=weight, total_weight=len(ids),
                                            shard=shard)
                else:
                    loss = self._train_step([a[mine] for a in x], target[mine], weight=weight, total_weight=len(ids),
                                            shard=shard)
                tot += loss * len(ids)
                seen += len(ids)
                _call(callbacks, "on_batch_end", bi, {"batch": bi, "size": len(ids), "loss": loss})
                if verbose == 1 and rank == 0:
                    print("\rEpoch %d/%d  %d/%d  loss: %.4f" % (epoch + 1, epochs, seen, n, tot / seen), end="")
                if self.stop_training:
                    break
            logs = {"loss": tot / max(seen, 1)}
            if verbose and rank == 0:
                dt = time.time() - t0
                print("%sEpoch %d/%d - %.1fs - loss: %.4f - %.0f note-steps/s" % (
                    "\r" if verbose == 1 else "", epoch + 1, epochs, dt, logs["loss"],
                    seen * x[0].shape[1] * x[0].shape[2] / max(dt, 1e-9)))
            _call(callbacks, "on_epoch_end", epoch, logs)
            if self.stop_training:
                break
        _call(callbacks, "on_train_end", {})
        return hist


class _HistoryCallback:
    def __init__(self, hist):
        self.hist = hist

    def on_epoch_end(self, epoch, logs=None):
        self.hist.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.hist.history.setdefault(k, []).append(v)


def _call(callbacks, name, *args):
    for cb in callbacks:
        fn = getattr(cb, name, None)
        if fn is not None:
            fn(*args)


def build_models(time_steps=K.SEQ_LEN, input_dropout=0.2, dropout=0.5, *, config: DeepJConfig | None = None,
                 backend=None, device=None, dtype=None, seed=1234):
    """Reference model.py:128-169.  Returns (model, time_model, note_model).

    Extra keyword-only arguments (defaults reproduce the reference):
      config  -- DeepJConfig override (e.g. num_notes=128 for the BASELINE shapes)
      dtype   -- "f32" (parity mode, default) or "bf16" (throughput mode); env DEEPJ_DTYPE
      backend -- injection point for tests; the default and only product backend is HIP
      seed    -- weight initialisation seed (Keras initialisers)
    """
    if config is None:
        config = DeepJConfig(time_steps=time_steps)
    dtype = dtype or os.environ.get("DEEPJ_DTYPE") or config.dtype
    if dtype != config.dtype or time_steps != config.time_steps:
        d = dict(config.__dict__)
        d.update(dtype=dtype, time_steps=time_steps)
        config = DeepJConfig(**d)
    if backend is None:
        backend = HipBackend(device)
    shared = _Shared(config, backend, input_dropout, dropout, seed)
    model = TrainableModel(shared, time_steps)
    time_model = Model(shared, "time", time_steps)
    note_model = Model(shared, "note", 1)
    return model, time_model, note_model
