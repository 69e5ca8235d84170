"""TEST-ONLY backend for music_generator_amd.model: the CPU oracle behind the same
interface as HipBackend, so the host logic (fit batching, shuffling, callbacks, weight
files, data-parallel sharding over gloo) can be tested without a GPU.  Never used by the
product: build_models() defaults to HipBackend and fails loudly without the HIP library."""
import numpy as np
import torch

from oracle import deepj_oracle as O


def _ocfg(cfg, T):
    return O.OracleConfig(num_notes=cfg.num_notes, time_steps=T, num_styles=cfg.num_styles,
                          notes_per_bar=cfg.notes_per_bar, octave=cfg.octave, octave_units=cfg.octave_units,
                          style_units=cfg.style_units, note_units=cfg.note_units,
                          time_axis_units=cfg.time_axis_units, note_axis_units=cfg.note_axis_units,
                          time_axis_layers=cfg.time_axis_layers, note_axis_layers=cfg.note_axis_layers,
                          recurrent_activation=cfg.recurrent_activation)


class _Engine:
    def __init__(self, cfg, batch, T, pin, pdr, kernel_flags=0):
        self.cfg, self.ocfg, self.B, self.T, self.pin, self.pdr = cfg, _ocfg(cfg, T), batch, T, pin, pdr
        self.loss = torch.zeros(2)            # [mean loss, cluster faults] like the HIP engine
        self.kernel_flags = kernel_flags
        self.inject_faults = []               # tests: fault counts reported by the next train_fwd_bwd calls

    def set_kernel_flags(self, flags):
        self.kernel_flags = int(flags)

    def take_async_faults(self, value=None):
        return int(value or 0)

    def _p(self, params):
        return O.unflatten_params(self.ocfg, params.numpy())

    def _masks(self, seed, b0, B):
        """Dropout masks of samples [b0, b0 + B): rows of the masks of a batch of b0 + B samples (the counter hash
        depends on the row index only)."""
        if not (self.pin or self.pdr):
            return None
        m = O.make_masks(self.ocfg, b0 + B, seed, self.pin, self.pdr, T=self.T)
        return {k: v[b0:b0 + B] for k, v in m.items()}

    def pitch_bins(self, notes_full, seed=0, train=True, batch_offset=0):
        x = torch.as_tensor(np.asarray(notes_full, np.float32))
        m = self._masks(seed, batch_offset, x.shape[0]) if train else None
        if m is not None and "notes" in m:
            x = x * m["notes"]
        return O.pitch_bins_table(self.ocfg, x)

    def train_fwd_bwd(self, params, grads, notes, chosen, beat, style, target, seed=0, out=None, accumulate=False,
                      full_batch=0, batch_offset=0, bins_full=None):
        B = notes.shape[0]
        if bins_full is None:
            masks, bins = self._masks(seed, 0, B), None
        else:          # a shard of a larger batch: that batch's masks and pitch_bins rows (dj_train_fwd_bwd_mb)
            masks = self._masks(seed, batch_offset, B)
            bins = O.pitch_bins_of_shard(self.ocfg, bins_full, batch_offset, B, self.ocfg.num_notes)
        loss, o, g = O.loss_and_grads(self.ocfg, self._p(params),
                                      [t.numpy() for t in (notes, chosen, beat, style, target)], masks, bins=bins)
        gf = torch.from_numpy(O.flatten_params(self.ocfg, g))
        if accumulate:
            grads.add_(gf)
        else:
            grads.copy_(gf)
        faults = float(self.inject_faults.pop(0)) if self.inject_faults else 0.0
        self.loss = torch.tensor([float("nan") if faults else loss, faults], dtype=torch.float32)
        return self.loss

    def predict(self, params, notes, chosen, beat, style, target=None):
        p = O.to_torch(self._p(params))
        with torch.no_grad():
            out = O.forward(self.ocfg, p, notes, chosen, beat, style)
            if target is not None:
                return out, torch.tensor([float(O.primary_loss(target, out))])
        return out

    def time_model_predict(self, params, notes, beat, style):
        return torch.from_numpy(O.time_model_predict(self.ocfg, self._p(params), notes.numpy(), beat.numpy(),
                                                     style.numpy()))

    def note_model_predict(self, params, feat, chosen, style):
        return torch.from_numpy(O.note_model_predict(self.ocfg, self._p(params), feat.numpy(), chosen.numpy(),
                                                     style.numpy()))


class _Nadam:
    def __init__(self, n, **kw):
        self.st = O.NadamState()
        self.kw = dict(lr=kw.get("lr", 0.002), beta1=kw.get("beta_1", 0.9), beta2=kw.get("beta_2", 0.999),
                       eps=kw.get("epsilon", 1e-8), schedule_decay=kw.get("schedule_decay", 0.004))

    def step(self, params, grads, grad_scale=1.0):
        new = O.nadam_step(params.numpy().copy(), grads.numpy() * np.float32(grad_scale), self.st, **self.kw)
        params.copy_(torch.from_numpy(new))


class OracleBackend:
    name = "oracle(test-only)"

    def __init__(self):
        self.device = torch.device("cpu")

    def tensor(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).clone()

    def numpy(self, t):
        return t.detach().numpy()

    def engine(self, cfg, batch, T, pin, pdr, kernel_flags=0):
        return _Engine(cfg, batch, T, pin, pdr, kernel_flags)

    def dense_layer(self, engine, params, name, x):
        p = O.unflatten_params(engine.ocfg, params.numpy())
        return np.asarray(x, np.float32) @ p[name + "/kernel"] + p[name + "/bias"]

    def optimizer(self, n, **kw):
        return _Nadam(n, **kw)

    def init_params(self, cfg, seed):
        oc = _ocfg(cfg, cfg.time_steps)
        return O.flatten_params(oc, O.init_params(oc, seed))

    def layout(self, cfg):
        oc = _ocfg(cfg, cfg.time_steps)
        out, off = [], 0
        for n, s in O.param_layout(oc):
            out.append((n, off, tuple(s)))
            off += int(np.prod(s))
        return out

    def sync(self):
        pass
