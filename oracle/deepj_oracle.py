"""Oracle: CPU restatement of the reference's model graph, loss and optimizer.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function cites the
reference lines it follows (paths relative to /root/reference).  Semantics that
come from Keras 2.x / TF 1.x (absent here) are marked "Keras:" -- they are
restated from the library's documented behaviour: **parity unpinned**.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

# --------------------------------------------------------------------------
# Configuration (constants.py:42-77).  Defaults are the reference's constants.
# --------------------------------------------------------------------------


@dataclass(frozen=True)
class OracleConfig:
    num_notes: int = 48          # constants.py:50-56
    time_steps: int = 128        # constants.py:67 SEQ_LEN
    num_styles: int = 23         # constants.py:42
    notes_per_bar: int = 16      # constants.py:63
    octave: int = 12             # constants.py:51
    octave_units: int = 64       # constants.py:70
    style_units: int = 64        # constants.py:71
    note_units: int = 3          # constants.py:72
    time_axis_units: int = 256   # constants.py:73
    note_axis_units: int = 128   # constants.py:74
    time_axis_layers: int = 2    # constants.py:76
    note_axis_layers: int = 2    # constants.py:77
    # Keras-version switches (the reference pins no version)
    recurrent_activation: str = "hard_sigmoid"
    nadam_epsilon: float = 1e-8

    @property
    def feat_dim(self) -> int:   # model.py:61-67: pos(1)+class(12)+bins(1)+conv(64)+beat(16)
        return 1 + self.octave + 1 + self.octave_units + self.notes_per_bar

    def time_in_dim(self, l: int) -> int:
        return self.feat_dim if l == 0 else self.time_axis_units

    def note_in_dim(self, l: int) -> int:
        return (self.time_axis_units + self.note_units) if l == 0 else self.note_axis_units


def param_layout(cfg: OracleConfig):
    """(name, shape) in the reference's layer-creation order (model.py:128-169).

    Keras layouts: Dense kernel [in, out]; Conv1D kernel [k, c_in, c_out];
    LSTM kernel [in, 4H], recurrent_kernel [H, 4H], bias [4H], gate column
    blocks i, f, c, o.
    """
    Ht, Hn = cfg.time_axis_units, cfg.note_axis_units
    out = [
        ("style/kernel", (cfg.num_styles, cfg.style_units)),        # model.py:141
        ("style/bias", (cfg.style_units,)),
        ("conv/kernel", (2 * cfg.octave, cfg.note_units, cfg.octave_units)),  # model.py:56
        ("conv/bias", (cfg.octave_units,)),
    ]
    for l in range(cfg.time_axis_layers):                             # model.py:75-85
        d = cfg.time_in_dim(l)
        out += [
            (f"time_dense{l}/kernel", (cfg.style_units, d)),
            (f"time_dense{l}/bias", (d,)),
            (f"time_lstm{l}/kernel", (d, 4 * Ht)),
            (f"time_lstm{l}/recurrent_kernel", (Ht, 4 * Ht)),
            (f"time_lstm{l}/bias", (4 * Ht,)),
        ]
    for l in range(cfg.note_axis_layers):                             # model.py:108-123
        d = cfg.note_in_dim(l)
        out += [
            (f"note_dense{l}/kernel", (cfg.style_units, d)),
            (f"note_dense{l}/bias", (d,)),
            (f"note_lstm{l}/kernel", (d, 4 * Hn)),
            (f"note_lstm{l}/recurrent_kernel", (Hn, 4 * Hn)),
            (f"note_lstm{l}/bias", (4 * Hn,)),
        ]
    out += [
        ("note_dense/kernel", (Hn, 2)), ("note_dense/bias", (2,)),     # model.py:94
        ("volume_dense/kernel", (Hn, 1)), ("volume_dense/bias", (1,)),  # model.py:95
    ]
    return out


def param_count(cfg: OracleConfig) -> int:
    return sum(int(np.prod(s)) for _, s in param_layout(cfg))


def init_params(cfg: OracleConfig, seed: int = 1234) -> dict:
    """Keras: default initialisers -- glorot_uniform kernels, orthogonal LSTM
    recurrent kernels, zero biases with unit forget bias."""
    rs = np.random.RandomState(seed)
    p = {}
    for name, shape in param_layout(cfg):
        if name.endswith("/bias"):
            w = np.zeros(shape, np.float32)
            if "lstm" in name:
                h = shape[0] // 4
                w[h:2 * h] = 1.0                                       # unit_forget_bias
        elif name.endswith("recurrent_kernel"):
            a = rs.normal(0.0, 1.0, shape)
            u, _, v = np.linalg.svd(a, full_matrices=False)
            q = u if u.shape == shape else v
            w = q.reshape(shape).astype(np.float32)
        else:
            if len(shape) == 3:                                       # conv: receptive field k
                fan_in, fan_out = shape[0] * shape[1], shape[0] * shape[2]
            else:
                fan_in, fan_out = shape[0], shape[1]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w = rs.uniform(-lim, lim, shape).astype(np.float32)
        p[name] = w
    return p


def flatten_params(cfg: OracleConfig, p: dict) -> np.ndarray:
    return np.concatenate([np.asarray(p[n], np.float32).ravel() for n, _ in param_layout(cfg)])


def unflatten_params(cfg: OracleConfig, flat) -> dict:
    out, o = {}, 0
    for n, s in param_layout(cfg):
        k = int(np.prod(s))
        out[n] = np.asarray(flat[o:o + k]).reshape(s)
        o += k
    return out


# --------------------------------------------------------------------------
# Dropout masks.  TF's RNG cannot be reproduced, so parity runs either use
# dropout = 0 or these counter-hash masks, which the HIP kernels regenerate
# bit-for-bit (music-generator_amd/csrc/dj_common.h: dj_keep()).
# Keras: inverted dropout, kept values scaled by 1/(1-p)  (SURVEY 8a a1).
# --------------------------------------------------------------------------

SITES = {"notes": 1, "beat": 2, "chosen": 3, "conv": 4,
         "t_style": 16, "t_out": 32, "n_style": 48, "n_out": 64}  # + layer index


def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def drop_threshold(p: float) -> int:
    return int(math.ceil(float(np.float32(p)) * 65536.0))


def keep_mask(seed: int, site: int, rows: int, d: int, p: float) -> np.ndarray:
    """Boolean keep mask [rows, d]; one hash serves a pair of columns: with
    key = lowbias32(lowbias32(seed_lo ^ site * 0x9E3779B9) + seed_hi) and
    h = lowbias32(lowbias32(lowbias32(row) + key) + (c >> 1) * 0x9E3779B9), element (row, c) is kept iff
    (h >> 16 if c is odd else h & 0xFFFF) >= ceil(p * 2^16).  Seed and row are each hashed before they
    meet, so the masks of neighbouring seeds / sites are not shifted copies of one another."""
    with np.errstate(over="ignore"):
        key = _lowbias32(np.array([(seed & 0xFFFFFFFF) ^ ((site * 0x9E3779B9) & 0xFFFFFFFF)], dtype=np.uint32))
        key = _lowbias32(key + np.uint32((seed >> 32) & 0xFFFFFFFF))[0]
        rk = _lowbias32(_lowbias32(np.arange(rows, dtype=np.uint32)) + key)
        cols = np.arange(d, dtype=np.uint32)
        h = _lowbias32(rk[:, None] + ((cols >> np.uint32(1)) * np.uint32(0x9E3779B9))[None, :])
        bits = np.where((cols & np.uint32(1))[None, :] == 1, h >> np.uint32(16), h & np.uint32(0xFFFF))
    return bits >= np.uint32(drop_threshold(p))


def make_masks(cfg: OracleConfig, B: int, seed: int, input_dropout: float, dropout: float,
               T: int | None = None) -> dict:
    """All dropout sites of the training graph (model.py:58,80,85,116,123,136-138).
    Row index of every per-note site is the canonical (b*T+t)*N+n, channel last."""
    T = cfg.time_steps if T is None else T
    N = cfg.num_notes
    m = {}

    def mk(name, site, rows, d, p, shape):
        if p <= 0.0:
            return
        k = keep_mask(seed, site, rows, d, p).reshape(shape)
        m[name] = torch.from_numpy(k.astype(np.float32)) * np.float32(1.0 / (1.0 - np.float32(p)))

    mk("notes", SITES["notes"], B * T * N, cfg.note_units, input_dropout, (B, T, N, cfg.note_units))
    mk("beat", SITES["beat"], B * T, cfg.notes_per_bar, input_dropout, (B, T, cfg.notes_per_bar))
    mk("chosen", SITES["chosen"], B * T * N, cfg.note_units, input_dropout, (B, T, N, cfg.note_units))
    mk("conv", SITES["conv"], B * T * N, cfg.octave_units, dropout, (B, T, N, cfg.octave_units))
    for l in range(cfg.time_axis_layers):
        d = cfg.time_in_dim(l)
        mk(f"t_style{l}", SITES["t_style"] + l, B * T * N, d, dropout, (B, T, N, d))
        mk(f"t_out{l}", SITES["t_out"] + l, B * T * N, cfg.time_axis_units, dropout,
           (B, T, N, cfg.time_axis_units))
    for l in range(cfg.note_axis_layers):
        d = cfg.note_in_dim(l)
        mk(f"n_style{l}", SITES["n_style"] + l, B * T * N, d, dropout, (B, T, N, d))
        mk(f"n_out{l}", SITES["n_out"] + l, B * T * N, cfg.note_axis_units, dropout,
           (B, T, N, cfg.note_axis_units))
    return m


# --------------------------------------------------------------------------
# Graph pieces
# --------------------------------------------------------------------------


def _drop(x, masks, name):
    if masks is None or name not in masks:
        return x
    return x * masks[name].to(x.dtype)


def hard_sigmoid(x):
    """Keras: hard_sigmoid = clip(0.2 x + 0.5, 0, 1)."""
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


def pitch_bins(cfg: OracleConfig, x: torch.Tensor) -> torch.Tensor:
    """model.py:43-49, bug-compatible.  x is the (dropped-out) notes [B,T,N,3].

    Reference: stack of the 12 strided slices x[:, :, i::12, 0], summed over the
    octave axis -> [12,B,T]; tiled x NUM_OCTAVES on axis 0; then a RAW reshape to
    [B,T,N,1].  Generalisation for N % 12 != 0 (BASELINE's N=128): sum
    the ragged slices, tile ceil(N/12) times, keep the first B*T*N elements of the
    flattened array -- identical to the reference whenever N % 12 == 0.
    """
    B, T, N, _ = x.shape
    O = cfg.octave
    bins = torch.stack([x[:, :, i::O, 0].sum(dim=2) for i in range(O)], dim=0)   # [12,B,T]
    reps = -(-N // O)
    tiled = bins.repeat(reps, 1, 1).reshape(-1)[: B * T * N]
    return tiled.reshape(B, T, N, 1)


def pitch_bins_table(cfg: OracleConfig, x: torch.Tensor) -> torch.Tensor:
    """The [octave, B, T] table of model.py:45 (before the tile and the raw reshape) of the (dropped-out) notes x."""
    return torch.stack([x[:, :, i::cfg.octave, 0].sum(dim=2) for i in range(cfg.octave)], dim=0)


def pitch_bins_of_shard(cfg: OracleConfig, table: torch.Tensor, b0: int, B: int, N: int) -> torch.Tensor:
    """The pitch_bins feature [B,T,N,1] of samples [b0, b0 + B) of the batch whose table [octave, B_full, T] is given:
    rows b0.. of what pitch_bins() returns on the whole batch (tile, flatten, keep B_full*T*N, reshape: model.py:46-47).
    Checker utility for evaluations that see a shard at a time (micro-batches, data-parallel ranks)."""
    O_, B_full, T = table.shape
    reps = -(-N // O_)
    full = table.repeat(reps, 1, 1).reshape(-1)[: B_full * T * N].reshape(B_full, T, N, 1)
    return full[b0:b0 + B]


def pitch_bins_closed_form(cfg: OracleConfig, x: np.ndarray) -> np.ndarray:
    """Closed form of the same quirk, used by the HIP kernel:
    out[b,t,n] = bins[(f // (B*T)) % 12, (f % (B*T)) // T, f % T],  f = (b*T+t)*N+n."""
    B, T, N, _ = x.shape
    O = cfg.octave
    bins = np.stack([x[:, :, i::O, 0].sum(axis=2) for i in range(O)], axis=0)
    f = np.arange(B * T * N)
    out = bins[(f // (B * T)) % O, (f % (B * T)) // T, f % T]
    return out.reshape(B, T, N, 1)


def conv_octave(p, x):
    """model.py:56: TimeDistributed(Conv1D(64, 24, padding='same')) along the note axis.
    Keras/TF: cross-correlation, zero pad 11 left / 12 right for the even kernel."""
    B, T, N, C = x.shape
    w = p["conv/kernel"]                       # [k, c_in, c_out]
    k = w.shape[0]
    left = (k - 1) // 2
    right = k - 1 - left
    xi = x.reshape(B * T, N, C).transpose(1, 2)                      # [BT, C, N]
    xi = torch.nn.functional.pad(xi, (left, right))
    y = torch.nn.functional.conv1d(xi, w.permute(2, 1, 0), p["conv/bias"])  # [BT, O, N]
    return y.transpose(1, 2).reshape(B, T, N, -1)


def lstm_seq(cfg, x, W, U, b, return_state=False, h0=None, c0=None):
    """Keras LSTM(return_sequences=True) over dim 1 of x [S, L, D]; zero initial
    state; z = xW + hU + b, gate blocks i,f,c,o; i,f,o = recurrent_activation,
    g = tanh; c' = f c + i g; h' = o tanh(c')  (SURVEY 8a a9)."""
    S, L, _ = x.shape
    H = U.shape[0]
    ract = hard_sigmoid if cfg.recurrent_activation == "hard_sigmoid" else torch.sigmoid
    zx = x @ W + b
    h = x.new_zeros(S, H) if h0 is None else h0
    c = x.new_zeros(S, H) if c0 is None else c0
    hs = []
    for t in range(L):
        z = zx[:, t] + h @ U
        i = ract(z[:, :H])
        f = ract(z[:, H:2 * H])
        g = torch.tanh(z[:, 2 * H:3 * H])
        o = ract(z[:, 3 * H:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        hs.append(h)
    out = torch.stack(hs, dim=1)
    if return_state:
        return out, h, c
    return out


def style_embed(p, style_in):
    """model.py:141-142: linear Dense(STYLE_UNITS), no activation."""
    return style_in @ p["style/kernel"] + p["style/bias"]


def time_axis(cfg, p, notes, beat, style, masks=None, bins=None):
    """model.py:51-89.  notes/beat are already input-dropped.  Returns [B,T,N,Ht].  bins: the pitch_bins feature
    [B,T,N,1] when these samples are a shard of a larger batch (pitch_bins_of_shard); None = computed from `notes`."""
    B, T, N, _ = notes.shape
    dt = notes.dtype
    octave = torch.tanh(conv_octave(p, notes))                        # model.py:56-57
    octave = _drop(octave, masks, "conv")                             # model.py:58
    pos = (torch.arange(N, dtype=torch.float32) / N).to(dt)            # model.py:22-30
    pos = pos.view(1, 1, N, 1).expand(B, T, N, 1)
    cls = torch.zeros(N, cfg.octave, dtype=dt)                        # model.py:32-41
    cls[torch.arange(N), torch.arange(N) % cfg.octave] = 1.0
    cls = cls.view(1, 1, N, cfg.octave).expand(B, T, N, cfg.octave)
    if bins is None:
        bins = pitch_bins(cfg, notes)                                 # model.py:43-49
    beat_r = beat.unsqueeze(2).expand(B, T, N, beat.shape[-1])         # model.py:66
    x = torch.cat([pos, cls, bins, octave, beat_r], dim=3)            # model.py:61-67
    for l in range(cfg.time_axis_layers):                              # model.py:75-85
        sp = style @ p[f"time_dense{l}/kernel"] + p[f"time_dense{l}/bias"]   # [B,T,D]
        sp = torch.tanh(sp).unsqueeze(2).expand(B, T, N, sp.shape[-1])
        sp = _drop(sp, masks, f"t_style{l}")
        x = x + sp
        D = x.shape[-1]
        xs = x.permute(0, 2, 1, 3).reshape(B * N, T, D)                # sequences over time
        hs = lstm_seq(cfg, xs, p[f"time_lstm{l}/kernel"], p[f"time_lstm{l}/recurrent_kernel"],
                      p[f"time_lstm{l}/bias"])
        x = hs.reshape(B, N, T, -1).permute(0, 2, 1, 3)
        x = _drop(x, masks, f"t_out{l}")
    return x


def note_axis(cfg, p, time_out, chosen, style, masks=None):
    """model.py:91-126.  chosen is already input-dropped.  Returns [B,T,N,3]."""
    B, T, N, _ = time_out.shape
    shift = torch.nn.functional.pad(chosen[:, :, :-1, :], (0, 0, 1, 0))    # model.py:101
    x = torch.cat([time_out, shift], dim=3)                                 # model.py:106
    for l in range(cfg.note_axis_layers):
        sp = style @ p[f"note_dense{l}/kernel"] + p[f"note_dense{l}/bias"]
        sp = torch.tanh(sp).unsqueeze(2).expand(B, T, N, sp.shape[-1])
        sp = _drop(sp, masks, f"n_style{l}")
        x = x + sp
        D = x.shape[-1]
        hs = lstm_seq(cfg, x.reshape(B * T, N, D), p[f"note_lstm{l}/kernel"],
                      p[f"note_lstm{l}/recurrent_kernel"], p[f"note_lstm{l}/bias"])
        x = hs.reshape(B, T, N, -1)
        x = _drop(x, masks, f"n_out{l}")
    pr = torch.sigmoid(x @ p["note_dense/kernel"] + p["note_dense/bias"])   # model.py:94
    vol = x @ p["volume_dense/kernel"] + p["volume_dense/bias"]              # model.py:95
    return torch.cat([pr, vol], dim=3)                                      # model.py:125


def to_torch(p: dict, dtype=torch.float32, requires_grad=False) -> dict:
    out = {}
    for k, v in p.items():
        t = torch.as_tensor(np.asarray(v)).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def forward(cfg, p, notes, chosen, beat, style_in, masks=None, return_time=False, bins=None):
    """Training graph model.py:129-151: inputs [notes, chosen, beat, style]."""
    notes_d = _drop(notes, masks, "notes")                             # model.py:136
    beat_d = _drop(beat, masks, "beat")                                # model.py:137
    chosen_d = _drop(chosen, masks, "chosen")                          # model.py:138
    style = style_embed(p, style_in)
    t_out = time_axis(cfg, p, notes_d, beat_d, style, masks, bins)
    out = note_axis(cfg, p, t_out, chosen_d, style, masks)
    if return_time:
        return out, t_out
    return out


def _bce(target, output, eps=1e-7):
    """Keras/TF1 K.binary_crossentropy(target, output) with from_logits=False:
    clip to [eps, 1-eps], logit = log(p/(1-p)), then TF's
    sigmoid_cross_entropy_with_logits: max(l,0) - l*t + log1p(exp(-|l|)), written with
    `where(l >= 0, ...)` selections like TF does, so that autodiff yields the true
    derivative sigmoid(l) - t also at l == 0 exactly  (SURVEY 8a a14)."""
    o = torch.clamp(output, eps, 1.0 - eps)
    l = torch.log(o / (1.0 - o))
    pos = l >= 0
    relu_l = torch.where(pos, l, torch.zeros_like(l))
    neg_abs = torch.where(pos, -l, l)
    return relu_l - l * target + torch.log1p(torch.exp(neg_abs))


def primary_loss(y_true, y_pred):
    """model.py:14-20 followed by Keras' mean over the remaining axes."""
    played = y_true[..., 0]
    bce_note = _bce(y_true[..., 0], y_pred[..., 0]).mean(dim=-1)
    bce_replay = _bce(y_true[..., 1], played * y_pred[..., 1] + (1 - played) * y_true[..., 1]).mean(dim=-1)
    mse = ((y_true[..., 2] - (played * y_pred[..., 2] + (1 - played) * y_true[..., 2])) ** 2).mean(dim=-1)
    return (bce_note + bce_replay + mse).mean()


def loss_and_grads(cfg, params_np: dict, batch, masks=None, dtype=torch.float32, bins=None):
    """One forward + BPTT (torch autograd over the restated graph).
    batch = (notes, chosen, beat, style_in, target) numpy arrays.
    Returns loss (float), out [B,T,N,3] numpy, grads dict of numpy."""
    p = to_torch(params_np, dtype, requires_grad=True)
    notes, chosen, beat, style_in, target = [torch.as_tensor(np.asarray(a)).to(dtype) for a in batch]
    out = forward(cfg, p, notes, chosen, beat, style_in, masks, bins=bins)
    loss = primary_loss(target, out)
    loss.backward()
    grads = {k: v.grad.detach().numpy() for k, v in p.items()}
    return float(loss.detach()), out.detach().numpy(), grads


# --------------------------------------------------------------------------
# Optimizer.  Keras: Nadam(lr=0.002, beta_1=0.9, beta_2=0.999, epsilon,
# schedule_decay=0.004) -- model.py:152 `optimizer='nadam'`.
# --------------------------------------------------------------------------


@dataclass
class NadamState:
    t: int = 0
    m_schedule: float = 1.0
    m: np.ndarray | None = None
    v: np.ndarray | None = None


def nadam_coeffs(t: int, m_schedule: float, beta1=0.9, beta2=0.999, schedule_decay=0.004):
    """Scalar schedule of Keras' Nadam.get_updates for step t (1-based)."""
    mu_t = beta1 * (1.0 - 0.5 * (0.96 ** (t * schedule_decay)))
    mu_t1 = beta1 * (1.0 - 0.5 * (0.96 ** ((t + 1) * schedule_decay)))
    ms_new = m_schedule * mu_t
    ms_next = m_schedule * mu_t * mu_t1
    return mu_t, mu_t1, ms_new, ms_next, 1.0 - beta2 ** t


def nadam_step(flat_p, flat_g, st: NadamState, lr=0.002, beta1=0.9, beta2=0.999, eps=1e-8,
               schedule_decay=0.004):
    """Returns new flat params; updates st in place.  float64 scalars, arrays in
    the dtype of flat_p."""
    if st.m is None:
        st.m = np.zeros_like(flat_p)
        st.v = np.zeros_like(flat_p)
    st.t += 1
    mu_t, mu_t1, ms_new, ms_next, bc2 = nadam_coeffs(st.t, st.m_schedule, beta1, beta2, schedule_decay)
    dt = flat_p.dtype.type
    g = flat_g
    g_prime = g / dt(1.0 - ms_new)
    st.m = dt(beta1) * st.m + dt(1.0 - beta1) * g
    m_prime = st.m / dt(1.0 - ms_next)
    st.v = dt(beta2) * st.v + dt(1.0 - beta2) * g * g
    v_prime = st.v / dt(bc2)
    m_bar = dt(1.0 - mu_t) * g_prime + dt(mu_t1) * m_prime
    st.m_schedule = ms_new
    return flat_p - dt(lr) * m_bar / (np.sqrt(v_prime) + dt(eps))


# --------------------------------------------------------------------------
# Generation sub-models (model.py:154-167), inference mode (no dropout).
# --------------------------------------------------------------------------


def time_model_predict(cfg, params_np, notes, beat, style_in, dtype=torch.float32):
    """time_model = Model([notes_in, beat_in, style_in] -> time_out) (model.py:155).
    notes [G,T,N,3], beat [G,T,16], style_in [G,T,S] -> [G,T,N,Ht] float32."""
    p = to_torch(params_np, dtype)
    with torch.no_grad():
        n, b, s = [torch.as_tensor(np.asarray(a)).to(dtype) for a in (notes, beat, style_in)]
        out = time_axis(cfg, p, n, b, style_embed(p, s))
    return out.numpy().astype(np.float32)


def note_model_predict(cfg, params_np, feat, chosen, style_in, dtype=torch.float32):
    """note_model (model.py:157-167): feat [G,1,N,Ht], chosen [G,1,N,3],
    style_in [G,1,S] -> [G,1,N,3] float32."""
    p = to_torch(params_np, dtype)
    with torch.no_grad():
        f, c, s = [torch.as_tensor(np.asarray(a)).to(dtype) for a in (feat, chosen, style_in)]
        out = note_axis(cfg, p, f, c, style_embed(p, s))
    return out.numpy().astype(np.float32)


# --------------------------------------------------------------------------
# Synthetic batches -- the same generator the GPU bench uses.
# --------------------------------------------------------------------------


def synthetic_batch(cfg: OracleConfig, B: int, seed: int = 0, T: int | None = None):
    T = cfg.time_steps if T is None else T
    N = cfg.num_notes
    rs = np.random.RandomState(seed)
    play = (rs.random_sample((B, T + 1, N)) < 0.05)
    replay = play & (rs.random_sample((B, T + 1, N)) < 0.2)
    vol = play * rs.uniform(0.2, 1.0, (B, T + 1, N))
    roll = np.stack([play, replay, vol], axis=-1).astype(np.float32)
    notes = roll[:, :T]
    target = roll[:, 1:]
    t0 = rs.randint(0, cfg.notes_per_bar, size=B)
    beat = np.zeros((B, T, cfg.notes_per_bar), np.float32)
    tt = (t0[:, None] + np.arange(T)[None, :]) % cfg.notes_per_bar
    beat[np.arange(B)[:, None], np.arange(T)[None, :], tt] = 1.0
    style = np.zeros((B, T, cfg.num_styles), np.float32)
    style[np.arange(B), :, np.arange(B) % cfg.num_styles] = 1.0
    return notes, target.copy(), beat, style, target
