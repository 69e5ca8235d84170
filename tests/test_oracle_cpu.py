"""Self-checks of the CPU oracle (test infrastructure): the Keras/TF arithmetic it restates
is "parity unpinned" by the reference, so it is pinned against independent formulations:
finite differences in fp64, torch.optim.NAdam, explicit conv loops, the NumPy restatement
of the pitch_bins reshape quirk, and documented identities of the loss."""
import numpy as np
import torch

from oracle import deepj_oracle as O

SMALL = dict(num_notes=12, time_steps=5, time_axis_units=6, note_axis_units=5, octave_units=4, style_units=3)


def test_param_count_reference_model():
    assert O.param_count(O.OracleConfig()) == 1269476


def test_autograd_matches_finite_differences_fp64():
    cfg = O.OracleConfig(**SMALL)
    p = O.init_params(cfg, seed=3)
    batch = O.synthetic_batch(cfg, 2, seed=4)
    masks = O.make_masks(cfg, 2, 5, 0.2, 0.5)
    _, _, g = O.loss_and_grads(cfg, p, batch, masks, dtype=torch.float64)
    p64 = {k: np.asarray(v, np.float64) for k, v in p.items()}
    t64 = [torch.as_tensor(np.asarray(a, np.float64)) for a in batch]

    def loss_of(pp):
        with torch.no_grad():
            out = O.forward(cfg, O.to_torch(pp, torch.float64), *t64[:4], masks)
            return float(O.primary_loss(t64[4], out))

    rs = np.random.RandomState(0)
    for name in ["conv/kernel", "time_lstm0/recurrent_kernel", "time_dense1/kernel", "note_lstm0/kernel",
                 "note_lstm1/bias", "style/kernel", "volume_dense/kernel", "note_dense/bias"]:
        for _ in range(3):
            idx = tuple(rs.randint(0, s) for s in p64[name].shape)
            eps = 1e-6
            pp = {k: v.copy() for k, v in p64.items()}
            pp[name][idx] += eps
            up = loss_of(pp)
            pp[name][idx] -= 2 * eps
            dn = loss_of(pp)
            fd = (up - dn) / (2 * eps)
            assert abs(fd - g[name][idx]) <= 1e-6 + 1e-4 * abs(fd), (name, idx, fd, g[name][idx])


def test_nadam_matches_torch_nadam():
    n = 257
    rs = np.random.RandomState(1)
    p0 = rs.randn(n).astype(np.float64)
    st = O.NadamState()
    p = p0.copy()
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.NAdam([tp], lr=0.002, betas=(0.9, 0.999), eps=1e-8, momentum_decay=0.004)
    for _ in range(20):
        g = rs.randn(n) * 0.3
        p = O.nadam_step(p, g, st)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
    np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1e-7, atol=1e-10)


def test_pitch_bins_quirk_closed_form_and_batch_coupling():
    cfg = O.OracleConfig()
    rs = np.random.RandomState(2)
    x = rs.rand(3, 4, 48, 3).astype(np.float32)
    a = O.pitch_bins(cfg, torch.from_numpy(x)).numpy()
    np.testing.assert_array_equal(a, O.pitch_bins_closed_form(cfg, x))
    # NumPy restatement of the TF ops of model.py:45-47 (stack, sum, tile, raw reshape)
    bins = np.stack([x[:, :, i::12, 0] for i in range(12)]).sum(axis=3)
    ref = np.tile(bins, [4, 1, 1]).reshape(3, 4, 48, 1)
    np.testing.assert_allclose(a, ref, rtol=1e-6)          # same gather, different fp32 summation order
    # SURVEY finding 2: a sample's feature depends on the batch it is evaluated in
    alone = O.pitch_bins(cfg, torch.from_numpy(x[:1])).numpy()
    assert np.abs(alone - a[:1]).max() > 0
    # generalisation used for N=128 equals the reference formula whenever N % 12 == 0
    for n in (24, 20, 128):
        c = O.OracleConfig(num_notes=n)
        y = rs.rand(2, 3, n, 3).astype(np.float32)
        np.testing.assert_allclose(O.pitch_bins(c, torch.from_numpy(y)).numpy(), O.pitch_bins_closed_form(c, y),
                                   rtol=1e-6)


def test_conv_same_padding_is_11_left_12_right():
    cfg = O.OracleConfig()
    p = O.to_torch(O.init_params(cfg, 1))
    rs = np.random.RandomState(3)
    x = rs.rand(1, 2, 48, 3).astype(np.float32)
    y = O.conv_octave(p, torch.from_numpy(x)).numpy()
    w, b = p["conv/kernel"].numpy(), p["conv/bias"].numpy()
    xp = np.pad(x, ((0, 0), (0, 0), (11, 12), (0, 0)))
    ref = np.zeros_like(y)
    for n in range(48):
        ref[:, :, n] = np.einsum("btkc,kco->bto", xp[:, :, n:n + 24], w) + b
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-6)


def test_loss_identities():
    t = torch.zeros(2, 3, 4, 3)
    t[..., 0] = torch.tensor([1.0, 0.0, 1.0, 0.0])
    t[..., 1] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    t[..., 2] = torch.tensor([0.7, 0.0, 0.3, 0.0])
    perfect = t.clone()
    # perfect prediction: only the 1e-7 clip of Keras' binary_crossentropy remains
    l0 = float(O.primary_loss(t, perfect))
    assert 0 < l0 < 1e-6
    # replay / volume terms are masked where the note is not played (model.py:18-19)
    bad = perfect.clone()
    bad[..., 1][t[..., 0] == 0] = 0.9
    bad[..., 2][t[..., 0] == 0] = 5.0
    assert abs(float(O.primary_loss(t, bad)) - l0) < 1e-9


def test_dropout_hash_statistics_and_determinism():
    m = O.keep_mask(11, 4, 4096, 64, 0.5)
    assert abs(m.mean() - 0.5) < 0.01
    m2 = O.keep_mask(11, 33, 4096, 64, 0.2)
    assert abs(m2.mean() - 0.8) < 0.01
    np.testing.assert_array_equal(m, O.keep_mask(11, 4, 4096, 64, 0.5))
    assert (O.keep_mask(12, 4, 64, 64, 0.5) != m[:64]).any()
    assert O.drop_threshold(0.5) == 32768 and O.drop_threshold(0.2) == 13108
