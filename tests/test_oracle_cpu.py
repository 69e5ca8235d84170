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


def test_dropout_masks_of_neighbouring_seeds_and_sites_are_not_shifted_copies():
    """Product seeds are consecutive integers (step * world + rank) and sites are small integers: the mask of
    seed s+1 (or site k+1) must not be the mask of seed s slid along the rows.  Agreement of two independent
    p = 0.5 masks is 0.5 +- 3 sigma at every row shift."""
    rows, cols, p = 4096, 64, 0.5
    sigma = 0.5 / np.sqrt((rows - 8) * cols)

    def agreement(a, b, shift):
        if shift >= 0:
            return (a[shift:] == b[:rows - shift]).mean()
        return (a[:rows + shift] == b[-shift:]).mean()

    pairs = [((s, 4), (s + 1, 4)) for s in (0, 7, 1000003, 2 ** 32 - 1, 123456789012)]
    pairs += [((5, k), (5, k + 1)) for k in (1, 2, 3, 16, 32, 48, 64)]
    pairs += [((5, 16), (5, 32)), ((9, 4), (9 + 2 ** 32, 4))]
    for (sa, ka), (sb, kb) in pairs:
        a, b = O.keep_mask(sa, ka, rows, cols, p), O.keep_mask(sb, kb, rows, cols, p)
        for shift in range(-8, 9):
            assert abs(agreement(a, b, shift) - 0.5) < 5 * sigma, ((sa, ka), (sb, kb), shift)


def test_lstm_seq_matches_torch_nn_lstm_with_sigmoid_gates():
    """Independent pin of the restated Keras cell: torch.nn.LSTM implements the same recurrence with gate order
    i, f, g, o (= Keras i, f, c, o) and sigmoid recurrent activations; with copied weights
    (weight_ih = W^T, weight_hh = U^T, bias_ih = b, bias_hh = 0) outputs and final state must agree.
    reference model.py:84,120 (Keras LSTM layers)."""
    cfg = O.OracleConfig(recurrent_activation="sigmoid")
    rs = np.random.RandomState(0)
    for S, L, D, H in [(5, 7, 9, 6), (3, 12, 94, 32)]:
        W = torch.from_numpy(rs.randn(D, 4 * H).astype(np.float64) * 0.3)
        U = torch.from_numpy(rs.randn(H, 4 * H).astype(np.float64) * 0.3)
        b = torch.from_numpy(rs.randn(4 * H).astype(np.float64) * 0.1)
        x = torch.from_numpy(rs.randn(S, L, D).astype(np.float64))
        out, h, c = O.lstm_seq(cfg, x, W, U, b, return_state=True)
        ref = torch.nn.LSTM(D, H, batch_first=True).double()
        with torch.no_grad():
            ref.weight_ih_l0.copy_(W.T)
            ref.weight_hh_l0.copy_(U.T)
            ref.bias_ih_l0.copy_(b)
            ref.bias_hh_l0.zero_()
            rout, (rh, rc) = ref(x)
        torch.testing.assert_close(out, rout, rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(h, rh[0], rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(c, rc[0], rtol=1e-10, atol=1e-12)
    # and the hard_sigmoid variant differs from it only through the gate nonlinearity: same cell with
    # clip(0.2 z + 0.5) restated step by step in NumPy
    cfg_h = O.OracleConfig()
    S, L, D, H = 4, 6, 5, 3
    W, U, b = rs.randn(D, 4 * H) * 0.5, rs.randn(H, 4 * H) * 0.5, rs.randn(4 * H) * 0.2
    x = rs.randn(S, L, D)
    out = O.lstm_seq(cfg_h, *[torch.from_numpy(a) for a in (x, W, U, b)]).numpy()
    hs = lambda z: np.clip(0.2 * z + 0.5, 0, 1)
    h, c = np.zeros((S, H)), np.zeros((S, H))
    for t in range(L):
        z = x[:, t] @ W + h @ U + b
        c = hs(z[:, H:2 * H]) * c + hs(z[:, :H]) * np.tanh(z[:, 2 * H:3 * H])
        h = hs(z[:, 3 * H:]) * np.tanh(c)
        np.testing.assert_allclose(out[:, t], h, rtol=1e-12, atol=1e-14)


def test_bce_matches_torch_binary_cross_entropy_away_from_the_clip():
    """Keras/TF1 binary_crossentropy on probabilities = -t log p - (1-t) log(1-p) wherever the 1e-7 clip is
    inactive; at the clip it saturates at -log(1e-7).  reference model.py:14-20."""
    rs = np.random.RandomState(1)
    p = torch.from_numpy(rs.uniform(1e-4, 1 - 1e-4, (64, 48)))
    t = torch.from_numpy((rs.rand(64, 48) < 0.3).astype(np.float64))
    ref = torch.nn.functional.binary_cross_entropy(p, t, reduction="none")
    torch.testing.assert_close(O._bce(t, p), ref, rtol=1e-9, atol=1e-12)
    soft = torch.from_numpy(rs.rand(64, 48))               # soft targets too (the masked replay term feeds them)
    ref = -(soft * torch.log(p) + (1 - soft) * torch.log(1 - p))
    torch.testing.assert_close(O._bce(soft, p), ref, rtol=1e-9, atol=1e-12)
    edge = O._bce(torch.tensor([1.0, 0.0]), torch.tensor([0.0, 1.0]).double())
    np.testing.assert_allclose(edge.numpy(), [-np.log(1e-7)] * 2, rtol=1e-6)
    # the gradient through the restated form is the textbook (p - t) / (p (1 - p))
    pg = p.clone().requires_grad_(True)
    O._bce(t, pg).sum().backward()
    torch.testing.assert_close(pg.grad, (p - t) / (p * (1 - p)), rtol=1e-7, atol=1e-9)
