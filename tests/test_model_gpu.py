"""Whole-path parity: the HIP training step / predict calls (through the C ABI) against
the CPU oracle on the same seeded inputs.  north_star tolerance: 1e-3 relative on the
outputs ("logits") in fp32 mode; gradients are checked at 2e-3 relative to the
per-tensor max (fp32 atomics reorder sums).  bf16 mode is checked at bf16 tolerance."""
import os
import numpy as np
import pytest
import torch

from oracle import deepj_oracle as O

pytestmark = pytest.mark.gpu


def _cfgs(**kw):
    from music_generator_amd.engine import DeepJConfig
    o = O.OracleConfig(**{k: v for k, v in kw.items() if k != "dtype"})
    d = DeepJConfig(**kw)
    return o, d


def _run_train(dcfg, B, T, flat, batch, seed, pin, pdr, dev):
    from music_generator_amd.engine import Engine
    eng = Engine(dcfg, B, T, device=dev, input_dropout=pin, dropout=pdr)
    P = torch.from_numpy(flat).to(dev)
    G = torch.empty_like(P)
    dn = [torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev) for a in batch]
    out = torch.empty((B, T, dcfg.num_notes, 3), dtype=torch.float32, device=dev)
    loss = eng.train_fwd_bwd(P, G, dn[0], dn[1], dn[2], dn[3], dn[4], seed=seed, out=out)
    torch.cuda.synchronize()
    return float(loss.cpu()[0]), out.cpu().numpy(), G.cpu().numpy(), eng


def _grad_report(ocfg, g_hip, g_ref):
    rows = []
    o = 0
    worst = 0.0
    for name, shape in O.param_layout(ocfg):
        n = int(np.prod(shape))
        a, b = g_hip[o:o + n], g_ref[name].ravel()
        scale = max(float(np.abs(b).max()), 1e-12)
        err = float(np.abs(a - b).max()) / scale
        rows.append((name, err, scale))
        worst = max(worst, err)
        o += n
    return worst, rows


# bf16 mode, worst gradient tensor relative to its max: 1.5e-2 is ~2.5-4x what these cases show (DESIGN.md "Tolerances":
# 5e-3 on the production-kernel case; the small cases print theirs) -- a regression of the 8-bit gate stash or of a
# bf16 operand path by a factor of a few fails here instead of hiding in an 8e-2 allowance
BF16_GRAD_TOL = 1.5e-2

CASES = [
    # name, cfg kwargs, B, T, input_dropout, dropout
    ("ref_dims_nodrop", dict(), 2, 8, 0.0, 0.0),
    ("ref_dims_dropout", dict(), 2, 8, 0.2, 0.5),
    ("ragged_n20", dict(num_notes=20), 3, 5, 0.2, 0.5),
    ("n128_baseline_shape", dict(num_notes=128), 1, 6, 0.0, 0.0),
    ("sigmoid_gates", dict(recurrent_activation="sigmoid", time_axis_units=128), 2, 4, 0.0, 0.0),
    ("three_layers", dict(time_axis_layers=3, note_axis_layers=1, time_axis_units=128, note_axis_units=256), 2, 4,
     0.0, 0.5),
    # widths without a persistent recurrent kernel run the per-step GEMM + gate path (dj_step.hip)
    ("scaled_3x1024", dict(time_axis_layers=3, note_axis_layers=3, time_axis_units=1024, note_axis_units=1024,
                           num_notes=12), 2, 4, 0.2, 0.5),
    ("mixed_512_96", dict(time_axis_units=512, note_axis_units=96, num_notes=24), 3, 5, 0.0, 0.5),
    ("step_time_persistent_note", dict(time_axis_units=64, note_axis_units=128, num_notes=40), 2, 6, 0.2, 0.0),
    # edges: a single sample of a single step; one octave of notes; ragged sequence tiles on both axes (B N = 792 and
    # B T = 66 sequences: 24.75 and 2.06 tiles); the reference's full window of 128 steps (constants.py:67) -- BPTT
    # through 128 recurrence steps against the oracle's autograd
    ("one_sample_one_step", dict(), 1, 1, 0.2, 0.5),
    ("one_octave_n12", dict(num_notes=12), 2, 3, 0.2, 0.5),
    ("ragged_b33", dict(num_notes=24), 33, 2, 0.2, 0.5),
    ("ref_window_t128", dict(), 2, 128, 0.2, 0.5),
]


@pytest.fixture(params=["default", "fused_xw"])
def xw_mode(request, djenv):
    """The library fuses x*W into the recurrent kernel only from 128 sequence tiles up (the BASELINE
    shape); DEEPJ_FUSE_XW_MIN_TILES=1 makes the small parity shapes take that kernel too."""
    if request.param == "fused_xw":
        djenv.set("DEEPJ_FUSE_XW_MIN_TILES", "1")
    return request.param


@pytest.mark.parametrize("name,kw,B,T,pin,pdr", CASES, ids=[c[0] for c in CASES])
def test_train_step_fp32_parity(gpu_device, xw_mode, name, kw, B, T, pin, pdr):
    ocfg, dcfg = _cfgs(time_steps=T, **kw)
    params = O.init_params(ocfg, seed=11)
    # perturb biases so that every bias gradient path is exercised with non-trivial values
    rs = np.random.RandomState(2)
    for k in params:
        if k.endswith("bias"):
            params[k] = params[k] + rs.uniform(-0.1, 0.1, params[k].shape).astype(np.float32)
    flat = O.flatten_params(ocfg, params)
    batch = O.synthetic_batch(ocfg, B, seed=3, T=T)
    seed = 99
    masks = O.make_masks(ocfg, B, seed, pin, pdr, T=T) if (pin > 0 or pdr > 0) else None
    loss_ref, out_ref, g_ref = O.loss_and_grads(ocfg, params, batch, masks)
    loss, out, g, _ = _run_train(dcfg, B, T, flat, batch, seed, pin, pdr, gpu_device)
    # outputs: 1e-3 relative (north_star), tiny atol for values near zero
    np.testing.assert_allclose(out, out_ref, rtol=1e-3, atol=1e-5)
    assert abs(loss - loss_ref) <= 1e-4 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    worst, rows = _grad_report(ocfg, g, g_ref)
    assert worst < 2e-3, sorted(rows, key=lambda r: -r[1])[:6]


@pytest.mark.parametrize("kw", [dict(), dict(time_axis_units=512, note_axis_units=512, num_notes=24)],
                         ids=["ref_dims", "step_path_512"])
def test_train_step_bf16_close(gpu_device, xw_mode, kw):
    T, B = 8, 2
    ocfg, dcfg = _cfgs(time_steps=T, dtype="bf16", **kw)
    params = O.init_params(ocfg, seed=11)
    flat = O.flatten_params(ocfg, params)
    batch = O.synthetic_batch(ocfg, B, seed=3, T=T)
    loss_ref, out_ref, g_ref = O.loss_and_grads(ocfg, params, batch, None)
    loss, out, g, _ = _run_train(dcfg, B, T, flat, batch, 1, 0.0, 0.0, gpu_device)
    np.testing.assert_allclose(out, out_ref, rtol=3e-2, atol=3e-3)     # bf16 operands: ~2^-8 relative
    assert abs(loss - loss_ref) <= 2e-2 * max(1.0, abs(loss_ref))
    worst, rows = _grad_report(ocfg, g, g_ref)
    print("bf16 step vs oracle (%s): |dloss| %.2e, worst grad tensor %.2e" % (kw or "ref dims", abs(loss - loss_ref), worst))
    assert worst < BF16_GRAD_TOL, sorted(rows, key=lambda r: -r[1])[:6]


# (name, B, N, fuse_xw_min_tiles): bf16 at the reference's window length, constants.py:67 SEQ_LEN = 128
BF16_T128 = [("b2_n48_default", 2, 48, 0), ("b2_n48_fused_xw", 2, 48, 1), ("b4_n128_cluster", 4, 128, 1)]


@pytest.mark.parametrize("name,B,N,fuse", BF16_T128, ids=[c[0] for c in BF16_T128])
def test_train_step_bf16_full_window_t128(gpu_device, djenv, name, B, N, fuse):
    """The THROUGHPUT mode over the reference's full window (constants.py:67 SEQ_LEN = 128), dropout on, against the fp32
    oracle: 128 recurrence steps of BPTT through the 8-bit activated-gate stash (|error| <= 1/508 per gate and step) with
    bf16 operands.  Until round 4 bf16 was compared with the oracle over at most 33 recurrence steps, and at T = 128 only
    with the build's own fp32 mode.  B2 x N48: the reference's dims, per-tile kernels (default) and the fused x*W
    kernels; B4 x N128: 16 time-axis tiles on the weight-stationary cluster kernel, the note axis over 128 steps too
    (N = 128).  The worst gradient tensor (relative to its max) is printed and held to the bf16 bound of the short
    cases; if it did not hold here that would be a finding about the 8-bit stash, not about the test."""
    if fuse:
        djenv.set("DEEPJ_FUSE_XW_MIN_TILES", str(fuse))
    T, seed, pin, pdr = 128, 4242, 0.2, 0.5
    ocfg, dcfg = _cfgs(time_steps=T, num_notes=N, dtype="bf16")
    params = O.init_params(ocfg, seed=11)
    rs = np.random.RandomState(2)
    for k in params:
        if k.endswith("bias"):
            params[k] = params[k] + rs.uniform(-0.1, 0.1, params[k].shape).astype(np.float32)
    flat = O.flatten_params(ocfg, params)
    batch = O.synthetic_batch(ocfg, B, seed=3, T=T)
    loss_ref, out_ref, g_ref = O.loss_and_grads(ocfg, params, batch, O.make_masks(ocfg, B, seed, pin, pdr, T=T))
    loss, out, g, eng = _run_train(dcfg, B, T, flat, batch, seed, pin, pdr, gpu_device)
    assert eng.cluster_faults() == 0
    worst, rows = _grad_report(ocfg, g, g_ref)
    top = sorted(rows, key=lambda r: -r[1])[:4]
    print("bf16 T=128 parity (%s): |dloss| %.2e, worst grad tensor %.2e (%s); next %s"
          % (name, abs(loss - loss_ref), worst, top[0][0], [(r[0], "%.1e" % r[1]) for r in top[1:]]))
    np.testing.assert_allclose(out, out_ref, rtol=3e-2, atol=3e-3)
    assert abs(loss - loss_ref) <= 2e-2 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    assert worst < BF16_GRAD_TOL, top


@pytest.mark.parametrize("kw,B,T", [(dict(time_axis_units=512, note_axis_units=64, num_notes=24), 3, 5),
                                     (dict(time_axis_layers=1, note_axis_layers=1, time_axis_units=1024, note_axis_units=96,
                                           num_notes=40, recurrent_activation="sigmoid"), 9, 7)],
                         ids=["512_64", "1024_96_sigmoid_ragged"])
def test_step_cell_epilogue_vs_gate_launches(gpu_device, kw, B, T):
    """Generic-width layers in bf16 (BASELINE configs[4]'s path), forward: ONE launch per recurrence step -- z_t =
    [x_t | h_{t-1}] [W ; U] + b with the LSTM cell as the GEMM's epilogue (round 4: gate-interleaved rows of the packed
    operand, fp32 carry in fragment layout, no x W pass, no gate launch) -- against the round-3 form (x W as one GEMM, then a
    GEMM + a gate launch per step: DJ_KF_NO_STEP_EPILOGUE), same inputs, dropout on.  The new form rounds z to bf16 once
    (the old one rounds x W + b and then the sum), so the two agree to bf16 rounding, not bit for bit; each is held to the
    oracle by the bf16 parity cases.  Widths with partial 256-column tiles (4 x 96 = 384), input widths that are not a
    multiple of the k stage (94 -> 128, 1027 -> 1088 with zero columns) and ragged sequence tiles included."""
    from music_generator_amd._lib import KF_NO_STEP_EPILOGUE
    from music_generator_amd.engine import Engine
    ocfg, dcfg = _cfgs(time_steps=T, dtype="bf16", **kw)
    flat = O.flatten_params(ocfg, O.init_params(ocfg, seed=11))
    batch = O.synthetic_batch(ocfg, B, seed=3, T=T)
    dn = [torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device) for a in batch]
    res = []
    for flags in (0, KF_NO_STEP_EPILOGUE):
        eng = Engine(dcfg, B, T, device=gpu_device, input_dropout=0.2, dropout=0.5, kernel_flags=flags)
        P = torch.from_numpy(flat).to(gpu_device)
        G = torch.empty_like(P)
        out = torch.empty((B, T, dcfg.num_notes, 3), dtype=torch.float32, device=gpu_device)
        loss = eng.train_fwd_bwd(P, G, *dn, seed=77, out=out)
        pred = eng.predict(P, dn[0], dn[1], dn[2], dn[3])
        torch.cuda.synchronize()
        res.append((float(loss.cpu()[0]), out.cpu().numpy(), G.cpu().numpy(), pred.cpu().numpy()))
    (l1, o1, g1, p1), (l0, o0, g0, p0) = res
    assert np.isfinite(o1).all() and np.isfinite(g1).all() and np.isfinite(p1).all()
    np.testing.assert_allclose(o1, o0, rtol=2e-2, atol=2e-3)          # training forward
    np.testing.assert_allclose(p1, p0, rtol=2e-2, atol=2e-3)          # inference forward (no stash written)
    assert abs(l1 - l0) <= 5e-3 * max(1.0, abs(l0))
    worst, rows = _grad_report(ocfg, g1, O.unflatten_params(ocfg, g0))
    print("step cell epilogue vs gate launches: |dloss| %.2e, worst gradient tensor differs by %.2e of its max"
          % (abs(l1 - l0), worst))
    assert worst < BF16_GRAD_TOL, sorted(rows, key=lambda r: -r[1])[:6]


def test_train_step_bf16_scaled_widths(gpu_device):
    """BASELINE configs[4]'s widths (3 x 1024 units per axis) in bf16 on a small shape: the per-step path with the
    recurrent product accumulated into the stash by the GEMM epilogue (dj_gemm_nt c_mode 3 on a row-block-strided view),
    the 16-byte gate kernels with the bias gradient summed in-kernel and the 256 x 128 tile choice, against the fp32 oracle
    at bf16 tolerance, dropout on."""
    T, B, seed, pin, pdr = 6, 6, 77, 0.2, 0.5
    kw = dict(time_axis_layers=3, note_axis_layers=3, time_axis_units=1024, note_axis_units=1024, num_notes=24)
    ocfg, dcfg = _cfgs(time_steps=T, dtype="bf16", **kw)
    params = O.init_params(ocfg, seed=11)
    flat = O.flatten_params(ocfg, params)
    batch = O.synthetic_batch(ocfg, B, seed=3, T=T)
    loss_ref, out_ref, g_ref = O.loss_and_grads(ocfg, params, batch, O.make_masks(ocfg, B, seed, pin, pdr, T=T))
    loss, out, g, _ = _run_train(dcfg, B, T, flat, batch, seed, pin, pdr, gpu_device)
    np.testing.assert_allclose(out, out_ref, rtol=3e-2, atol=3e-3)
    assert abs(loss - loss_ref) <= 2e-2 * max(1.0, abs(loss_ref))
    worst, rows = _grad_report(ocfg, g, g_ref)
    print("bf16 3x1024 step vs oracle: |dloss| %.2e, worst grad tensor %.2e" % (abs(loss - loss_ref), worst))
    assert worst < BF16_GRAD_TOL, sorted(rows, key=lambda r: -r[1])[:6]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_step_production_kernels_vs_oracle(gpu_device, djenv, dtype):
    """The kernel selection of the bench shape, as ONE forward + BPTT step against the oracle, dropout on:
    B16 x T16 x N128 gives 64 time-axis sequence tiles, so bf16 runs the weight-stationary cluster kernel
    (both time layers) feeding the glue, stash, BPTT and weight-gradient kernels; the note axis (8 tiles) is
    forced onto its fused x*W / fused dX kernels.  fp32 at the north_star tolerance, bf16 at bf16 tolerance."""
    djenv.set("DEEPJ_FUSE_XW_MIN_TILES", "1")
    B, T, seed, pin, pdr = 16, 16, 1234567, 0.2, 0.5
    ocfg, dcfg = _cfgs(time_steps=T, num_notes=128, dtype=dtype)
    params = O.init_params(ocfg, seed=11)
    rs = np.random.RandomState(2)
    for k in params:
        if k.endswith("bias"):
            params[k] = params[k] + rs.uniform(-0.1, 0.1, params[k].shape).astype(np.float32)
    flat = O.flatten_params(ocfg, params)
    batch = O.synthetic_batch(ocfg, B, seed=3, T=T)
    loss_ref, out_ref, g_ref = O.loss_and_grads(ocfg, params, batch, O.make_masks(ocfg, B, seed, pin, pdr, T=T))
    loss, out, g, eng = _run_train(dcfg, B, T, flat, batch, seed, pin, pdr, gpu_device)
    assert eng.cluster_faults() == 0
    worst, rows = _grad_report(ocfg, g, g_ref)
    if dtype == "f32":
        np.testing.assert_allclose(out, out_ref, rtol=1e-3, atol=1e-5)
        assert abs(loss - loss_ref) <= 1e-4 * max(1.0, abs(loss_ref)), (loss, loss_ref)
        assert worst < 2e-3, sorted(rows, key=lambda r: -r[1])[:6]
    else:
        np.testing.assert_allclose(out, out_ref, rtol=3e-2, atol=3e-3)
        assert abs(loss - loss_ref) <= 2e-2 * max(1.0, abs(loss_ref)), (loss, loss_ref)
        assert worst < BF16_GRAD_TOL, sorted(rows, key=lambda r: -r[1])[:6]
    print("production-kernel parity (%s): |dloss| %.2e, worst grad tensor %.2e" % (dtype, abs(loss - loss_ref), worst))


def test_predict_models_fp32(gpu_device):
    from music_generator_amd.engine import Engine
    T, G = 16, 3
    ocfg, dcfg = _cfgs(time_steps=T)
    params = O.init_params(ocfg, seed=5)
    flat = torch.from_numpy(O.flatten_params(ocfg, params)).to(gpu_device)
    notes, chosen, beat, style, target = O.synthetic_batch(ocfg, G, seed=8, T=T)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
    eng = Engine(dcfg, G, T, device=gpu_device)
    # full model, inference mode, with loss
    out, loss = eng.predict(flat, d(notes), d(chosen), d(beat), d(style), d(target))
    p = O.to_torch(params)
    with torch.no_grad():
        ref = O.forward(ocfg, p, *[torch.from_numpy(a) for a in (notes, chosen, beat, style)])
        lref = float(O.primary_loss(torch.from_numpy(target), ref))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-3, atol=1e-5)
    assert abs(float(loss.cpu()[0]) - lref) < 1e-4 * max(1.0, lref)
    # time_model (generate.py:108)
    tout = eng.time_model_predict(flat, d(notes), d(beat), d(style)).cpu().numpy()
    tref = O.time_model_predict(ocfg, params, notes, beat, style)
    np.testing.assert_allclose(tout, tref, rtol=1e-3, atol=2e-5)
    # note_model on one time step (generate.py:114)
    eng1 = Engine(dcfg, G, 1, device=gpu_device)
    feat = tref[:, -1:, :, :]
    ch = chosen[:, -1:, :, :]
    st = style[:, -1:, :]
    nout = eng1.note_model_predict(flat, d(feat), d(ch), d(st)).cpu().numpy()
    nref = O.note_model_predict(ocfg, params, feat, ch, st)
    np.testing.assert_allclose(nout, nref, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("G,T,N", [(3, 128, 48), (5, 16, 40), (1, 7, 128)], ids=["gen_window", "ragged_tiles", "one_piece"])
def test_time_axis_pair_launch_matches_layer_by_layer(gpu_device, djenv, G, T, N):
    """Inference of the bf16 time axis runs both 256-unit layers as one wavefront launch (dj_lstm.hip ClPair): its
    output must be bit-identical to the layer-by-layer launches (same arithmetic, the glue folded into the lower
    layer's store), close to the fp32 oracle, and leave no cluster fault."""
    from music_generator_amd.engine import Engine
    ocfg, dcfg = _cfgs(time_steps=T, num_notes=N, dtype="bf16")
    params = O.init_params(ocfg, seed=11)
    flat = torch.from_numpy(O.flatten_params(ocfg, params)).to(gpu_device)
    notes, chosen, beat, style, target = O.synthetic_batch(ocfg, G, seed=4, T=T)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
    eng = Engine(dcfg, G, T, device=gpu_device)
    from music_generator_amd._lib import KF_NO_CLUSTER_PAIR
    djenv.unset("DEEPJ_CLUSTER_PAIR")
    pair = eng.time_model_predict(flat, d(notes), d(beat), d(style)).float().cpu().numpy()
    pair2 = eng.time_model_predict(flat, d(notes), d(beat), d(style)).float().cpu().numpy()
    eng.set_kernel_flags(KF_NO_CLUSTER_PAIR)                 # per-engine switch (dj_config.kernel_flags)
    seq = eng.time_model_predict(flat, d(notes), d(beat), d(style)).float().cpu().numpy()
    eng.set_kernel_flags(0)
    djenv.set("DEEPJ_CLUSTER_PAIR", "0")                     # the same switch as a process default
    seq_env = eng.time_model_predict(flat, d(notes), d(beat), d(style)).float().cpu().numpy()
    djenv.unset("DEEPJ_CLUSTER_PAIR")
    np.testing.assert_array_equal(seq, seq_env)
    # the cooperative pair's members exchange tagged h slices by default; the counted protocol gives the same bits
    from music_generator_amd._lib import KF_COUNTED_EXCHANGE
    djenv.unset("DEEPJ_TAGGED_EXCHANGE")
    eng.set_kernel_flags(KF_COUNTED_EXCHANGE)
    counted = eng.time_model_predict(flat, d(notes), d(beat), d(style)).float().cpu().numpy()
    eng.set_kernel_flags(0)
    np.testing.assert_array_equal(pair, counted)
    assert eng.cluster_faults() == 0
    assert np.isfinite(pair).all()
    np.testing.assert_array_equal(pair, pair2)
    np.testing.assert_array_equal(pair, seq)
    tref = O.time_model_predict(ocfg, params, notes, beat, style)
    assert np.abs(pair - tref).max() < 3e-2


@pytest.mark.parametrize("G,T,N", [(3, 128, 48), (5, 16, 40), (1, 7, 128)], ids=["gen_window", "ragged_tiles", "one_piece"])
def test_time_axis_fp32_cluster_matches_per_tile_kernel(gpu_device, djenv, G, T, N):
    """fp32 inference of the time axis with at most 8 sequence tiles runs on clusters of 8 workgroups with U resident in
    LDS (lstm_fwd_cluster_f32_kernel): bit-identical to the per-tile kernel (same sums in the same order), within the
    north_star's 1e-3 of the oracle, no cluster fault."""
    from music_generator_amd.engine import Engine
    ocfg, dcfg = _cfgs(time_steps=T, num_notes=N)
    params = O.init_params(ocfg, seed=12)
    flat = torch.from_numpy(O.flatten_params(ocfg, params)).to(gpu_device)
    notes, chosen, beat, style, target = O.synthetic_batch(ocfg, G, seed=6, T=T)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
    eng = Engine(dcfg, G, T, device=gpu_device)
    from music_generator_amd._lib import KF_NO_CLUSTER_F32
    djenv.unset("DEEPJ_CLUSTER_F32")
    cl = eng.time_model_predict(flat, d(notes), d(beat), d(style)).cpu().numpy()
    eng.set_kernel_flags(KF_NO_CLUSTER_F32)
    pt = eng.time_model_predict(flat, d(notes), d(beat), d(style)).cpu().numpy()
    eng.set_kernel_flags(0)
    assert eng.cluster_faults() == 0
    np.testing.assert_array_equal(cl, pt)
    tref = O.time_model_predict(ocfg, params, notes, beat, style)
    np.testing.assert_allclose(cl, tref, rtol=1e-3, atol=2e-5)


def test_seed_reproducible_and_mask_sensitive(gpu_device):
    T, B = 4, 2
    ocfg, dcfg = _cfgs(time_steps=T)
    flat = O.flatten_params(ocfg, O.init_params(ocfg, seed=1))
    batch = O.synthetic_batch(ocfg, B, seed=0, T=T)
    l1, o1, _, _ = _run_train(dcfg, B, T, flat, batch, 5, 0.2, 0.5, gpu_device)
    l2, o2, _, _ = _run_train(dcfg, B, T, flat, batch, 5, 0.2, 0.5, gpu_device)
    l3, o3, _, _ = _run_train(dcfg, B, T, flat, batch, 6, 0.2, 0.5, gpu_device)
    np.testing.assert_array_equal(o1, o2)
    assert np.abs(o1 - o3).max() > 1e-4


def test_keras_surface_on_hip(gpu_device, tmp_path):
    """build_models -> fit / predict / save / load / generate through the public surface
    (HIP backend), against the oracle-backed twin of the same host code."""
    from music_generator_amd.model import build_models
    from music_generator_amd.data import synthetic_batch
    from oracle_backend import OracleBackend
    T = 8
    hm = build_models(time_steps=T, input_dropout=0.0, dropout=0.0, seed=4)
    om = build_models(time_steps=T, input_dropout=0.0, dropout=0.0, seed=4, backend=OracleBackend())
    a = synthetic_batch(48, T, 4, seed=2)
    x, y = [a[0], a[1], a[2], a[3]], [a[4]]
    hh = hm[0].fit(x, y, epochs=2, batch_size=2, verbose=0, shuffle=False)
    oh = om[0].fit(x, y, epochs=2, batch_size=2, verbose=0, shuffle=False)
    np.testing.assert_allclose(hh.history["loss"], oh.history["loss"], rtol=2e-4)
    for wa, wb in zip(hm[0].get_weights(), om[0].get_weights()):
        np.testing.assert_allclose(wa, wb, rtol=0, atol=2e-4)        # 4 Nadam steps of lr 2e-3
    # predict of all three models on the trained HIP weights vs the oracle on the SAME weights
    om[0].set_weights(hm[0].get_weights())
    np.testing.assert_allclose(hm[0].predict(x), om[0].predict(x), rtol=1e-3, atol=1e-5)
    tf = hm[1].predict([x[0], x[2], x[3]])
    np.testing.assert_allclose(tf, om[1].predict([x[0], x[2], x[3]]), rtol=1e-3, atol=2e-5)
    feat, ch, st = tf[:, -1:], x[1][:, -1:], x[3][:, -1]
    np.testing.assert_allclose(hm[2].predict([feat, ch, st[:, None]]), om[2].predict([feat, ch, st[:, None]]),
                               rtol=1e-3, atol=1e-5)
    assert abs(hm[0].evaluate(x, y) - om[0].evaluate(x, y)) < 1e-4
    # f-4: the style-embedding export on the trained HIP weights (device kernel) vs the oracle twin
    from music_generator_amd import visualize
    np.testing.assert_allclose(visualize.style_embeddings(hm), visualize.style_embeddings(om), rtol=1e-5, atol=1e-6)
    ck = str(tmp_path / "m.npz")
    hm[0].save_weights(ck)
    hm2 = build_models(time_steps=T, seed=9)
    hm2[0].load_weights(ck)
    np.testing.assert_array_equal(hm2[0].get_weights()[5], hm[0].get_weights()[5])


@pytest.mark.fault_injection
def test_expired_wait_is_counted_described_and_costs_one_bound(gpu_device, djenv):
    """DEEPJ_DEBUG_CLUSTER_LATE: the last member of every cluster never arrives in round 0 of the exchange, so every other
    wave's bound REALLY runs out (2^19 polls, ~0.1 s of polling).  The launch must then (a) count it and describe
    the first expired wait in the fault line -- kernel, cluster, member, wave, step -1, counter 7 of 8, the polls made;
    (b) poison the tiles (NaN loss); (c) cost ONE bound per launch, not one per remaining step: the first wave to give
    up sets the poison bit in the cluster's counter, which releases every other waiter, and no poisoned wave waits again
    (until round 4 every one of the 128 steps of a faulted launch waited the full bound again: 2 layers x 128 x 0.1 s).
    B16 x T128 x N128 in bf16 = 64 time-axis tiles = 8 clusters, the weight-stationary sweep of both time layers."""
    import time
    from music_generator_amd import engine as E
    from music_generator_amd.engine import Engine
    djenv.set("DEEPJ_DEBUG_CLUSTER_LATE", "1")
    B, T = 16, 128
    ocfg, dcfg = _cfgs(time_steps=T, num_notes=128, dtype="bf16")
    flat = O.flatten_params(ocfg, O.init_params(ocfg, seed=11))
    batch = O.synthetic_batch(ocfg, B, seed=3, T=T)
    eng = Engine(dcfg, B, T, device=gpu_device, input_dropout=0.2, dropout=0.5)
    P = torch.from_numpy(flat).to(gpu_device)
    G = torch.empty_like(P)
    dn = [torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device) for a in batch]
    torch.cuda.synchronize()
    n0, t0 = len(E.FAULT_LOG), time.time()
    loss = eng.train_fwd_bwd(P, G, *dn, seed=5)
    host = loss.cpu().numpy()
    dt = time.time() - t0
    rep = eng.cluster_fault_report()
    print("expired-wait launch: %.3f s, [loss, faults, expired, misplaced] = %s, report %s" % (dt, host, rep))
    assert np.isnan(host[0]) and host[1] >= 1 and host[2] >= 1 and host[1] == host[2] + host[3]
    f = rep["first_expired"]
    assert f is not None and f["kernel"] == "bf16 sweep" and f["step"] == -1 and not f["producer_counter"]
    assert f["counter_seen"] == 7 and f["target"] == 8 and f["polls"] == 2 ** 19 and 0 <= f["member"] < 8
    assert f["elapsed_cycles"] > 2 ** 19 * 128                      # at least the sleeps
    assert 0.02 < dt < 4.0, dt                                       # one bound per launch (two launches), not one per step
    assert eng.take_async_faults(host[1]) == int(host[1])
    assert len(E.FAULT_LOG) == n0 + 1 and E.FAULT_LOG[-1]["first_expired"] == f
    assert "first expired wait: bf16 sweep" in E.describe_fault_report(E.FAULT_LOG[-1])
    assert eng.cluster_fault_report()["first_expired"] is None       # the host has taken the description
    djenv.unset("DEEPJ_DEBUG_CLUSTER_LATE")
    loss = eng.train_fwd_bwd(P, G, *dn, seed=5)                      # the hook is gone: a clean step on the same engine
    host = loss.cpu().numpy()
    assert np.isfinite(host[0]) and host[1] == 0 and eng.cluster_faults() == 0


def _cluster_step(gpu_device, T, kernel_flags=0, seed=5):
    """one bf16 training step at B16 x T x N128 (64 time-axis tiles = 8 clusters) -> engine, [loss, faults..], gradient"""
    from music_generator_amd.engine import Engine
    B = 16
    ocfg, dcfg = _cfgs(time_steps=T, num_notes=128, dtype="bf16")
    flat = O.flatten_params(ocfg, O.init_params(ocfg, seed=11))
    batch = O.synthetic_batch(ocfg, B, seed=3, T=T)
    eng = Engine(dcfg, B, T, device=gpu_device, input_dropout=0.2, dropout=0.5, kernel_flags=kernel_flags)
    P = torch.from_numpy(flat).to(gpu_device)
    G = torch.empty_like(P)
    dn = [torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device) for a in batch]
    torch.cuda.synchronize()
    loss = eng.train_fwd_bwd(P, G, *dn, seed=seed)
    return eng, loss.cpu().numpy(), G


def test_tagged_and_counted_exchange_agree(gpu_device, djenv):
    """The time-axis sweep's members exchange h slices that announce themselves by a tag in a spare exponent bit (|h| < 1)
    instead of closing every step through the cluster's counter (DJ_KF_COUNTED_EXCHANGE keeps that protocol).  The
    sweeps themselves agree to the last bit (tests/test_kernels_gpu.py); a whole step carries atomically summed
    reductions (loss, weight gradients), so here: to their rounding, over a recurrence long enough for both ring slots
    and both tag values to be reused many times."""
    from music_generator_amd._lib import KF_COUNTED_EXCHANGE
    for name in ("DEEPJ_TAGGED_EXCHANGE", "DEEPJ_CLUSTER"):
        djenv.unset(name)
    for T in (5, 64):
        e1, l1, g1 = _cluster_step(gpu_device, T)
        e2, l2, g2 = _cluster_step(gpu_device, T, kernel_flags=KF_COUNTED_EXCHANGE)
        assert np.isfinite(l1[0]) and l1[1] == 0 and l2[1] == 0
        assert abs(l1[0] - l2[0]) <= 2e-6 * abs(l1[0]), (T, l1, l2)
        assert float((g1 - g2).abs().max()) <= 1e-5 * float(g1.abs().max()), T
        assert e1.cluster_faults() == 0 and e2.cluster_faults() == 0


@pytest.mark.fault_injection
def test_muted_member_expires_the_tagged_waits_once(gpu_device, djenv):
    """DEEPJ_DEBUG_CLUSTER_MUTE: the last member of every cluster stops publishing its h slices at step 2 of a tagged
    sweep.  Every wave of the cluster -- the muted member's own included -- then polls two stale fragments of h_2 until
    its bound runs out (2^17 polls of the fragments), all of them at the same time; each is counted, the first is described (kind: tagged
    fragments, step 3, 14 of 16 fragments there), the tiles are poisoned (NaN loss), and no poisoned wave waits again:
    the launch costs one bound, not one per remaining step."""
    import time
    from music_generator_amd import engine as E
    for name in ("DEEPJ_TAGGED_EXCHANGE", "DEEPJ_CLUSTER"):
        djenv.unset(name)
    djenv.set("DEEPJ_DEBUG_CLUSTER_MUTE", "1")
    n0, t0 = len(E.FAULT_LOG), time.time()
    eng, host, _ = _cluster_step(gpu_device, 24)
    dt = time.time() - t0
    rep = eng.cluster_fault_report()
    print("muted-member launch: %.3f s, [loss, faults, expired, misplaced] = %s, report %s" % (dt, host, rep))
    assert np.isnan(host[0]) and host[2] >= 8 and host[3] == 0 and host[1] == host[2]
    f = rep["first_expired"]
    assert f is not None and f["kernel"].startswith("bf16 sweep, tagged") and f["step"] == 3
    assert f["counter_seen"] == 14 and f["target"] == 16 and f["polls"] == 2 ** 17
    assert 0.02 < dt < 6.0, dt                                       # two launches (time layers), one bound each
    assert eng.take_async_faults(host[1]) == int(host[1])
    assert len(E.FAULT_LOG) == n0 + 1
    djenv.unset("DEEPJ_DEBUG_CLUSTER_MUTE")
    eng2, host2, _ = _cluster_step(gpu_device, 24)
    assert np.isfinite(host2[0]) and host2[1] == 0 and eng2.cluster_faults() == 0


@pytest.mark.fault_injection
def test_muted_member_in_the_inference_pair_raises_described(gpu_device, djenv):
    """The same hook on the cooperative inference pair (generation's time axis: both layers in one launch, tagged
    exchange): the gate waves of every member run out of polls on the muted member's fragments, the output is NaN, the
    census counts it and describes the first wait to run out: a gate wave on the muted member's fragments (kind
    'cooperative body, tagged', step 3, 14 of 16) or the upper layer on the stalled lower layer's counter."""
    from music_generator_amd.engine import Engine
    for name in ("DEEPJ_TAGGED_EXCHANGE", "DEEPJ_CLUSTER", "DEEPJ_CLUSTER_PAIR", "DEEPJ_CLUSTER_COOP"):
        djenv.unset(name)
    djenv.set("DEEPJ_DEBUG_CLUSTER_MUTE", "1")
    G, T, N = 3, 16, 48
    ocfg, dcfg = _cfgs(time_steps=T, num_notes=N, dtype="bf16")
    flat = torch.from_numpy(O.flatten_params(ocfg, O.init_params(ocfg, seed=11))).to(gpu_device)
    notes, chosen, beat, style, target = O.synthetic_batch(ocfg, G, seed=4, T=T)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
    eng = Engine(dcfg, G, T, device=gpu_device)
    out = eng.time_model_predict(flat, d(notes), d(beat), d(style)).float().cpu().numpy()
    rep = eng.cluster_fault_report()
    print("muted pair:", rep)
    assert np.isnan(out).any() and rep["expired"] >= 4 and rep["misplaced"] == 0
    f = rep["first_expired"]
    assert f is not None and f["kernel"].startswith("bf16 cooperative body")
    if "tagged" in f["kernel"]:          # a gate wave of the lower or the upper layer, on the muted member's fragments
        assert f["step"] == 3 and f["counter_seen"] == 14 and f["target"] == 16 and f["polls"] == 2 ** 17
    else:                                # or the upper layer first, on the counter of the (stalled) producing layer
        assert f["producer_counter"] and f["polls"] == 2 ** 19
    assert eng.cluster_faults("muted pair (test)") == rep["expired"]
    djenv.unset("DEEPJ_DEBUG_CLUSTER_MUTE")
    out = eng.time_model_predict(flat, d(notes), d(beat), d(style)).float().cpu().numpy()
    assert np.isfinite(out).all() and eng.cluster_faults() == 0


@pytest.mark.fault_injection
def test_injected_cluster_fault_is_never_silent(gpu_device, djenv, capsys):
    """DEEPJ_DEBUG_CLUSTER_FAULT makes the bf16 cluster kernels fail their placement check on the device (rows
    poisoned with NaN, the event counted in the workspace).  What the host side must make of it: train_on_batch
    reads the census WITH the loss (one copy), notices before the optimizer step, switches THIS model's engines to the
    per-tile kernel (dj_config.kernel_flags; the process environment is not touched, other models keep the cluster
    kernels) and repeats the step -- the result equals a fault-free step on the per-tile kernel; predict, evaluate and
    generation raise instead of returning NaN (or silence sampled from NaN)."""
    from music_generator_amd import generate as Gn
    from music_generator_amd._lib import DeepJError, KF_NO_CLUSTER
    from music_generator_amd.data import synthetic_batch
    from music_generator_amd.dataset import compute_genre
    from music_generator_amd.model import build_models
    T = 8
    a = synthetic_batch(48, T, 3, seed=2)
    x, y = [a[0], a[1], a[2], a[3]], [a[4]]
    djenv.unset("DEEPJ_DEBUG_CLUSTER_FAULT")
    djenv.unset("DEEPJ_CLUSTER")
    ref = build_models(time_steps=T, dtype="bf16", input_dropout=0.0, dropout=0.0, seed=4)
    ref[0]._s.add_kernel_flags(KF_NO_CLUSTER)                # reference: the per-tile kernels, no fault
    l_ref = ref[0].train_on_batch(x, y)
    w_ref = ref[0].get_weights()
    hm = build_models(time_steps=T, dtype="bf16", input_dropout=0.0, dropout=0.0, seed=4)
    djenv.set("DEEPJ_DEBUG_CLUSTER_FAULT", "1")
    from music_generator_amd import engine as E
    n_log = len(E.FAULT_LOG)
    with pytest.raises(DeepJError, match="cluster faults"):
        hm[0].predict(x)
    assert len(E.FAULT_LOG) == n_log + 1 and E.FAULT_LOG[-1]["misplaced"] > 0 and E.FAULT_LOG[-1]["expired"] == 0
    with pytest.raises(DeepJError, match="cluster faults"):
        hm[1].predict([x[0], x[2], x[3]])                    # time model: the wavefront launch
    np.random.seed(1)
    gm = build_models(dtype="bf16", seed=4)
    with pytest.raises(DeepJError, match="cluster faults"):
        list(Gn.generate(gm, 1, [compute_genre(i) for i in range(3)]))
    l_f = hm[0].train_on_batch(x, y)                         # falls back and repeats the step
    assert "falling back to the per-tile kernel" in capsys.readouterr().out
    assert hm[0]._s.kernel_flags == KF_NO_CLUSTER and "DEEPJ_CLUSTER" not in os.environ
    assert gm[0]._s.kernel_flags == 0                        # another model family is not downgraded
    assert np.isfinite(l_f) and abs(l_f - l_ref) < 1e-5 * abs(l_ref)     # (fp32 atomics reorder the loss / gradient sums)
    for wa, wb in zip(hm[0].get_weights(), w_ref):
        np.testing.assert_allclose(wa, wb, rtol=0, atol=1e-4)             # one Nadam step of lr 2e-3 on equal gradients
    l_g = hm[0].train_on_batch(x, y)                         # stays on the per-tile kernel: no fault, no message
    assert np.isfinite(l_g) and "falling back" not in capsys.readouterr().out
    assert np.isfinite(hm[0].predict(x)).all()               # its inference engines inherited the flag
    djenv.unset("DEEPJ_DEBUG_CLUSTER_FAULT")
    # and the hook is gone: a fresh model runs the cluster kernels without a fault
    ok = build_models(time_steps=T, dtype="bf16", input_dropout=0.0, dropout=0.0, seed=4)
    assert np.isfinite(ok[0].predict(x)).all()
    assert np.isfinite(ok[0].train_on_batch(x, y)) and ok[0]._s.kernel_flags == 0


def test_generate_with_hip_models_matches_oracle_models(gpu_device):
    """generate() (reference sampling semantics, NumPy RNG stream) with the HIP models vs the
    same harness with the CPU-oracle models, 160 time steps x 3 pieces x 48 notes -- longer than the 128-step
    window, so the last 32 steps see a window made of generated notes only (no zero prefix): identical sampled
    rolls under the same seed.  A Bernoulli decision u <= p can only differ between two
    implementations if u lies closer to p than their p's differ; the device sampler counts the
    draws within 1e-5 of p (fp32 outputs agree to ~1e-6), so the comparison is CERTIFIED rather than
    probable: with no near tie the whole run must be bit-equal, otherwise every step before the first
    near tie must be (DESIGN.md 'Sampling parity')."""
    from music_generator_amd import generate as Gn
    from music_generator_amd.dataset import compute_genre
    from music_generator_amd.model import build_models
    from oracle_backend import OracleBackend
    hm = build_models(seed=21)
    om = build_models(seed=21, backend=OracleBackend())
    for m in (hm, om):                                      # a head that actually plays notes
        w = m[0].get_weights()
        names = [n for n, _, _ in m[0]._s.layout]
        w[names.index("note_dense/bias")] = np.array([0.3, 0.0], np.float32)
        m[0].set_weights(w)
    styles = [compute_genre(i) for i in range(3)]
    steps = 160
    np.random.seed(1)                   # a stream without a near tie in these 160 steps (seeds 1, 6, 7, 9, 10: none)
    a = np.array(list(Gn.generate(hm, steps // 16, styles)))
    stats = dict(Gn.last_run_stats)
    assert a.shape[0] == steps and stats["draws"] >= steps * 3 * 48 and stats["near_ties"] >= 0
    sure = steps if stats["near_ties"] == 0 else stats["first_near_tie_step"]
    # ~45 k draws at 2e-5 each: a near tie somewhere is likely (60 %), one before step 144 would leave the
    # fully-generated window untested -- the seed is fixed, so this is a property of the build, not luck per run
    assert sure >= 144, stats
    np.random.seed(1)
    g = Gn.generate(om, steps // 16, styles)
    b = np.array([next(g) for _ in range(sure)])
    assert a[:sure, :, :, 0].sum() > 10 * sure                # both draw branches are exercised, in every step
    assert (a[16:sure, :, :, 0].sum(axis=(1, 2)) > 0).all()   # no all-silent step: the window really fills up
    np.testing.assert_array_equal(a[:sure, :, :, :2], b[:, :, :, :2])       # play / replay decisions
    np.testing.assert_allclose(a[:sure, :, :, 2], b[:, :, :, 2], rtol=1e-3, atol=1e-5)   # volumes


def test_fused_generation_step_matches_predict_loop(gpu_device, monkeypatch):
    """dj_generate_step (incremental note-axis state, device-side draws) vs the reference-shaped
    loop over time_model.predict / note_model.predict on the same HIP weights: same sampled
    rolls, same NumPy RNG position afterwards (the draw count is data dependent)."""
    from music_generator_amd import generate as Gn
    from music_generator_amd.dataset import compute_genre
    from music_generator_amd.model import build_models
    hm = build_models(seed=33)
    # make the note head less shy so that both draw branches are exercised
    w = hm[0].get_weights()
    names = [n for n, _, _ in hm[0]._s.layout]
    w[names.index("note_dense/bias")] = np.array([0.5, 0.0], np.float32)
    hm[0].set_weights(w)
    styles = [compute_genre(i) for i in range(3)]
    monkeypatch.setenv("DEEPJ_GENERATE_STEPWISE", "1")          # the per-step API (dj_generate_step)
    np.random.seed(11)
    fast = np.array(list(Gn.generate(hm, 1, styles))[:6])
    after_fast = np.random.random_sample(3)
    monkeypatch.setenv("DEEPJ_GENERATE_SLOW", "1")
    np.random.seed(11)
    g = Gn.generate(hm, 1, styles)
    slow = np.array([next(g) for _ in range(6)])
    assert slow[..., 0].sum() > 20                       # notes are actually being played
    np.testing.assert_array_equal(fast[:6, :, :, :2], slow[:, :, :, :2])
    np.testing.assert_allclose(fast[:6, :, :, 2], slow[:, :, :, 2], rtol=1e-3, atol=1e-5)
    # RNG stream position: the slow generator was advanced 6 steps, the fast one ran all 16
    np.random.seed(11)
    fast6 = []
    gf_env = monkeypatch.delenv("DEEPJ_GENERATE_SLOW")
    g2 = Gn.generate(hm, 1, styles)
    for _ in range(6):
        fast6.append(next(g2))
    pos_fast = np.random.random_sample(3)
    monkeypatch.setenv("DEEPJ_GENERATE_SLOW", "1")
    np.random.seed(11)
    g3 = Gn.generate(hm, 1, styles)
    for _ in range(6):
        next(g3)
    pos_slow = np.random.random_sample(3)
    np.testing.assert_array_equal(pos_fast, pos_slow)


def test_resident_graph_generation_matches_stepwise(gpu_device, monkeypatch):
    """Device-resident, hipGraph-replayed generation == step-wise fused == predict loop."""
    from music_generator_amd import generate as Gn
    from music_generator_amd.dataset import compute_genre
    from music_generator_amd.model import build_models
    hm = build_models(seed=33)
    w = hm[0].get_weights()
    names = [n for n, _, _ in hm[0]._s.layout]
    w[names.index("note_dense/bias")] = np.array([-0.5, 0.0], np.float32)
    hm[0].set_weights(w)
    styles = [compute_genre(i) for i in range(3)]
    np.random.seed(3)
    gen = Gn.generate(hm, 2, styles)                           # 32 steps: 2 chunks, graph replays
    head = [next(gen) for _ in range(5)]
    pos_mid = np.random.get_state()[2]                         # stream position after 5 yielded steps
    res = np.array(head + list(gen))
    pos_res = np.random.random_sample(2)
    monkeypatch.setenv("DEEPJ_GENERATE_STEPWISE", "1")
    np.random.seed(3)
    gen = Gn.generate(hm, 2, styles)
    head = [next(gen) for _ in range(5)]
    assert np.random.get_state()[2] == pos_mid
    stp = np.array(head + list(gen))
    pos_stp = np.random.random_sample(2)
    np.testing.assert_array_equal(res, stp)
    np.testing.assert_array_equal(pos_res, pos_stp)
    assert 0 < res[..., 0].sum() < res[..., 0].size              # some notes, some silence


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_prepared_generation_step_equals_resident_step(gpu_device, dtype):
    """dj_generate_prepare + dj_generate_step_prepared (per-run constants done once) against dj_generate_step_resident
    (redone every step): bit-equal notes, state and draw offset; another use of the engine between two run() calls
    (which overwrites the workspace) must not matter, because every run() prepares again."""
    from music_generator_amd.engine import Engine, ResidentGeneration, init_params_numpy
    from music_generator_amd.dataset import compute_genre
    ocfg, dcfg = _cfgs(dtype=dtype)
    G, T, N = 3, dcfg.time_steps, dcfg.num_notes
    eng = Engine(dcfg, G, T, device=gpu_device)
    flat = init_params_numpy(dcfg, seed=21)
    params = torch.from_numpy(flat).to(gpu_device)
    styles = [compute_genre(i) for i in range(G)]
    rng = np.random.RandomState(5)
    u1, u2 = rng.random_sample(2 * N * G * 6), rng.random_sample(2 * N * G * 5)
    outs = []
    for prepared in (True, False):
        run = ResidentGeneration(eng, params, styles, steps_cap=64, prepared=prepared)
        a, da = run.run(6, u1)
        # someone else uses the engine (and its workspace) between the two chunks
        z = torch.zeros(G, T, N, 3, device=gpu_device)
        eng.time_model_predict(params, z, torch.zeros(G, T, dcfg.notes_per_bar, device=gpu_device),
                               torch.zeros(G, T, dcfg.num_styles, device=gpu_device))
        b, db = run.run(5, u2)
        st = run.read_state()
        outs.append((a, da, b, db, int(st["step"]), int(st["near_ties"])))
    for x, y in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(x, y)
    assert outs[0][0][..., 0].sum() > 0


def test_resident_graph_generation_1024_steps(gpu_device, monkeypatch):
    """BASELINE configs[3] at its stated length: 3 style vectors, 1024-step pieces, hipGraph-replayed
    device-resident step vs the step-wise API on the same weights: bit-equal rolls, same NumPy RNG
    position, same near-tie census (reference generate.py:98-121).  The head bias (-6) makes most steps silent at
    T = 1, so the run spends most of its time in the heated branch of apply_temperature / end_time
    (generate.py:60-71,81-91) and still plays notes."""
    from music_generator_amd import generate as Gn
    from music_generator_amd.dataset import compute_genre
    from music_generator_amd.model import build_models
    hm = build_models(seed=33)
    w = hm[0].get_weights()
    names = [n for n, _, _ in hm[0]._s.layout]
    w[names.index("note_dense/bias")] = np.array([-6.0, 0.0], np.float32)
    hm[0].set_weights(w)
    styles = [compute_genre(i) for i in range(3)]
    bars = 1024 // 16
    np.random.seed(3)
    res = np.array(list(Gn.generate(hm, bars, styles)))
    pos_res = np.random.random_sample(2)
    st_res = dict(Gn.last_run_stats)
    assert res.shape == (1024, 3, 48, 3)
    monkeypatch.setenv("DEEPJ_GENERATE_STEPWISE", "1")
    np.random.seed(3)
    stp = np.array(list(Gn.generate(hm, bars, styles)))
    pos_stp = np.random.random_sample(2)
    st_stp = dict(Gn.last_run_stats)
    np.testing.assert_array_equal(res, stp)
    np.testing.assert_array_equal(pos_res, pos_stp)
    assert st_res == st_stp and st_res["draws"] == 1024 * 3 * 48 + int(res[..., 0].sum())
    assert 0 < res[..., 0].sum() < res[..., 0].size
    silent = res.reshape(1024, 3, -1).any(axis=2) == 0                      # [steps, pieces]
    assert silent.any() and (~silent).any()                                # silence AND notes
    assert st_res["silent_steps"] == int(silent.sum()) and st_res["max_temperature"] >= 1.3, st_res


def test_full_size_properties(gpu_device, djenv):
    """BASELINE shape (B64 x T128 x N128), where the CPU oracle is out of reach: size-independent
    properties instead.  (1) The fp32 gradient is the derivative of the fp32 loss: central
    difference along a random direction, dropout masks fixed by the seed.  (2) The two forward
    structures (x*W fused into the recurrence / separate GEMM) agree in fp32.  (3) bf16 agrees with
    fp32 at bf16 tolerance.  (4) The forward is bitwise reproducible."""
    from music_generator_amd.data import synthetic_batch
    from music_generator_amd.engine import DeepJConfig, Engine, init_params_numpy
    B, T, N, seed = 64, 128, 128, 7
    batch = [torch.from_numpy(a).to(gpu_device) for a in synthetic_batch(N, T, B, seed=0)]
    cfg32 = DeepJConfig(num_notes=N, time_steps=T, dtype="f32")
    P0 = torch.from_numpy(init_params_numpy(cfg32, seed=1234)).to(gpu_device)

    def run(cfg, P, fuse_min_tiles=None):
        djenv.unset("DEEPJ_FUSE_XW_MIN_TILES")
        eng = Engine(cfg, B, T, device=gpu_device, input_dropout=0.2, dropout=0.5,
                     fuse_xw_min_tiles=fuse_min_tiles or 0)          # dj_config.fuse_xw_min_tiles
        G = torch.empty_like(P)
        out = torch.empty((B, T, N, 3), dtype=torch.float32, device=gpu_device)
        loss = eng.train_fwd_bwd(P, G, *batch, seed=seed, out=out)
        torch.cuda.synchronize()
        res = float(loss.cpu()[0]), out.clone(), G.clone()
        del eng
        torch.cuda.empty_cache()
        return res

    l0, out0, g0 = run(cfg32, P0)
    assert np.isfinite(l0) and bool(torch.isfinite(g0).all())
    # (4) reproducible forward
    l0b, out0b, _ = run(cfg32, P0)
    assert torch.equal(out0, out0b)
    # (1) directional derivative along the gradient itself (a random direction in 1.27 M dimensions has a
    # slope of ~5e-4, below the ~1e-6 rounding noise of the fp32 loss sum divided by a usable eps); along
    # d = g/|g| the slope is |g| and the central difference resolves it to well under a percent
    gn = float(g0.double().norm())
    d = (g0.double() / gn).float()
    eps = 1e-2
    lp, _, _ = run(cfg32, P0 + eps * d)
    lm, _, _ = run(cfg32, P0 - eps * d)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - gn) <= 2e-2 * gn, (fd, gn)
    # (2) unfused forward structure (threshold above the 256 tiles of this shape)
    l1, out1, g1 = run(cfg32, P0, fuse_min_tiles=100000)
    torch.testing.assert_close(out1, out0, rtol=1e-3, atol=1e-5)
    assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0))
    assert float((g1 - g0).abs().max()) <= 2e-3 * float(g0.abs().max())
    # (3) bf16 against fp32
    cfg16 = DeepJConfig(num_notes=N, time_steps=T, dtype="bf16")
    l2, out2, g2 = run(cfg16, P0)
    assert abs(l2 - l0) <= 2e-2 * max(1.0, abs(l0))
    assert float((out2 - out0).abs().max()) <= 5e-2
    cos = float((g2.double() * g0.double()).sum() / (g2.double().norm() * g0.double().norm()))
    assert cos > 0.99, cos


def test_gradient_accumulation_over_micro_batches(gpu_device):
    """dj_train_fwd_bwd_acc: the gradients of a second micro-batch are added to the first's
    (every gradient kernel accumulates), which is how a global batch larger than one workspace runs."""
    from music_generator_amd.engine import Engine
    T, B = 6, 2
    ocfg, dcfg = _cfgs(time_steps=T, num_notes=24)
    flat = O.flatten_params(ocfg, O.init_params(ocfg, seed=4))
    P = torch.from_numpy(flat).to(gpu_device)
    eng = Engine(dcfg, B, T, device=gpu_device, input_dropout=0.2, dropout=0.5)
    mb = [[torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device)
           for a in O.synthetic_batch(ocfg, B, seed=s, T=T)] for s in (1, 2)]
    g1, g2, gacc = torch.empty_like(P), torch.empty_like(P), torch.empty_like(P)
    l1 = float(eng.train_fwd_bwd(P, g1, *mb[0], seed=11).cpu()[0])
    l2 = float(eng.train_fwd_bwd(P, g2, *mb[1], seed=12).cpu()[0])
    la = float(eng.train_fwd_bwd(P, gacc, *mb[0], seed=11).cpu()[0])
    lb = float(eng.train_fwd_bwd(P, gacc, *mb[1], seed=12, accumulate=True).cpu()[0])
    assert la == pytest.approx(l1, rel=1e-6) and lb == pytest.approx(l2, rel=1e-6)
    ref = (g1 + g2).cpu().numpy()
    np.testing.assert_allclose(gacc.cpu().numpy(), ref, rtol=1e-4, atol=1e-6 * float(np.abs(ref).max()))


@pytest.mark.parametrize("kw,B,micro", [(dict(num_notes=24), 4, 2), (dict(num_notes=20), 6, 2), (dict(num_notes=48), 6, 3),
                                        (dict(num_notes=24, time_axis_units=64, note_axis_units=96, time_axis_layers=3), 4, 1)])
def test_exact_micro_batches_vs_oracle_full_batch(gpu_device, kw, B, micro):
    """dj_pitch_bins + dj_train_fwd_bwd_mb: a batch run as micro-batches of `micro` samples through a workspace of that
    size is the step on the WHOLE batch -- loss and every gradient tensor against the oracle evaluated at the full
    batch (fp32, 1e-3 class tolerances, dropout on).  The reference's pitch_bins reshape (model.py:43-49) couples the
    samples of a batch and dropout masks are indexed by the batch's rows, so a naive split is a different model: the
    same micro-batches through dj_train_fwd_bwd_acc must NOT match (N = 20 also exercises N % 12 != 0; the last case
    the per-step path of other layer widths)."""
    from music_generator_amd.engine import Engine
    T, seed, pin, pdr = 5, 4242, 0.2, 0.5
    ocfg, dcfg = _cfgs(time_steps=T, **kw)
    params = O.init_params(ocfg, seed=8)
    flat = O.flatten_params(ocfg, params)
    batch = O.synthetic_batch(ocfg, B, seed=6, T=T)
    loss_ref, _, g_ref = O.loss_and_grads(ocfg, params, batch, O.make_masks(ocfg, B, seed, pin, pdr, T=T))
    dn = [torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(gpu_device) for a in batch]
    P = torch.from_numpy(flat).to(gpu_device)
    eng = Engine(dcfg, micro, T, device=gpu_device, input_dropout=pin, dropout=pdr)
    bins = eng.pitch_bins(dn[0], seed=seed)
    G = torch.empty_like(P)
    losses = []
    for i in range(B // micro):
        mb = [a[i * micro:(i + 1) * micro].contiguous() for a in dn]
        losses.append(float(eng.train_fwd_bwd(P, G, *mb, seed=seed, accumulate=i > 0, full_batch=B, batch_offset=i * micro,
                                              bins_full=bins).cpu()[0]))
    g = G.cpu().numpy() * (micro / B)
    loss = float(np.mean(losses))
    assert abs(loss - loss_ref) <= 1e-4 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    worst, rows = _grad_report(ocfg, g, g_ref)
    assert worst < 2e-3, sorted(rows, key=lambda r: -r[1])[:6]
    # the naive split (each micro-batch as a batch of its own) is a different computation
    G2 = torch.empty_like(P)
    for i in range(B // micro):
        mb = [a[i * micro:(i + 1) * micro].contiguous() for a in dn]
        eng.train_fwd_bwd(P, G2, *mb, seed=seed, accumulate=i > 0)
    worst2, _ = _grad_report(ocfg, G2.cpu().numpy() * (micro / B), g_ref)
    assert worst2 > 10 * worst, (worst, worst2)
    # argument checks: offsets outside the full batch
    with pytest.raises(Exception):
        eng.train_fwd_bwd(P, G2, *mb, seed=seed, full_batch=B, batch_offset=B - micro + 1, bins_full=bins)


def test_fit_micro_batches(gpu_device, monkeypatch):
    """DEEPJ_MICRO_BATCH: train_on_batch splits the batch into equal micro-batches with gradient accumulation --
    exactly the step on the whole batch (dj_train_fwd_bwd_mb): same loss and same weights after two steps as the
    same model trained without the split (fp32; the sums are merely ordered differently)."""
    from music_generator_amd.engine import DeepJConfig
    from music_generator_amd.model import build_models
    cfg = DeepJConfig(num_notes=24, time_steps=6)
    ocfg = O.OracleConfig(num_notes=24, time_steps=6)
    batch = O.synthetic_batch(ocfg, 4, seed=2, T=6)
    x, y = [batch[0], batch[4], batch[2], batch[3]], [batch[4]]
    ref = build_models(time_steps=6, config=cfg, seed=3)[0]
    lr = [ref.train_on_batch(x, y), ref.train_on_batch(x, y)]
    monkeypatch.setenv("DEEPJ_MICRO_BATCH", "2")
    m = build_models(time_steps=6, config=cfg, seed=3)[0]
    before = m.get_weights()[0].copy()
    l1 = m.train_on_batch(x, y)
    l2 = m.train_on_batch(x, y)
    assert np.isfinite(l1) and np.isfinite(l2) and l2 < l1 + 0.5
    assert float(np.abs(m.get_weights()[0] - before).max()) > 0
    assert l1 == pytest.approx(lr[0], rel=1e-5) and l2 == pytest.approx(lr[1], rel=1e-4)
    for a, b in zip(m.get_weights(), ref.get_weights()):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-5)
