"""Device sampler, temperature path (reference generate.py:47-91): apply_temperature (float32 logit / T / sigmoid,
:81-91) and end_time's heating schedule (:60-71: a silent step adds 1 to silent_time and, from NOTES_PER_BAR silent
steps on, 0.1 to the temperature; a step with notes resets both; silent_time STARTS at NOTES_PER_BAR, so the very
first silent step already heats).

With random-init weights and note_dense/bias = [-9, 0] a step is silent with p ~ 0.98 at T = 1, so the temperature
climbs 0.1 per step until notes appear (T ~ 2) and resets: every run below executes logf/expf in the sampler, the
fp64 -> fp32 temperature hand-off and both branches of gen_state_kernel many times.  Three product paths on the same
HIP weights (hipGraph-replayed resident run, step-wise fused API, the reference-shaped predict() loop) and the
CPU-oracle models are compared under the near-tie certificate (DESIGN.md "Sampling parity")."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _set_head_bias(models, play_bias):
    w = models[0].get_weights()
    names = [n for n, _, _ in models[0]._s.layout]
    w[names.index("note_dense/bias")] = np.array([play_bias, 0.0], np.float32)
    models[0].set_weights(w)


def _host_schedule(rolls, notes_per_bar, default_temp=1.0):
    """end_time() of the reference replayed on emitted rolls [steps, G, N, 3] -> per-step (temperature, silent_time)
    AFTER the step, float64 like MusicGeneration."""
    steps, G = rolls.shape[:2]
    temp = [float(default_temp)] * G
    silent = [notes_per_bar] * G
    out_t, out_s = np.zeros((steps, G)), np.zeros((steps, G), np.int64)
    for t in range(steps):
        for g in range(G):
            if np.count_nonzero(rolls[t, g]) == 0:
                silent[g] += 1
                if silent[g] >= notes_per_bar:
                    temp[g] += 0.1
            else:
                silent[g] = 0
                temp[g] = default_temp
            out_t[t, g], out_s[t, g] = temp[g], silent[g]
    return out_t, out_s


def _run(models, bars, styles, seed, steps=None):
    from music_generator_amd import generate as Gn
    np.random.seed(seed)
    g = Gn.generate(models, bars, styles)
    rolls = np.array(list(g) if steps is None else [next(g) for _ in range(steps)])
    pos = np.random.get_state()[2] if steps is not None else None
    tail = np.random.random_sample(2)
    return rolls, dict(Gn.last_run_stats), pos, tail


def test_temperature_schedule_on_three_hip_paths(gpu_device, monkeypatch):
    """Forced silence -> heating -> notes -> reset, on the resident hipGraph path, the step-wise fused API and the
    predict() loop (same HIP weights, same seed): the first two are bit-equal (rolls, RNG position, schedule census);
    the predict() loop -- another kernel path for the probabilities and NumPy's float32 log/exp in apply_temperature
    -- is equal in every step the near-tie census certifies."""
    from music_generator_amd.dataset import compute_genre
    from music_generator_amd.model import build_models
    hm = build_models(seed=33)
    _set_head_bias(hm, -9.0)
    styles = [compute_genre(i) for i in range(3)]
    bars, steps = 3, 48
    monkeypatch.delenv("DEEPJ_GENERATE_STEPWISE", raising=False)
    monkeypatch.delenv("DEEPJ_GENERATE_SLOW", raising=False)
    res, st_res, _, tail_res = _run(hm, bars, styles, seed=7)
    # the schedule really ran: silence, heating well above 1, notes, resets
    want_t, want_s = _host_schedule(res, 16)
    assert st_res["silent_steps"] == int((want_s > 0).sum()) > 30
    assert st_res["max_temperature"] == want_t.max() and want_t.max() >= 1.5
    played = res[..., 0].sum(axis=(2,)) > 0                        # [steps, G]
    assert played.any() and (~played).any()
    resets = (want_t[1:] < want_t[:-1]).sum()
    assert resets >= 1, "no piece ever came back from a heated temperature"
    # at least one note was drawn while T != 1 (the temperature in force for step t is the one after step t - 1)
    before = np.vstack([np.ones((1, 3)), want_t[:-1]])
    assert (played & (before > 1.0)).any()
    monkeypatch.setenv("DEEPJ_GENERATE_STEPWISE", "1")
    stp, st_stp, _, tail_stp = _run(hm, bars, styles, seed=7)
    np.testing.assert_array_equal(res, stp)
    np.testing.assert_array_equal(tail_res, tail_stp)
    assert st_res == st_stp
    # the reference-shaped loop over time_model.predict / note_model.predict (+ host apply_temperature)
    monkeypatch.setenv("DEEPJ_GENERATE_SLOW", "1")
    sure = steps if st_res["near_ties"] == 0 else st_res["first_near_tie_step"]
    assert sure >= 8, st_res
    slow, st_slow, pos_slow, _ = _run(hm, bars, styles, seed=7, steps=sure)
    np.testing.assert_array_equal(res[:sure, :, :, :2], slow[:, :, :, :2])
    np.testing.assert_allclose(res[:sure, :, :, 2], slow[:, :, :, 2], rtol=1e-3, atol=1e-5)
    assert st_slow["max_temperature"] == want_t[:sure].max()
    # RNG stream position after `sure` steps: one draw per note + one per played note
    monkeypatch.delenv("DEEPJ_GENERATE_SLOW")
    monkeypatch.delenv("DEEPJ_GENERATE_STEPWISE")
    _, _, pos_res, _ = _run(hm, bars, styles, seed=7, steps=sure)
    assert pos_res == pos_slow


@pytest.mark.parametrize("notes_per_bar", [16, 8])
def test_device_schedule_state_at_every_chunk(gpu_device, notes_per_bar):
    """dj_gen_state.temperature / silent against the host replay of end_time() after chunks of uneven length, for the
    reference's NOTES_PER_BAR = 16 and for 8 (gen_state_kernel takes the threshold from dj_config.notes_per_bar: after a
    reset, heating resumes after notes_per_bar silent steps, which tells 8 from a hard-coded 16)."""
    from music_generator_amd.dataset import compute_genre
    from music_generator_amd.engine import DeepJConfig, Engine, ResidentGeneration, init_params_numpy, param_layout
    cfg = DeepJConfig(notes_per_bar=notes_per_bar)
    G, T, N = 3, cfg.time_steps, cfg.num_notes
    flat = init_params_numpy(cfg, seed=33)
    for name, off, shape in param_layout(cfg):
        if name == "note_dense/bias":
            flat[off:off + 2] = [-9.0, 0.0]
    params = torch.from_numpy(flat).to(gpu_device)
    eng = Engine(cfg, G, T, device=gpu_device)
    styles = [compute_genre(i) for i in range(G)]
    run = ResidentGeneration(eng, params, styles, steps_cap=128)
    rng = np.random.RandomState(11)
    rolls = []
    for k in (5, 2, 23, 1, 30, 19):
        notes, _ = run.run(k, rng.random_sample(2 * N * G * k))
        rolls.append(notes)
        want_t, want_s = _host_schedule(np.concatenate(rolls), notes_per_bar)
        st = run.last_state
        np.testing.assert_array_equal(st["temperature"][:G], want_t[-1])          # float64, same additions: bit-equal
        np.testing.assert_array_equal(st["silent"][:G], want_s[-1])
        assert int(st["step"]) == len(want_t)
    want_t, want_s = _host_schedule(np.concatenate(rolls), notes_per_bar)
    assert want_t.max() >= 1.5 and (want_t[1:] < want_t[:-1]).any()
    # a reset followed by at least notes_per_bar silent steps at T = 1 and then heating again: the threshold is exercised
    cold = (want_s > 0) & (want_s < notes_per_bar)
    assert cold.any() and (want_t[cold] == 1.0).all()
    if notes_per_bar == 8:
        assert ((want_s >= 8) & (want_s < 16) & (want_t > 1.0)).any()


def test_temperature_path_hip_models_vs_oracle_models(gpu_device):
    """The heated sampler against the CPU-oracle models through the same harness (reference semantics, NumPy stream):
    every step before the first certified near tie has identical play / replay decisions."""
    from music_generator_amd import generate as Gn
    from music_generator_amd.dataset import compute_genre
    from music_generator_amd.model import build_models
    from oracle_backend import OracleBackend
    hm = build_models(seed=21)
    om = build_models(seed=21, backend=OracleBackend())
    for m in (hm, om):
        _set_head_bias(m, -9.0)
    styles = [compute_genre(i) for i in range(3)]
    steps = 32
    a, stats, _, _ = _run(hm, 2, styles, seed=5)
    want_t, _ = _host_schedule(a, 16)
    sure = steps if stats["near_ties"] == 0 else stats["first_near_tie_step"]
    assert sure >= 16, stats
    assert want_t[:sure].max() >= 1.5 and a[:sure, :, :, 0].sum() > 0
    b, st_o, _, _ = _run(om, 2, styles, seed=5, steps=sure)
    np.testing.assert_array_equal(a[:sure, :, :, :2], b[:, :, :, :2])
    np.testing.assert_allclose(a[:sure, :, :, 2], b[:, :, :, 2], rtol=1e-3, atol=1e-5)
    assert st_o["max_temperature"] == want_t[:sure].max()


@pytest.mark.parametrize("G", [1, 8])
def test_fused_generation_piece_counts(gpu_device, monkeypatch, G):
    """generate() with one piece (generate.py --styles i j: a single mean style vector, generate.py:146-148) and with
    eight (the most the single-workgroup sampler takes): the resident hipGraph path against the reference-shaped loop
    over predict() on the same HIP weights -- same rolls in every step the near-tie census certifies, same RNG position."""
    from music_generator_amd.model import build_models
    from music_generator_amd.util import one_hot
    hm = build_models(seed=12)
    _set_head_bias(hm, 0.4)
    rs = np.random.RandomState(G)
    styles = [np.mean([one_hot(int(i), 23) for i in rs.randint(0, 23, size=2)], axis=0) for _ in range(G)]
    steps = 8
    monkeypatch.delenv("DEEPJ_GENERATE_STEPWISE", raising=False)
    monkeypatch.delenv("DEEPJ_GENERATE_SLOW", raising=False)
    res, st, _, _ = _run(hm, 1, styles, seed=3)
    assert res.shape == (16, G, 48, 3) and res[..., 0].sum() > 10 * G
    sure = 16 if st["near_ties"] == 0 else st["first_near_tie_step"]
    sure = min(sure, steps)
    assert sure >= 4, st
    _, _, pos_res, _ = _run(hm, 1, styles, seed=3, steps=sure)
    monkeypatch.setenv("DEEPJ_GENERATE_SLOW", "1")
    slow, _, pos_slow, _ = _run(hm, 1, styles, seed=3, steps=sure)
    np.testing.assert_array_equal(res[:sure, :, :, :2], slow[:, :, :, :2])
    np.testing.assert_allclose(res[:sure, :, :, 2], slow[:, :, :, 2], rtol=1e-3, atol=1e-5)
    assert pos_res == pos_slow


def test_matrix_core_sampler_matches_vector_sampler_bf16(gpu_device, djenv):
    """bf16 mode: the note walk of a generated step on the matrix cores (gen_sample_mfma_kernel: bf16 weight fragments,
    fp32 accumulation, round 5) against the same walk on the vector ALUs with the fp32 master weights
    (DJ_KF_NO_GEN_MFMA / DEEPJ_GEN_MFMA=0: the kernel the fp32 mode keeps), through dj_generate_step on ONE engine and one
    window.  Uniforms of 1e-9 make every draw a success whatever the probability, so both walks feed the same `chosen`
    (play = 1, replay = 1, the sampled volume) back into the note axis and consume 2 draws per note and piece; the
    VOLUMES -- a linear read-out of the top hidden state at every note, i.e. of the whole 48-note x 2-layer recurrence --
    then agree to bf16 operand rounding.  Uniforms of 1 - 1e-9 make every draw a failure: all-zero notes, one draw each."""
    import torch
    from music_generator_amd.engine import DeepJConfig, Engine, init_params_numpy
    dev = gpu_device
    G, T, N = 3, 128, 48
    cfg = DeepJConfig(num_notes=N, time_steps=T, dtype="bf16")
    rs = np.random.RandomState(5)
    P = init_params_numpy(cfg, seed=21)
    P = torch.from_numpy(P).to(dev)
    notes = torch.from_numpy((rs.rand(G, T, N, 3) < 0.1).astype(np.float32)).to(dev)
    beat = torch.zeros(G, T, 16, device=dev)
    beat[:, torch.arange(T), torch.arange(T) % 16] = 1
    style = torch.zeros(G, T, 23, device=dev)
    style[torch.arange(G), :, torch.arange(G) * 7 % 23] = 1
    temp = torch.ones(G, dtype=torch.float32, device=dev)

    def run(mfma, uniforms):
        djenv.set("DEEPJ_GEN_MFMA", "1" if mfma else "0")
        eng = Engine(cfg, G, T, device=dev)
        u = torch.from_numpy(uniforms).to(dev)
        out, used = eng.generate_step(P, notes, beat, style, u, temp)
        torch.cuda.synchronize()
        r = out.cpu().numpy(), used.cpu().numpy().copy()
        eng.close()
        return r

    lo = np.full(2 * N * G, 1e-9)
    a, ua = run(True, lo)
    b, ub = run(False, lo)
    assert ua[0] == ub[0] == 2 * N * G
    assert np.all(a[..., :2] == 1.0) and np.all(b[..., :2] == 1.0)
    assert np.isfinite(a).all() and np.abs(b[..., 2]).max() > 1e-3
    err = np.abs(a[..., 2] - b[..., 2]).max() / max(np.abs(b[..., 2]).max(), 1e-6)
    print("matrix-core vs vector sampler, volumes over 48 notes x 3 pieces: max |diff| / max |v| = %.2e" % err)
    assert err < 3e-2, err                                 # bf16 operands (tolerance of the bf16 forward)
    hi = np.full(2 * N * G, 1.0 - 1e-9)
    a, ua = run(True, hi)
    b, ub = run(False, hi)
    assert ua[0] == ub[0] == N * G and not a.any() and not b.any()
