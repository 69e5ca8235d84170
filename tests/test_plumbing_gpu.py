"""BASELINE configs[0] on the GPU: the same 4 synthetic .mid files -> load_all -> train.main ->
Model.fit(batch_size=2, T=8, epochs=1) with the HIP backend, compared batch by batch with the
oracle-backed twin of the same host code (dropout ON: both regenerate the same counter-hash masks);
then generate.main on the HIP models writes decodable .mid files.
reference train.py:14-29, dataset.py:39-76, generate.py:98-150."""
import os

import numpy as np
import pytest

import plumbing
from oracle_backend import OracleBackend

pytestmark = pytest.mark.gpu


def test_midi_corpus_fit_matches_oracle_backend(gpu_device, tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    plumbing.write_corpus(str(tmp_path))
    from music_generator_amd import constants as K, dataset, train, util
    from music_generator_amd.callbacks import LambdaCallback
    T = 8
    x, y = dataset.load_all(K.styles, 2, T)
    losses = {}

    def run(tag, **kw):
        models = util.build_or_load(allow_load=False, time_steps=T, seed=7, **kw)
        got = []
        np.random.seed(0)
        hist = models[0].fit(x, y, epochs=1, batch_size=2, verbose=0,
                             callbacks=[LambdaCallback(on_batch_end=lambda b, logs: got.append(logs["loss"]))])
        losses[tag] = (np.array(got), hist.history["loss"][0], models[0].get_weights())

    run("hip")
    run("oracle", backend=OracleBackend())
    hb, he, hw = losses["hip"]
    ob, oe, ow = losses["oracle"]
    assert len(hb) == (x[0].shape[0] + 1) // 2 >= 4
    np.testing.assert_allclose(hb, ob, rtol=5e-4)                      # every batch loss of the epoch
    assert abs(he - oe) <= 2e-4 * abs(oe)
    for a, b in zip(hw, ow):
        np.testing.assert_allclose(a, b, rtol=0, atol=3e-4)            # after ~6 Nadam steps of lr 2e-3
    # and through the CLI entry point, checkpoint + log files included
    np.random.seed(0)
    hist = train.main(["--batch-size", "2", "--time-steps", str(T), "--epochs", "1", "--dtype", "f32"])
    assert np.isfinite(hist.history["loss"][0]) and os.path.exists(K.MODEL_FILE)
    assert "Total params: 1,269,476" in capsys.readouterr().out


def test_generate_main_writes_decodable_files(gpu_device, tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    from music_generator_amd import constants as K, dataset, generate, midi_util, smf, util
    np.random.seed(4)
    generate.main(["--bars", "1"])
    out = capsys.readouterr().out
    files = [os.path.join(K.SAMPLES_DIR, "output_%d.mid" % i) for i in range(3)]
    assert all(os.path.exists(f) for f in files) and "Unable to load model from file." in out
    np.random.seed(4)
    steps = list(generate.generate(util.build_or_load(), 1, [generate.compute_genre(i) for i in range(3)]))
    assert len(steps) == 16
    for i, f in enumerate(files):
        roll = np.array([s[i] for s in steps])
        got = midi_util.midi_decode(smf.read_midifile(f))
        # the file holds exactly what the codec makes of the sampled roll (generate.py:123-134): the same events as
        # encoding the roll in memory.  (That is not the roll itself where the wire format cannot carry it: the
        # velocity byte is int(volume * 127) & 0x7F of the raw, unclipped volume head (midi_util.py:43,66;
        # generate.py:55) and a velocity of 0 is a note-off -- with random-init weights a share of the notes comes
        # out that way, as in the reference; writer / reader / codec are pinned byte-exactly in test_golden_cpu.py.)
        wire = roll.copy()                         # what the 7-bit velocity byte keeps of the volume
        wire[:, :, 2] = ((roll[:, :, 2] * 127).astype(np.int64) & 0x7F).astype(np.float64) / 127 + 0.5 / 127
        want = midi_util.midi_decode(midi_util.midi_encode(dataset.unclamp_midi(wire)))
        np.testing.assert_array_equal(got[:, :, :2], want[:, :, :2])
        np.testing.assert_allclose(got[:, :, 2], want[:, :, 2], atol=1e-6)
        on = np.nonzero(roll[:, :, 0].any(axis=1))[0]
        if len(on):
            L = on.max() + 1
            assert got[:L, 36:84, 0][roll[:L, :, 0] == 0].sum() == 0      # nothing that was not sampled
            assert got[:L, 36:84, 0].sum() > 0


def test_visualize_main_on_hip_models(gpu_device, tmp_path, monkeypatch, capsys):
    """SURVEY f-4 on the product backend (reference visualize.py:11-43): build_or_load -> get_layer('style') applied to
    the identity ON THE GPU (dj_style_embedding) -> the two TSV files the reference writes, with its names, its
    (Genre, Artist) header and one row per style; the matrix equals kernel + bias of the style Dense layer."""
    monkeypatch.chdir(tmp_path)
    from music_generator_amd import constants as K, util, visualize
    models = util.build_or_load()
    w, b = models[0].get_layer('style').get_weights()
    emb = visualize.main(models=models)
    assert emb.shape == (K.NUM_STYLES, K.STYLE_UNITS) and emb.dtype == np.float32
    np.testing.assert_allclose(emb, w + b, rtol=1e-6, atol=1e-7)
    got = np.loadtxt(os.path.join(K.OUT_DIR, 'style_embedding_vec.tsv'), delimiter='\t')
    np.testing.assert_allclose(got, emb, rtol=1e-6)
    rows = [ln.rstrip('\n').split('\t') for ln in open(os.path.join(K.OUT_DIR, 'style_embedding_labels.tsv'))]
    assert rows[0] == ['Genre', 'Artist'] and len(rows) == 1 + K.NUM_STYLES
    assert rows[1] == ['baroque', 'data/baroque/bach'] and rows[-1][0] == 'romantic'
    # non-identity input through the same device kernel
    x = np.random.RandomState(0).rand(5, K.NUM_STYLES)
    np.testing.assert_allclose(models[0].get_layer('style')(x), x.astype(np.float32) @ w + b, rtol=1e-5, atol=1e-6)
    visualize.main()                                            # the CLI form: build_or_load inside
    assert "Writing to out directory" in capsys.readouterr().out


def test_engine_close_and_pending_census(gpu_device):
    """An engine's finalizer does no HIP work (round 5: a finalizer may run inside somebody's graph capture): close() takes
    the last fault census and releases the workspace; an engine that is simply dropped hands its workspace to
    engine.drain_pending(), which takes the census at the next safe point -- Engine(), close(), the test fixture."""
    import gc
    import torch
    from music_generator_amd import engine as E
    from music_generator_amd.engine import DeepJConfig, Engine
    cfg = DeepJConfig(num_notes=24, time_steps=4)
    gc.collect()
    E.drain_pending()
    n0 = E.CENSUS["engines"]
    eng = Engine(cfg, 2, 4, device=gpu_device)
    assert eng in E._LIVE
    eng.close()
    eng.close()                                            # idempotent
    assert E.CENSUS["engines"] == n0 + 1 and eng not in E._LIVE and eng.ws is None
    eng2 = Engine(cfg, 2, 4, device=gpu_device)
    ws_ptr = eng2.ws.data_ptr()
    del eng2
    gc.collect()
    assert len(E._PENDING) == 1 and E._PENDING[0][2].data_ptr() == ws_ptr      # workspace parked, not read
    assert E.CENSUS["engines"] == n0 + 1
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x = torch.zeros(8, device=gpu_device)
        with torch.cuda.graph(g, stream=s):
            assert E.drain_pending() == 0                  # never inside a capture
            x += 1
    assert len(E._PENDING) == 1
    assert E.drain_pending() == 1 and not E._PENDING and E.CENSUS["engines"] == n0 + 2
    with Engine(cfg, 2, 4, device=gpu_device) as eng3:     # context manager = close()
        assert eng3.cluster_faults() == 0
    assert E.CENSUS["engines"] == n0 + 3
