"""BASELINE configs[0] end to end on the CPU (oracle backend injected behind the product's host code):
4 synthetic .mid files -> dataset.load_all -> train.train / train.main -> Model.fit(batch_size=2, T=8,
epochs=1); generate.write_file and generate.main -> .mid files that decode back to the sampled rolls.
reference train.py:14-29, dataset.py:39-76, generate.py:123-150."""
import os

import numpy as np

import plumbing
from fake_models import FakeNoteModel, FakeTimeModel
from oracle_backend import OracleBackend


def _oracle_build_or_load(**kw):
    from music_generator_amd import util
    kw.pop("dtype", None)
    return util.build_or_load(backend=OracleBackend(), seed=7, **kw)


def test_midi_files_to_load_all_to_fit(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    paths = plumbing.write_corpus(str(tmp_path))
    from music_generator_amd import constants as K, dataset, midi_util, smf, train
    T = 8
    # the files on disk decode back to what was encoded (play channel; python-midi free SMF round trip)
    for (d, name, length, seed), path in zip(plumbing.FILES, paths):
        roll = midi_util.midi_decode(smf.read_midifile(path))
        want = plumbing.random_roll(length, seed)
        last = np.nonzero(want[:, :, 0].any(axis=1))[0].max() + 1      # trailing silence carries no events
        np.testing.assert_array_equal(roll[:last, :, 0], want[:last, :, 0])
    x, y = dataset.load_all(K.styles, 2, T)
    n = sum(len(range(0, length, K.NOTES_PER_BAR)) for _, _, length, _ in plumbing.FILES)
    assert x[0].shape[1:] == (T, 48, 3) and x[2].shape[1:] == (T, 16) and x[3].shape[1:] == (T, 23)
    assert x[0].shape[0] == x[1].shape[0] == x[2].shape[0] == x[3].shape[0] == y[0].shape[0] >= 8
    np.testing.assert_array_equal(x[1], y[0])                               # chosen_in IS the target (dataset.py:76)
    np.testing.assert_array_equal(x[0][:, 1:], y[0][:, :-1])                # Y = X one step later
    assert set(np.argmax(x[3][:, 0], axis=1)) == {0, 8, 12}                 # bach, mozart, chopin style ids
    assert os.path.exists(os.path.join("out", "cache", "data/baroque/bach/a.mid.npy"))  # load_midi's .npy cache

    # train.main: argparse -> build_or_load -> load_all -> fit with the reference's callbacks
    monkeypatch.setattr(train, "build_or_load", _oracle_build_or_load)
    np.random.seed(0)
    hist = train.main(["--batch-size", "2", "--time-steps", str(T), "--epochs", "1"])
    out = capsys.readouterr().out
    assert "Loading data" in out and "Training" in out and "Total params: 1,269,476" in out
    assert len(hist.history["loss"]) == 1 and np.isfinite(hist.history["loss"][0])
    assert os.path.exists(K.MODEL_FILE)                                      # ModelCheckpoint(save_best_only)
    assert os.path.exists(os.path.join("out", "logs", "scalars.csv"))
    # the checkpoint is what a second build_or_load restores (util.py:13-23)
    m2 = _oracle_build_or_load(time_steps=T)
    assert "Loaded model from file." in capsys.readouterr().out
    with np.load(K.MODEL_FILE) as z:
        np.testing.assert_array_equal(m2[0].get_weights()[0], z["style/kernel"])


def test_write_file_and_generate_main(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    from music_generator_amd import constants as K, generate, midi_util, smf
    monkeypatch.setattr(generate, "build_or_load", lambda: (None, FakeTimeModel(), FakeNoteModel()))
    np.random.seed(4)
    generate.main(["--bars", "1"])                                           # 3 genre pieces, 16 steps
    out = capsys.readouterr().out
    files = [os.path.join(K.SAMPLES_DIR, "output_%d.mid" % i) for i in range(3)]
    assert all(("Writing file " + f) in out for f in files) and all(os.path.exists(f) for f in files)
    # the same run again through generate() gives the rolls the files must hold
    np.random.seed(4)
    steps = list(generate.generate((None, FakeTimeModel(), FakeNoteModel()),
                                   1, [generate.compute_genre(i) for i in range(3)]))
    assert len(steps) == 16
    for i, f in enumerate(files):
        roll = np.array([s[i] for s in steps])                               # [16, 48, 3]
        got = midi_util.midi_decode(smf.read_midifile(f))
        assert got.shape[1:] == (128, 3)
        last = np.nonzero(roll[:, :, 0].any(axis=1))[0]
        if len(last):
            L = last.max() + 1
            np.testing.assert_array_equal(got[:L, 36:84, 0], roll[:L, :, 0])
            assert got[:L, :36].sum() == 0 and got[:L, 84:].sum() == 0       # unclamp_midi placement
    # --styles mixes one-hot style vectors into ONE piece (generate.py:146-148)
    np.random.seed(4)
    generate.main(["--bars", "1", "--styles", "0", "5"])
    assert "output_0.mid" in capsys.readouterr().out
