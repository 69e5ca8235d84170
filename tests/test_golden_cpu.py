"""Host-side logic against the golden vectors captured from the REFERENCE's own code
(tests/golden/make_golden.py): codec, dataset windowing, temperature, and the sampling
harness driven by deterministic fake models.  No GPU, no reference tree needed."""
import json
import os

import numpy as np
import pytest

from fake_models import FakeNoteModel, FakeTimeModel

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _events(pattern, midi):
    out = []
    for track in pattern:
        tr = []
        for e in track:
            kind = 1 if type(e) is midi.NoteOnEvent else 0 if type(e) is midi.NoteOffEvent else 2
            tr.append([kind, int(e.tick), int(getattr(e, "pitch", 0)) if kind != 2 else 0,
                       int(getattr(e, "velocity", 0)) if kind != 2 else 0])
        out.append(tr)
    return out


def _pattern(events, res, midi):
    p = midi.Pattern(resolution=res)
    for tr in events:
        t = midi.Track()
        for kind, tick, pitch, vel in tr:
            t.append(midi.NoteOnEvent(tick=tick, pitch=pitch, velocity=vel) if kind == 1 else
                     midi.NoteOffEvent(tick=tick, pitch=pitch, velocity=vel) if kind == 0 else
                     midi.EndOfTrackEvent(tick=tick))
        p.append(t)
    return p


def test_constants_match_reference():
    from music_generator_amd import constants as C
    ref = np.load(os.path.join(G, "dataset.npz"))["constants"]
    mine = [C.NUM_STYLES, C.NUM_NOTES, C.NOTES_PER_BAR, C.BATCH_SIZE, C.SEQ_LEN, C.OCTAVE_UNITS, C.STYLE_UNITS,
            C.NOTE_UNITS, C.TIME_AXIS_UNITS, C.NOTE_AXIS_UNITS, C.TIME_AXIS_LAYERS, C.NOTE_AXIS_LAYERS, C.MIN_NOTE,
            C.MAX_NOTE, C.MIDI_MAX_NOTES, C.MAX_VELOCITY, C.DEFAULT_RES]
    assert list(ref) == mine


def test_codec_decode_cases():
    from music_generator_amd import midi_util, smf
    for c in json.load(open(os.path.join(G, "codec.json")))["decode_cases"]:
        roll = midi_util.midi_decode(_pattern(c["events"], c["res"], smf), c["classes"], step=c["step"])
        np.testing.assert_array_equal(roll, np.array(c["roll"]), err_msg=c["name"])


def test_codec_encode_cases_and_roundtrip():
    from music_generator_amd import midi_util, smf
    for c in json.load(open(os.path.join(G, "codec.json")))["encode_cases"]:
        roll = np.array(c["roll"])
        pat = midi_util.midi_encode(roll, step=c["step"])
        assert pat.resolution == c["resolution"]
        assert _events(pat, smf) == c["events"], c["name"]
        back = midi_util.midi_decode(pat, c["classes"], step=c["step"])
        np.testing.assert_array_equal(back, np.array(c["decoded"]), err_msg=c["name"])


def test_reference_unit_test_vectors():
    """The known answers of the reference's own test.py (test.py:7-53,55-77,110-131,134-155)."""
    from music_generator_amd import midi_util, smf
    from music_generator_amd.constants import NOTES_PER_BEAT
    comp = [[0, 1, 0, 0], [0, 1, 0, 0], [0, 1, 0, 1], [0, 1, 0, 1], [0, 0, 0, 1], [0, 0, 0, 0]]
    roll = np.stack([comp, np.zeros((6, 4)), np.array(comp) * 0.5], 2)
    pat = midi_util.midi_encode(roll, step=1)
    assert pat.resolution == NOTES_PER_BEAT and len(pat) == 1 and len(pat[0]) == 5
    on1, on2, off1, off2 = pat[0][:-1]
    assert [type(e) for e in (on1, on2, off1, off2)] == [smf.NoteOnEvent, smf.NoteOnEvent, smf.NoteOffEvent,
                                                         smf.NoteOffEvent]
    assert [(e.tick, e.pitch) for e in (on1, on2, off1, off2)] == [(0, 1), (2, 3), (2, 1), (1, 3)]
    np.testing.assert_array_equal(midi_util.midi_decode(pat, 4, step=1)[:, :, 0], comp)
    p = _pattern([[[1, 0, 0, 127], [1, 96, 1, 127], [0, 0, 0, 127], [0, 48, 1, 127], [2, 1, 0, 0]]], 96, smf)
    np.testing.assert_array_equal(midi_util.midi_decode(p, 4, step=48)[:, :, 0],
                                  [[1, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0]])
    p = _pattern([[[1, 0, 1, 127], [1, 0, 3, 127], [0, 1, 1, 127], [1, 2, 1, 127], [1, 2, 3, 127], [2, 1, 0, 0]]],
                 96, smf)
    np.testing.assert_array_equal(midi_util.midi_decode(p, 4, step=3)[:, :, 1],
                                  [[0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 0]])
    p = _pattern([[[1, 0, 0, 24], [1, 96, 1, 89], [0, 0, 0, 0], [0, 48, 1, 0], [2, 1, 0, 0]]], 96, smf)
    np.testing.assert_array_almost_equal(midi_util.midi_decode(p, 4, step=48)[:, :, 2],
                                         [[24 / 127, 0, 0, 0], [24 / 127, 0, 0, 0], [0, 89 / 127, 0, 0],
                                          [0, 0, 0, 0]], decimal=5)


def test_smf_reads_and_rewrites_reference_archives(tmp_path):
    """Real files written by the reference (midi_encode + python-midi): our reader must parse
    them, our writer must reproduce them byte for byte, and decode must match the reference."""
    from music_generator_amd import midi_util, smf
    z = np.load(os.path.join(G, "archive.npz"))
    for k in range(2):
        src = os.path.join(G, "archive_%d.mid" % k)
        pat = smf.read_midifile(src)
        assert pat.resolution == int(z["archive_%d_resolution" % k]) and len(pat) == 1
        out = tmp_path / ("re_%d.mid" % k)
        smf.write_midifile(str(out), pat)
        assert out.read_bytes() == open(src, "rb").read()
        roll = midi_util.midi_decode(pat)
        np.testing.assert_array_equal(roll, z["archive_%d_roll" % k])
        # encode(decode(file)) reproduces the note events of the file (it was produced by midi_encode)
        again = midi_util.midi_encode(roll)
        assert _events(again, smf) == _events(pat, smf)


def test_dataset_functions():
    from music_generator_amd import dataset as D
    from music_generator_amd.constants import NOTES_PER_BAR, genre
    from music_generator_amd.util import one_hot
    z = np.load(os.path.join(G, "dataset.npz"))
    clamped = D.clamp_midi(z["roll"])
    np.testing.assert_array_equal(clamped, z["clamped"])
    np.testing.assert_array_equal(D.unclamp_midi(clamped), z["unclamped"])
    X, Y = D.stagger(clamped, 8)
    np.testing.assert_array_equal(np.array(X), z["stagger_x"])
    np.testing.assert_array_equal(np.array(Y), z["stagger_y"])
    BX, _ = D.stagger([D.compute_beat(i, NOTES_PER_BAR) for i in range(len(clamped))], 8)
    np.testing.assert_array_equal(np.array(BX), z["beat_x"])
    np.testing.assert_array_equal([D.compute_beat(i, NOTES_PER_BAR) for i in range(40)], z["compute_beat"])
    np.testing.assert_array_equal([D.compute_genre(i) for i in range(len(genre))], z["compute_genre"])
    np.testing.assert_array_equal([one_hot(i, 7) for i in range(7)], z["one_hot"])


def test_apply_temperature_table():
    from music_generator_amd.generate import apply_temperature
    z = np.load(os.path.join(G, "temperature.npz"))
    p32 = z["temp_p"]
    assert p32.dtype == np.float32
    for ti, t in enumerate(z["temps"]):
        t = 1 if ti == 0 else float(t)
        for ri, row in enumerate(p32):
            o32 = apply_temperature(row, t)
            assert o32.dtype == np.float32            # float32 in, float32 out (SURVEY a-G (4))
            np.testing.assert_array_equal(o32, z["temp_out32"][ti, ri])
            np.testing.assert_array_equal(apply_temperature(row.astype(np.float64), t), z["temp_out64"][ti, ri])


@pytest.mark.parametrize("tag", ["genres", "single"])
def test_generate_harness_trace(tag):
    """generate() with deterministic fake models reproduces the reference's emitted rolls,
    per-call model inputs (digests), temperature/silence trajectory and RNG position."""
    from music_generator_amd import generate as Gn
    z = np.load(os.path.join(G, "generate.npz"))
    meta = json.load(open(os.path.join(G, "generate_meta.json")))[tag]
    tm, nm = FakeTimeModel(256), FakeNoteModel()
    pieces = []
    orig = Gn.MusicGeneration

    class Spy(orig):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            pieces.append(self)
            self.trace = []

        def end_time(self, t):
            r = super().end_time(t)
            self.trace.append((self.temperature, self.silent_time))
            return r

    draws = [0]
    real = np.random.random

    def counting(*a, **k):
        draws[0] += 1
        return real(*a, **k)

    np.random.seed(meta["seed"])
    Gn.MusicGeneration = Spy
    np.random.random = counting
    try:
        steps = list(Gn.generate((None, tm, nm), meta["bars"], list(z[tag + "_styles"])))
    finally:
        Gn.MusicGeneration = orig
        np.random.random = real
    after = np.random.random_sample(4)
    rolls = np.array(steps)
    assert str(rolls.dtype) == meta["roll_dtype"]
    np.testing.assert_array_equal(rolls, z[tag + "_rolls"])
    assert draws[0] == meta["draws"]
    np.testing.assert_array_equal(after, z[tag + "_rng_after"])
    np.testing.assert_array_equal(np.array([g.trace for g in pieces]), z[tag + "_trace"])
    np.testing.assert_allclose(tm.digests, z[tag + "_time_digests"], rtol=0, atol=0)
    np.testing.assert_allclose(nm.digests, z[tag + "_note_digests"], rtol=0, atol=0)
    assert [list(map(list, tm.shapes[0]))] == meta["time_shapes"] and tm.dtypes[0] == meta["time_dtypes"]
    assert [list(map(list, nm.shapes[0]))] == meta["note_shapes"] and nm.dtypes[0] == meta["note_dtypes"]
    assert len(tm.digests) == meta["n_time_calls"] and len(nm.digests) == meta["n_note_calls"]
