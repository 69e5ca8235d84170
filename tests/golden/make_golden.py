#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE's own non-Keras code.

Runs ONLY in the authoring container (needs /root/reference).  The reference's
`dataset`, `generate`, `midi_util`, `util`, `constants` modules are imported
unmodified with two build-owned stand-ins first on sys.path (tests/_standins:
an empty `tensorflow`, a ~40-line `midi`), exactly as SURVEY.md 8c describes.
Outputs are DATA ONLY (inputs + expected outputs) written next to this script;
no reference source is copied.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
REF = os.environ.get("DEEPJ_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(TESTS, "_standins"))
sys.path.insert(1, REF)
sys.path.insert(2, TESTS)
os.chdir(tempfile.mkdtemp(prefix="deepj_golden_"))

import numpy as np  # noqa: E402

import midi  # noqa: E402  (stand-in)
import constants as C  # noqa: E402  (reference)
import dataset as D  # noqa: E402
import generate as Gn  # noqa: E402
import midi_util as MU  # noqa: E402
import util as U  # noqa: E402
from fake_models import FakeNoteModel, FakeTimeModel  # noqa: E402


def events_of(pattern):
    out = []
    for track in pattern:
        tr = []
        for e in track:
            kind = {midi.NoteOnEvent: 1, midi.NoteOffEvent: 0, midi.EndOfTrackEvent: 2}[type(e)]
            tr.append([kind, int(e.tick), int(getattr(e, "pitch", 0)), int(getattr(e, "velocity", 0))])
        out.append(tr)
    return out


def pattern_from(events, resolution):
    p = midi.Pattern(resolution=resolution)
    for tr in events:
        t = midi.Track()
        for kind, tick, pitch, vel in tr:
            if kind == 1:
                t.append(midi.NoteOnEvent(tick=tick, pitch=pitch, velocity=vel))
            elif kind == 0:
                t.append(midi.NoteOffEvent(tick=tick, pitch=pitch, velocity=vel))
            else:
                t.append(midi.EndOfTrackEvent(tick=tick))
        p.append(t)
    return p


def random_roll(rs, L, n, p_on=0.25):
    """A valid piano roll: held notes with random onsets, replays only while held."""
    play = np.zeros((L, n))
    replay = np.zeros((L, n))
    vol = np.zeros((L, n))
    for k in range(n):
        t = 0
        while t < L:
            if rs.random_sample() < p_on:
                dur = rs.randint(1, 6)
                v = rs.randint(20, 127) / 127.0
                play[t:t + dur, k] = 1
                vol[t:t + dur, k] = v
                for u in range(t + 1, min(t + dur, L)):
                    if rs.random_sample() < 0.2:
                        replay[u, k] = 1
                t += dur + rs.randint(0, 3)
            else:
                t += 1
    return np.stack([play, replay, vol], axis=2)


def codec_golden():
    g = {}
    cases = []
    # Event lists of the reference's own decode tests (test.py:55-77,110-131,134-155) -- data only.
    decode_cases = [
        dict(name="test_decode", res=96, classes=4, step=48,
             events=[[[1, 0, 0, 127], [1, 96, 1, 127], [0, 0, 0, 127], [0, 48, 1, 127], [2, 1, 0, 0]]]),
        dict(name="test_replay_decode", res=96, classes=4, step=3,
             events=[[[1, 0, 1, 127], [1, 0, 3, 127], [0, 1, 1, 127], [1, 2, 1, 127], [1, 2, 3, 127], [2, 1, 0, 0]]]),
        dict(name="test_volume_decode", res=96, classes=4, step=48,
             events=[[[1, 0, 0, 24], [1, 96, 1, 89], [0, 0, 0, 0], [0, 48, 1, 0], [2, 1, 0, 0]]]),
        dict(name="two_tracks", res=8, classes=6, step=2,
             events=[[[1, 0, 2, 100], [0, 6, 2, 0], [2, 2, 0, 0]],
                     [[1, 3, 4, 60], [1, 2, 5, 70], [0, 9, 4, 0], [0, 0, 5, 0], [2, 0, 0, 0]]]),
        dict(name="default_step", res=8, classes=5, step=None,
             events=[[[1, 1, 0, 90], [1, 3, 0, 50], [0, 5, 0, 0], [1, 0, 3, 127], [0, 7, 3, 0], [2, 4, 0, 0]]]),
    ]
    for c in decode_cases:
        roll = MU.midi_decode(pattern_from(c["events"], c["res"]), c["classes"], step=c["step"])
        cases.append(dict(c, roll=roll.tolist()))
    g["decode_cases"] = cases

    # Rolls of the reference's encode tests (test.py:7-53,79-108,158-193) + seeded random rolls.
    comp1 = [[0, 1, 0, 0], [0, 1, 0, 0], [0, 1, 0, 1], [0, 1, 0, 1], [0, 0, 0, 1], [0, 0, 0, 0]]
    rep1 = np.zeros((6, 4)).tolist()
    vol1 = (np.array(comp1) * 0.5).tolist()
    comp2 = [[0, 1, 0, 1], [0, 0, 0, 1], [0, 0, 0, 1], [0, 1, 0, 1], [0, 1, 0, 1], [0, 1, 0, 1], [0, 0, 0, 0]]
    rep2 = [[0, 0, 0, 0]] * 4 + [[0, 0, 0, 1], [0, 1, 0, 1], [0, 0, 0, 0]]
    vol2 = (np.array(comp2) * 0.5).tolist()
    rs = np.random.RandomState(11)
    enc = []
    rolls = [("test_encode", np.stack([comp1, rep1, vol1], 2), 1, 4),
             ("test_replay_encode_decode", np.stack([comp2, rep2, vol2], 2), 2, 4)]
    for i in range(4):
        n = [4, 12, 48, 128][i]
        rolls.append((f"random{i}", random_roll(rs, 24 + 8 * i, n), [1, 2, 1, 3][i], n))
    for name, roll, step, classes in rolls:
        pat = MU.midi_encode(roll, step=step)
        back = MU.midi_decode(pat, classes, step=step)
        enc.append(dict(name=name, step=step, classes=classes, roll=np.asarray(roll).tolist(),
                        resolution=int(pat.resolution), events=events_of(pat), decoded=back.tolist()))
    g["encode_cases"] = enc
    return g


def archive_golden():
    """Two of the reference's archived generated samples (archives/v1/long_samples): DATA
    files written by the reference's own midi_encode + python-midi.  The bytes pin this
    build's SMF reader/writer; the decoded rolls pin midi_decode on real files."""
    import shutil
    out = {}
    src = os.path.join(REF, "archives", "v1", "long_samples")
    for k, name in enumerate(["Baroque 1.mid", "Romantic 2.mid"]):
        dst = os.path.join(HERE, "archive_%d.mid" % k)
        shutil.copyfile(os.path.join(src, name), dst)
        os.chmod(dst, 0o644)
        pat = midi.read_midifile(dst)
        out["archive_%d_roll" % k] = MU.midi_decode(pat)
        out["archive_%d_resolution" % k] = np.array(pat.resolution)
    return out


def dataset_golden():
    rs = np.random.RandomState(5)
    roll = random_roll(rs, 40, C.MIDI_MAX_NOTES)
    clamped = D.clamp_midi(roll)
    X, Y = D.stagger(clamped, 8)
    beats = [D.compute_beat(i, C.NOTES_PER_BAR) for i in range(len(clamped))]
    BX, _ = D.stagger(beats, 8)
    return dict(
        roll=roll, clamped=clamped, unclamped=D.unclamp_midi(clamped),
        stagger_x=np.array(X), stagger_y=np.array(Y), beat_x=np.array(BX),
        compute_beat=np.array([D.compute_beat(i, C.NOTES_PER_BAR) for i in range(40)]),
        compute_genre=np.array([D.compute_genre(i) for i in range(len(C.genre))]),
        one_hot=np.array([U.one_hot(i, 7) for i in range(7)]),
        constants=np.array([C.NUM_STYLES, C.NUM_NOTES, C.NOTES_PER_BAR, C.BATCH_SIZE, C.SEQ_LEN,
                            C.OCTAVE_UNITS, C.STYLE_UNITS, C.NOTE_UNITS, C.TIME_AXIS_UNITS,
                            C.NOTE_AXIS_UNITS, C.TIME_AXIS_LAYERS, C.NOTE_AXIS_LAYERS,
                            C.MIN_NOTE, C.MAX_NOTE, C.MIDI_MAX_NOTES, C.MAX_VELOCITY, C.DEFAULT_RES]),
    )


def temperature_golden():
    p32 = np.array([[1e-6, 0.5], [0.01, 0.99], [0.25, 0.75], [0.5, 0.123], [0.9, 0.3], [0.999, 1e-4]], np.float32)
    temps = [1, 1.0, 1.1, 1.2000000000000002, 1.5, 2.0, 0.5, 3.1]
    out32 = np.stack([np.stack([Gn.apply_temperature(row, t) for row in p32]) for t in temps])
    out64 = np.stack([np.stack([Gn.apply_temperature(row.astype(np.float64), t) for row in p32]) for t in temps])
    return dict(temp_p=p32, temps=np.array(temps, np.float64), temp_out32=out32, temp_out32_dtype=str(out32.dtype),
                temp_out64=out64)


def generate_golden():
    out = {}
    meta = {}
    for tag, seed, bars, styles in [
        ("genres", 123, 2, [D.compute_genre(i) for i in range(len(C.genre))]),
        ("single", 7, 3, [np.mean([U.one_hot(i, C.NUM_STYLES) for i in (1, 5, 20)], axis=0)]),
    ]:
        tm, nm = FakeTimeModel(C.TIME_AXIS_UNITS), FakeNoteModel()
        draws = [0]
        real_random = np.random.random

        def counting_random(*a, **k):
            draws[0] += 1
            return real_random(*a, **k)

        np.random.seed(seed)
        np.random.random = counting_random
        try:
            gens = []
            orig_MG = Gn.MusicGeneration

            class Spy(orig_MG):
                def __init__(self, *a, **k):
                    super().__init__(*a, **k)
                    gens.append(self)
                    self.trace = []

                def end_time(self, t):
                    r = super().end_time(t)
                    self.trace.append((self.temperature, self.silent_time))
                    return r

            Gn.MusicGeneration = Spy
            steps = list(Gn.generate((None, tm, nm), bars, styles))
        finally:
            np.random.random = real_random
            Gn.MusicGeneration = orig_MG
        after = np.random.random_sample(4)       # pins the RNG position after generation
        rolls = np.array(steps)                  # [steps, G, N, 3] float64
        out[f"{tag}_rolls"] = rolls
        out[f"{tag}_styles"] = np.array(styles)
        out[f"{tag}_trace"] = np.array([g.trace for g in gens])   # [G, steps, 2]
        out[f"{tag}_time_digests"] = np.array(tm.digests)
        out[f"{tag}_note_digests"] = np.array(nm.digests)
        out[f"{tag}_rng_after"] = after
        meta[tag] = dict(seed=seed, bars=bars, draws=draws[0],
                         time_shapes=[list(map(list, tm.shapes[0]))], time_dtypes=tm.dtypes[0],
                         note_shapes=[list(map(list, nm.shapes[0]))], note_dtypes=nm.dtypes[0],
                         n_time_calls=len(tm.digests), n_note_calls=len(nm.digests),
                         roll_dtype=str(rolls.dtype))
    return out, meta


def main():
    codec = codec_golden()
    with open(os.path.join(HERE, "codec.json"), "w") as f:
        json.dump(codec, f)
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **dataset_golden())
    np.savez_compressed(os.path.join(HERE, "archive.npz"), **archive_golden())
    np.savez_compressed(os.path.join(HERE, "temperature.npz"), **temperature_golden())
    gen, meta = generate_golden()
    np.savez_compressed(os.path.join(HERE, "generate.npz"), **gen)
    with open(os.path.join(HERE, "generate_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("golden written:", sorted(os.listdir(HERE)))
    print(json.dumps(meta, indent=1)[:1500])


if __name__ == "__main__":
    main()
