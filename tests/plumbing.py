"""Shared by the CPU and GPU plumbing tests: BASELINE configs[0] -- "train.py on 4 synthetic random MIDI
files, batch=2, 8 timesteps, 1 epoch" -- i.e. .mid files on disk -> load_all -> Model.fit
(reference train.py:18-29, dataset.py:39-76)."""
import os

import numpy as np

FILES = [("data/baroque/bach", "a.mid", 40, 1), ("data/baroque/bach", "b.mid", 33, 2),
         ("data/classical/mozart", "c.mid", 24, 3), ("data/romantic/chopin", "d.mid", 56, 4)]


def random_roll(length, seed):
    """[L, 128, 3] piano roll with notes inside the model's 48-note range, velocities on the 1/127 grid the
    MIDI wire format can carry."""
    rs = np.random.RandomState(seed)
    roll = np.zeros((length, 128, 3))
    play = rs.random_sample((length, 48)) < 0.08
    vel = rs.randint(20, 128, size=(length, 48)) / 127.0
    roll[:, 36:84, 0] = play
    roll[:, 36:84, 2] = play * vel
    return roll


def write_corpus(root):
    """4 synthetic .mid files under root/data/<genre>/<composer>/ written with the package's own SMF writer."""
    from music_generator_amd import midi_util, smf
    paths = []
    for d, name, length, seed in FILES:
        os.makedirs(os.path.join(root, d), exist_ok=True)
        path = os.path.join(root, d, name)
        smf.write_midifile(path, midi_util.midi_encode(random_roll(length, seed)))
        paths.append(path)
    return paths
