"""MIDI files -> training windows (reference dataset.py:14-88): clamp to the 48-note
range, prefix `time_steps` zero frames, cut a window every bar, inputs X and targets Y
offset by one step; beat = one-hot position in the bar, style = one-hot composer id."""
import multiprocessing
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .constants import *  # noqa: F401,F403
from .midi_util import load_midi
from .util import get_all_files, one_hot


def compute_beat(beat, notes_in_bar):
    return one_hot(beat % notes_in_bar, notes_in_bar)


def compute_completion(beat, len_melody):
    return np.array([beat / len_melody])


def compute_genre(genre_id):
    """Uniform mixture over the composer styles of one genre (dataset.py:20-26)."""
    first = sum(len(s) for s in styles[:genre_id])
    count = len(styles[genre_id])
    vec = np.zeros((NUM_STYLES,))
    vec[first:first + count] = 1 / count
    return vec


def stagger(data, time_steps):
    """Windows of `time_steps` frames starting every NOTES_PER_BAR frames over the
    sequence prefixed with `time_steps` zero frames; Y is X shifted by one frame
    (dataset.py:28-37)."""
    padded = [np.zeros_like(data[0])] * time_steps + list(data)
    starts = range(0, len(padded) - time_steps, NOTES_PER_BAR)
    xs = [padded[i:i + time_steps] for i in starts]
    ys = [padded[i + 1:i + time_steps + 1] for i in starts]
    return xs, ys


def clamp_midi(sequence):
    """Keep MIDI notes MIN_NOTE..MAX_NOTE-1 (dataset.py:78-82)."""
    return sequence[:, MIN_NOTE:MAX_NOTE, :]


def unclamp_midi(sequence):
    """Inverse placement: MIN_NOTE silent notes below, nothing added above (dataset.py:84-88)."""
    return np.pad(sequence, ((0, 0), (MIN_NOTE, 0), (0, 0)), 'constant')


def load_all(styles, batch_size, time_steps):
    """-> ([note_data, note_target, beat_data, style_data], [note_target]) exactly as the
    reference feeds Model.fit (dataset.py:39-76).  `batch_size` is unused there too."""
    notes, targets, beats, style_vecs = [], [], [], []
    flat_styles = [d for group in styles for d in group]
    workers = max(1, multiprocessing.cpu_count())
    for style_id, style_dir in enumerate(flat_styles):
        hot = one_hot(style_id, NUM_STYLES)
        files = get_all_files([style_dir])
        with ThreadPoolExecutor(max_workers=workers) as pool:     # order-preserving, like joblib threads
            seqs = list(pool.map(load_midi, files))
        for seq in seqs:
            if len(seq) < time_steps:
                continue
            seq = clamp_midi(seq)
            x, y = stagger(seq, time_steps)
            notes += x
            targets += y
            beats += stagger([compute_beat(i, NOTES_PER_BAR) for i in range(len(seq))], time_steps)[0]
            style_vecs += stagger([hot for _ in range(len(seq))], time_steps)[0]
    note_target = np.array(targets)
    return [np.array(notes), note_target, np.array(beats), np.array(style_vecs)], [note_target]
