"""Synthetic piano-roll batches: the workload bench.py and the tests use
in place of MIDI data (no dataset in the image).  Mirrors what dataset.load_all
feeds Model.fit (reference dataset.py:28-37,76): notes = roll[:, :T], target = chosen
= roll[:, 1:], beat = one_hot(t mod 16), style = one_hot(style id)."""
import numpy as np


def synthetic_batch(num_notes, time_steps, batch, seed=0, notes_per_bar=16, num_styles=23):
    """-> notes, chosen, beat, style, target (float32 numpy).  MT19937 streams are
    stable across NumPy versions, so every rank / the CPU baseline see the same data."""
    T, N, B = time_steps, num_notes, batch
    rs = np.random.RandomState(seed)
    play = rs.random_sample((B, T + 1, N)) < 0.05
    replay = play & (rs.random_sample((B, T + 1, N)) < 0.2)
    vol = play * rs.uniform(0.2, 1.0, (B, T + 1, N))
    roll = np.stack([play, replay, vol], axis=-1).astype(np.float32)
    t0 = rs.randint(0, notes_per_bar, size=B)
    beat = np.zeros((B, T, notes_per_bar), np.float32)
    tt = (t0[:, None] + np.arange(T)[None, :]) % notes_per_bar
    beat[np.arange(B)[:, None], np.arange(T)[None, :], tt] = 1.0
    style = np.zeros((B, T, num_styles), np.float32)
    style[np.arange(B), :, np.arange(B) % num_styles] = 1.0
    target = roll[:, 1:].copy()
    return roll[:, :T].copy(), target.copy(), beat, style, target
