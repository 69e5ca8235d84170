"""Deterministic duck-typed `.predict` objects.

Used (a) by tests/golden/make_golden.py to drive the REFERENCE's generate()
harness (generate.py:98-121) and capture golden traces, and (b) by the tests to
drive the build's own harness on the same inputs.  Cheap closed-form functions
of their inputs, causal along the note axis like the real note model.
"""
import numpy as np


def _digest(arrs):
    """Order-sensitive float64 digest of a list of arrays."""
    tot = 0.0
    for k, a in enumerate(arrs):
        a = np.asarray(a, np.float64).ravel()
        w = np.cos(0.37 * np.arange(a.size) + k)
        tot += float(a @ w)
    return tot


class FakeTimeModel:
    def __init__(self, units=256):
        self.units = units
        self.digests = []
        self.shapes = []
        self.dtypes = []

    def predict(self, ins):
        notes, beat, style = [np.asarray(a) for a in ins]
        self.shapes.append([a.shape for a in (notes, beat, style)])
        self.dtypes.append([str(a.dtype) for a in (notes, beat, style)])
        self.digests.append(_digest([notes, beat, style]))
        notes, beat, style = [a.astype(np.float64) for a in (notes, beat, style)]
        G, T, N, _ = notes.shape
        a = notes.sum(axis=(2, 3)) + beat.argmax(axis=2) + style @ np.arange(style.shape[-1])
        # weight recent history so the sliding window matters
        hist = np.zeros((G, T))
        for k in range(4):
            hist[:, k:] += a[:, :T - k] * (0.5 ** k)
        n = np.arange(N)[None, None, :, None]
        h = np.arange(self.units)[None, None, None, :]
        out = np.tanh(0.3 * hist[:, :, None, None] + 0.1 * n + 0.01 * h)
        out[..., 1] = np.sin(2.0 * hist)[:, :, None]          # silence gate, read by FakeNoteModel
        return out.astype(np.float32)


class FakeNoteModel:
    def __init__(self):
        self.digests = []
        self.shapes = []
        self.dtypes = []

    def predict(self, ins):
        feat, chosen, style = [np.asarray(a) for a in ins]
        self.shapes.append([a.shape for a in (feat, chosen, style)])
        self.dtypes.append([str(a.dtype) for a in (feat, chosen, style)])
        self.digests.append(_digest([feat, chosen, style]))
        feat, chosen, style = [a.astype(np.float64) for a in (feat, chosen, style)]
        G, _, N, _ = chosen.shape
        # causal: position n sees chosen[:n] only (model.py:101 shift)
        played = chosen[:, 0, :, 0] + 0.5 * chosen[:, 0, :, 1] + 0.25 * chosen[:, 0, :, 2]
        s = np.concatenate([np.zeros((G, 1)), np.cumsum(played, axis=1)[:, :-1]], axis=1)
        n = np.arange(N)[None, :]
        f0 = feat[:, 0, :, 0]
        sty = (style[:, 0, :] @ np.arange(style.shape[-1]))[:, None]
        gate = feat[:, 0, :, 1] > 0.3                          # silent steps -> temperature path
        lp = -1.2 + 0.9 * np.sin(0.7 * n + 3.0 * f0 + s + 0.1 * sty) - 8.0 * gate
        lr = 0.3 * np.cos(0.4 * n + s)
        vol = 0.5 + 0.3 * np.sin(0.2 * n + f0)
        out = np.stack([1 / (1 + np.exp(-lp)), 1 / (1 + np.exp(-lr)), vol], axis=-1)
        return out[:, None].astype(np.float32)
