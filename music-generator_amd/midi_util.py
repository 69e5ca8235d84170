"""Piano-roll <-> MIDI codec with the reference's function names and semantics
(/root/reference/midi_util.py:9-210), written against this package's own SMF module.

A roll is float [L, classes, 3] = (play, replay, volume).  The behaviours below are
pinned by the golden vectors captured from the reference (tests/golden/codec.json):
 * encode emits events only on ticks whose play row differs from the previous tick, so
   a replay flag on an otherwise unchanged row is dropped (midi_util.py:35,57-71);
   velocity = int(volume * 127) truncates (midi_util.py:43,66);
 * decode down-samples by `step` ticks: replay = any(replay) and volume = max(volume)
   over the window; a re-struck held note sets replay and keeps the OLD volume, but only
   if the previous raw tick is still in the window buffer (midi_util.py:118-148).
"""
import os

import numpy as np

from . import smf as midi
from .constants import *  # noqa: F401,F403


def midi_encode(note_seq, resolution=NOTES_PER_BEAT, step=1):
    """Piano roll -> single-track Pattern (reference midi_util.py:9-95)."""
    roll = np.asarray(note_seq)
    play, replay, volume = roll[:, :, 0], roll[:, :, 1], roll[:, :, 2]
    track = midi.Track()
    pattern = midi.Pattern([track], resolution=resolution)

    held = np.zeros_like(play[0])        # play row of the previous tick
    last_tick = 0                        # tick of the last emitted event
    idle = 0                             # ticks since the last change (EndOfTrack delta)

    def emit(evt_cls, tick, pitch, **kw):
        nonlocal last_tick
        track.append(evt_cls(tick=(tick - last_tick) * step, pitch=pitch, **kw))
        last_tick = tick

    tick = -1
    for tick in range(len(play)):
        row = np.array(play[tick])
        if np.array_equal(held, row):
            idle += 1
        else:
            idle = 0
            for pitch in range(row.shape[0]):
                now_on, was_on = row[pitch] > 0, held[pitch] > 0
                if now_on and held[pitch] == 0:
                    emit(midi.NoteOnEvent, tick, pitch, velocity=int(volume[tick][pitch] * MAX_VELOCITY))
                elif was_on and row[pitch] == 0:
                    emit(midi.NoteOffEvent, tick, pitch)
                elif was_on and now_on and replay[tick][pitch] > 0:
                    emit(midi.NoteOffEvent, tick, pitch)
                    track.append(midi.NoteOnEvent(tick=0, pitch=pitch,
                                                  velocity=int(volume[tick][pitch] * MAX_VELOCITY)))
        held = row
    tick += 1
    for pitch in range(held.shape[0]):   # release whatever is still sounding
        if held[pitch] > 0:
            emit(midi.NoteOffEvent, tick, pitch)
            idle = 0
    track.append(midi.EndOfTrackEvent(tick=idle))
    return pattern


class _Downsampler:
    """The reference's replay/volume window buffers (midi_util.py:112-136) as explicit state:
    `n` raw ticks are buffered; all but the newest are folded into (rep_sum, vol_max)."""

    def __init__(self, classes, step):
        self.step = step
        self.n = 1
        self.vol = np.zeros((classes,))       # newest raw row
        self.rep = np.zeros((classes,))
        self.prev_vol = None                  # row before the newest, while it is still buffered
        self.first_vol = self.vol             # oldest buffered row (aliases vol while n == 1)
        self.rep_sum = np.zeros((classes,))
        self.vol_max = np.zeros((classes,))
        self.replay_out, self.volume_out = [], []

    def advance(self):
        """One raw tick passes: freeze the newest row, start a copy of it."""
        self.rep_sum = self.rep_sum + self.rep
        self.vol_max = np.maximum(self.vol_max, self.vol)
        self.prev_vol = self.vol
        self.vol = self.vol.copy()
        self.rep = np.zeros_like(self.rep)
        self.n += 1
        if self.n > self.step:
            self.replay_out.append(np.minimum(self.rep_sum, 1))
            self.volume_out.append(self.vol_max)
            self.n = 1
            self.prev_vol = None
            self.first_vol = self.vol
            self.rep_sum = np.zeros_like(self.rep)
            self.vol_max = np.zeros_like(self.vol)

    def note_on(self, pitch, velocity):
        self.vol[pitch] = velocity / MAX_VELOCITY
        if self.prev_vol is not None and self.prev_vol[pitch] > 0 and self.vol[pitch] > 0:
            self.rep[pitch] = 1                # re-strike of a held note
            self.vol[pitch] = self.prev_vol[pitch]

    def note_off(self, pitch):
        self.vol[pitch] = 0

    def finish(self):
        self.replay_out.append(np.minimum(self.rep_sum + self.rep, 1))
        self.volume_out.append(self.first_vol)
        return np.array(self.replay_out), np.array(self.volume_out)


def midi_decode(pattern, classes=MIDI_MAX_NOTES, step=None):
    """Pattern -> piano roll [L, classes, 3] (reference midi_util.py:97-191)."""
    if step is None:
        step = pattern.resolution // NOTES_PER_BEAT
    merged_replay = merged_volume = None
    for track in pattern:
        ds = _Downsampler(classes, step)
        for event in track:
            for _ in range(event.tick):
                ds.advance()
            if isinstance(event, midi.EndOfTrackEvent):
                break
            if isinstance(event, midi.NoteOnEvent):
                ds.note_on(event.data[0], event.data[1])
            elif isinstance(event, midi.NoteOffEvent):
                ds.note_off(event.data[0])
        replay, volume = ds.finish()
        if merged_volume is None:
            merged_replay, merged_volume = replay, volume
        else:                                  # sum tracks, zero-padding the shorter one
            if len(volume) > len(merged_volume):
                replay, merged_replay = merged_replay, replay
                volume, merged_volume = merged_volume, volume
            pad = ((0, len(merged_volume) - len(volume)), (0, 0))
            merged_replay = merged_replay + np.pad(replay, pad, 'constant')
            merged_volume = merged_volume + np.pad(volume, pad, 'constant')
    roll = np.stack([np.ceil(merged_volume), merged_replay, merged_volume], axis=2)
    return np.minimum(roll, 1)                 # stacked duplicate notes saturate at 1


def load_midi(fname):
    """Decode a .mid file with an .npy cache under CACHE_DIR (reference midi_util.py:193-210)."""
    cache_path = os.path.join(CACHE_DIR, fname + '.npy')
    try:
        note_seq = np.load(cache_path)
    except Exception:
        note_seq = midi_decode(midi.read_midifile(fname))
        os.makedirs(os.path.dirname(cache_path), exist_ok=True)
        np.save(cache_path, note_seq)
    assert note_seq.ndim == 3 and note_seq.shape[1:] == (MIDI_MAX_NOTES, 3), note_seq.shape
    assert (note_seq >= 0).all() and (note_seq <= 1).all()
    return note_seq
