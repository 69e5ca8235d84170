"""Autoregressive sampling with the reference's surface (reference generate.py:13-153):
MusicGeneration, apply_temperature, process_inputs, generate, write_file, main.

Per generated time step: the sliding 128-step window goes through `time_model` (stateless,
from zero state -- an incrementally carried time-axis state would NOT be equivalent,
SURVEY.md a-G), then the notes are sampled low to high, each conditioned on the notes
already chosen.  Random draws come from NumPy's global MT19937 stream in the reference's
order (note-major, piece-minor; the replay draw only happens after a successful play
draw), so a seeded run reproduces the reference's sampled rolls given equal model outputs.
"""
import argparse
import os
from collections import deque

import numpy as np

from . import smf as midi
from .constants import *  # noqa: F401,F403
from .dataset import compute_beat, compute_genre, unclamp_midi
from .midi_util import midi_encode
from .util import build_or_load, one_hot

try:
    from tqdm import tqdm
except Exception:                                  # pragma: no cover
    def tqdm(x, **kw):
        return x


class MusicGeneration:
    """State of one piece being generated (reference generate.py:13-79)."""

    def __init__(self, style, default_temp=1):
        def window(item):
            return deque([item() for _ in range(SEQ_LEN)], maxlen=SEQ_LEN)

        self.notes_memory = window(lambda: np.zeros((NUM_NOTES, NOTE_UNITS)))
        self.beat_memory = window(lambda: np.zeros(NOTES_PER_BAR))
        self.style_memory = window(lambda: style)
        self.next_note = np.zeros((NUM_NOTES, NOTE_UNITS))    # the time step under construction
        self.silent_time = NOTES_PER_BAR
        self.results = []
        self.default_temp = default_temp
        self.temperature = default_temp

    def build_time_inputs(self):
        return np.array(self.notes_memory), np.array(self.beat_memory), np.array(self.style_memory)

    def build_note_inputs(self, note_features):
        # one time step only: [1, N, units], [1, N, 3], [1, styles]
        return np.array(note_features), np.array([self.next_note]), np.array(list(self.style_memory)[-1:])

    def choose(self, prob, n):
        """Sample note n from (p_play, p_replay, volume) (generate.py:47-58)."""
        vol = prob[n, -1]
        p = apply_temperature(prob[n, :-1], self.temperature)
        if np.random.random() <= p[0]:
            self.next_note[n, 0] = 1
            self.next_note[n, 2] = vol
            if np.random.random() <= p[1]:
                self.next_note[n, 1] = 1

    def end_time(self, t):
        """Close the time step: temperature schedule + window update (generate.py:60-79)."""
        if np.count_nonzero(self.next_note) == 0:
            self.silent_time += 1
            if self.silent_time >= NOTES_PER_BAR:
                self.temperature += 0.1
        else:
            self.silent_time = 0
            self.temperature = self.default_temp
        done = self.next_note
        self.notes_memory.append(done)
        self.beat_memory.append(compute_beat(t, NOTES_PER_BAR))
        self.results.append(done)
        self.next_note = np.zeros((NUM_NOTES, NOTE_UNITS))
        return done


def apply_temperature(prob, temperature):
    """Rescale sigmoid probabilities: sigmoid(logit(p) / T); identity at T == 1
    (generate.py:81-91).  dtype follows `prob` (float32 from predict)."""
    if temperature != 1:
        logit = -np.log(1 / prob - 1)
        prob = 1 / (1 + np.exp(-logit / temperature))
    return prob


def process_inputs(ins):
    """[(a0, b0, c0), (a1, b1, c1), ...] -> [array(a*), array(b*), array(c*)] (generate.py:93-96)."""
    return [np.array(col) for col in zip(*ins)]


# time steps per device launch batch of the resident path: the host work of a batch (two state read-backs, the pool
# upload, the per-run constants, the result copy) is not overlapped with the device, so it is spread over four bars
GEN_CHUNK = 4 * NOTES_PER_BAR

# Filled by the fused paths of generate(): how many Bernoulli draws of the last run fell within 1e-5 of the
# probability they were compared with, and the time step of the first one (-1 = none).  Zero near ties certifies
# the sampled notes against any model whose probabilities agree with the HIP model's to 1e-5 (DESIGN.md
# "Sampling parity"); otherwise the rolls are certified up to `first_near_tie_step`.
# `max_temperature` / `silent_steps`: the heating schedule of end_time() as the run saw it (host MusicGeneration state).
last_run_stats = {"draws": 0, "near_ties": 0, "first_near_tie_step": -1, "max_temperature": 1.0, "silent_steps": 0}
repeated_steps = 0      # step-wise steps computed twice because of a cluster fault (DEEPJ_GENERATE_RETRY=1 only)


def _note_schedule(pieces):
    """Record the temperature schedule after a time step (every path calls it right after end_time)."""
    last_run_stats["max_temperature"] = max([last_run_stats["max_temperature"]] + [float(g.temperature) for g in pieces])
    last_run_stats["silent_steps"] += sum(1 for g in pieces if g.silent_time > 0)


def _fused_engine(models, n_pieces):
    """The fused MI355X path is used when the models are this package's HIP models; any other
    duck-typed models (or DEEPJ_GENERATE_SLOW=1) take the reference loop over predict() below."""
    if os.environ.get("DEEPJ_GENERATE_SLOW") or n_pieces > 8:
        return None
    shared = getattr(models[1], "_s", None)
    if shared is None or getattr(shared.backend, "name", "") != "hip":
        return None
    if shared.cfg.note_axis_units > 256 or shared.cfg.time_axis_units + 3 > 512:
        return None                                # wider than the single-workgroup sampler (dj_gen.hip): predict() loop
    return shared, shared.engine(n_pieces, SEQ_LEN, train=False)


def _draw_ahead(n):
    """n uniforms from NumPy's global stream WITHOUT consuming them."""
    state = np.random.get_state()
    u = np.random.random_sample(n)
    np.random.set_state(state)
    return u


def _fused_step(shared, engine, pieces):
    """All N notes of one time step for every piece in one device call (dj_generate_step).  The
    device consumes pre-drawn uniforms in the reference's order; NumPy's global stream is then
    advanced by exactly the number consumed, so its position matches the reference loop."""
    import torch
    be = shared.backend
    notes, beat, style = process_inputs([g.build_time_inputs() for g in pieces])
    u_dev = torch.as_tensor(_draw_ahead(2 * NUM_NOTES * len(pieces)), dtype=torch.float64).to(be.device)
    temps = be.tensor(np.array([g.temperature for g in pieces], np.float32))
    args = (shared.params, be.tensor(notes), be.tensor(beat), be.tensor(style), u_dev, temps)
    nxt, used = engine.generate_step(*args)
    nxt = be.numpy(nxt)
    used = used.cpu().numpy()
    if hasattr(engine, "raise_on_cluster_faults"):
        # NaN-poisoned rows would be sampled as silence: never used.  A cluster fault RAISES, with the description of
        # the first expired wait (engine.describe_fault_report).  Until round 4 the step was silently computed again
        # once; that hid a fault instead of explaining it and is now opt-in hang protection (DEEPJ_GENERATE_RETRY=1:
        # the step has no device-side state -- windows, temperatures and uniforms are its inputs -- so it can be
        # repeated; every repeat is printed, counted in `repeated_steps` and recorded in engine.FAULT_LOG).
        if os.environ.get("DEEPJ_GENERATE_RETRY") == "1" and engine.cluster_faults("generation (step repeated)"):
            print("[deepj] generation: cluster fault in a time step; repeating the step (DEEPJ_GENERATE_RETRY=1)",
                  flush=True)
            global repeated_steps
            repeated_steps += 1
            nxt, used = engine.generate_step(*args)
            nxt = be.numpy(nxt)
            used = used.cpu().numpy()
        engine.raise_on_cluster_faults("generation")
    np.random.random_sample(int(used[0]))
    last_run_stats["draws"] += int(used[0])
    last_run_stats["near_ties"] += int(used[1])
    for i, g in enumerate(pieces):
        g.next_note[:, :] = nxt[i]


def _generate_resident(shared, engine, pieces, total_steps):
    """Device-resident run: windows / temperature schedule / draw offset stay in HBM and the
    step is replayed from a hipGraph (engine.ResidentGeneration).  The host mirrors every step
    into the MusicGeneration objects (same end_time logic), GEN_CHUNK steps at a time."""
    from .engine import ResidentGeneration
    run = ResidentGeneration(engine, shared.params, [g.style_memory[-1] for g in pieces],
                             default_temp=pieces[0].default_temp, steps_cap=total_steps,
                             use_graph=not os.environ.get("DEEPJ_GENERATE_NOGRAPH"))
    t = 0
    while t < total_steps:
        k = min(GEN_CHUNK, total_steps - t)
        notes, used = run.run(k, _draw_ahead(2 * NUM_NOTES * len(pieces) * k))
        spent = 0
        for j in range(k):
            # advance the real stream step by step: one draw per note + one per played note, so the
            # stream position at every yield equals the reference loop's
            d = NUM_NOTES * len(pieces) + int(notes[j, :, :, 0].sum())
            np.random.random_sample(d)
            spent += d
            for i, g in enumerate(pieces):
                g.next_note[:, :] = notes[j, i]
            done = [g.end_time(t + j) for g in pieces]
            _note_schedule(pieces)
            yield done
        assert spent == used, (spent, used)
        last_run_stats["draws"] += spent
        t += k
        # device schedule (gen_state_kernel: float64 temperature, silent_time) == host schedule, after EVERY chunk:
        # both add 0.1 in float64 the same number of times, so the values are bit-equal
        st = run.last_state
        last_run_stats.update(near_ties=int(st["near_ties"]), first_near_tie_step=int(st["first_near_step"]))
        for i, g in enumerate(pieces):
            if float(st["temperature"][i]) != float(g.temperature) or int(st["silent"][i]) != g.silent_time:
                raise RuntimeError("device temperature schedule diverged from the host mirror at step %d, piece %d: "
                                   "device (T %.17g, silent %d) vs host (T %.17g, silent %d)"
                                   % (t, i, st["temperature"][i], st["silent"][i], g.temperature, g.silent_time))


def generate(models, num_bars, styles):
    """Generator over time steps; yields the list of per-piece note arrays [N, 3]
    (reference generate.py:98-121)."""
    print('Generating with styles:', styles)
    _, time_model, note_model = models
    pieces = [MusicGeneration(style) for style in styles]
    last_run_stats.update(draws=0, near_ties=0, first_near_tie_step=-1, max_temperature=1.0, silent_steps=0)
    fused = _fused_engine(models, len(pieces))
    if fused is not None and not os.environ.get("DEEPJ_GENERATE_STEPWISE"):
        yield from tqdm(_generate_resident(fused[0], fused[1], pieces, NOTES_PER_BAR * num_bars),
                        total=NOTES_PER_BAR * num_bars)
        return
    for t in tqdm(range(NOTES_PER_BAR * num_bars)):
        if fused is not None:
            ties = last_run_stats["near_ties"]
            _fused_step(fused[0], fused[1], pieces)
            if ties == 0 and last_run_stats["near_ties"]:
                last_run_stats["first_near_tie_step"] = t
            done = [g.end_time(t) for g in pieces]
            _note_schedule(pieces)
            yield done
            continue
        # note-invariant features of the whole window, last step only
        feats = np.array(time_model.predict(process_inputs([g.build_time_inputs() for g in pieces])))[:, -1:, :]
        for n in range(NUM_NOTES):
            ins = process_inputs([g.build_note_inputs(feats[i, :, :, :]) for i, g in enumerate(pieces)])
            pred = np.array(note_model.predict(ins))
            for i, g in enumerate(pieces):
                g.choose(pred[i][-1], n)
        done = [g.end_time(t) for g in pieces]
        _note_schedule(pieces)
        yield done


def write_file(name, results):
    """One .mid per generated piece under SAMPLES_DIR (generate.py:123-134)."""
    for i, result in enumerate(zip(*list(results))):
        fpath = os.path.join(SAMPLES_DIR, name + '_' + str(i) + '.mid')
        print('Writing file', fpath)
        os.makedirs(os.path.dirname(fpath), exist_ok=True)
        midi.write_midifile(fpath, midi_encode(unclamp_midi(result)))


def main(argv=None):
    parser = argparse.ArgumentParser(description='Generates music.')
    parser.add_argument('--bars', default=32, type=int, help='Number of bars to generate')
    parser.add_argument('--styles', default=None, type=int, nargs='+', help='Styles to mix together')
    args = parser.parse_args(argv)
    models = build_or_load()
    if args.styles:
        style_vecs = [np.mean([one_hot(i, NUM_STYLES) for i in args.styles], axis=0)]
    else:
        style_vecs = [compute_genre(i) for i in range(len(genre))]
    write_file('output', generate(models, args.bars, style_vecs))


if __name__ == '__main__':
    main()
