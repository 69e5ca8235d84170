"""Dev tool (GPU box): throughput of the Keras-surface Model.fit at the BASELINE shape with HOST inputs
(float64 NumPy arrays, as the reference's dataset.py produces) -- the PCIe- and host-inclusive rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from music_generator_amd.data import synthetic_batch
from music_generator_amd.engine import DeepJConfig
from music_generator_amd.model import build_models
N, T, B, nb = 128, 128, 64, 6
notes, chosen, beat, style, target = [np.concatenate([a] * nb).astype(np.float64) for a in synthetic_batch(N, T, B, seed=0)]
model = build_models(time_steps=T, config=DeepJConfig(num_notes=N, time_steps=T, dtype="bf16"), dtype="bf16")[0]
model.fit([notes, target, beat, style], [target], epochs=1, batch_size=B, verbose=0)      # warm-up
t0 = time.time()
model.fit([notes, target, beat, style], [target], epochs=8, batch_size=B, verbose=0)
dt = time.time() - t0
steps = 8 * nb
print(f"fit: {dt / steps * 1e3:.1f} ms/batch, {steps * B * T * N / dt / 1e6:.1f} M note-steps/s (host float64 inputs, B={B})")
