"""Dev tool (GPU box): 4096 generated time steps (bf16, 3 pieces, hipGraph replay of the prepared step with the
wavefront / cooperative time-axis launch); no cluster wait may expire, the notes stay finite, throughput printed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from music_generator_amd import generate as Gn
from music_generator_amd.dataset import compute_genre
from music_generator_amd.model import build_models

models = build_models(dtype="bf16", seed=5)
styles = [compute_genre(i) for i in range(3)]
np.random.seed(0)
n, played = 0, 0.0
t0 = time.time()
for step in Gn.generate(models, 4096 // 16, styles):
    a = np.asarray(step)
    assert np.isfinite(a).all()
    played += a[..., 0].sum()
    n += 1
torch.cuda.synchronize()
dt = time.time() - t0
eng = models[0]._s.backend.engine if hasattr(models[0]._s, "backend") else None
print(f"{n} steps, {played:.0f} notes played, {dt / n * 1e3:.2f} ms/step incl. host, stats {Gn.last_run_stats}")
