"""Generation throughput (dev tool): fused dj_generate_step, G=3 genre styles, N=48."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from music_generator_amd import generate as Gn
from music_generator_amd.dataset import compute_genre
from music_generator_amd.model import build_models

def run(dtype, bars, slow=False):
    if slow: os.environ["DEEPJ_GENERATE_SLOW"] = "1"
    else: os.environ.pop("DEEPJ_GENERATE_SLOW", None)
    models = build_models(dtype=dtype, seed=5)
    styles = [compute_genre(i) for i in range(3)]
    np.random.seed(0)
    g = Gn.generate(models, bars, styles)
    # the resident path computes GEN_CHUNK steps per device batch and yields them one by one: time whole
    # chunks only (skip the first chunk, which carries the graph capture), so yields == computed steps
    skip = Gn.GEN_CHUNK if not slow else 1
    for _ in range(skip): next(g)
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    for _ in g: n += 1
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"[{dtype}{' slow' if slow else ''}] {n} steps: {dt/n*1e3:.2f} ms/step, {3*48*n/dt:.0f} notes/s", flush=True)

if __name__ == "__main__":
    run("f32", 12); run("bf16", 12); run("bf16", 1, slow=True)
