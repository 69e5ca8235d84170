"""Dev tool: bf16 resident generation only (for rocprofv3 traces).  argv: dtype, number of pieces (default 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.gen_bench as gb
if len(sys.argv) > 2:
    n = int(sys.argv[2])
    _cg = gb.compute_genre
    gb.compute_genre = lambda i: _cg(i)
    _run = gb.run
    def run(dtype, bars, slow=False, _n=n):
        import numpy as np, torch, time
        from music_generator_amd import generate as Gn
        models = gb.build_models(dtype=dtype, seed=5)
        styles = [gb.compute_genre(i) for i in range(_n)]
        np.random.seed(0)
        for _ in Gn.generate(models, bars, styles):
            pass
        torch.cuda.synchronize()
    run(sys.argv[1], 12)
else:
    gb.run(sys.argv[1] if len(sys.argv) > 1 else "bf16", 12)
