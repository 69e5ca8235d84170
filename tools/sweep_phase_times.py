"""Dev tool (GPU box): cycles per phase of one step of the NOTE-axis sweeps -- lstm_fwd_fused_kernel<bf16,128> (both layers
summed) and lstm_bwd_kernel<bf16,128,DX=1|2> (both summed) -- from a dev build with s_memtime stamps at their phase
boundaries (tools/experiments/note_sweeps_phase_times_instrumentation.diff; workgroup 7, waves 0 and 2):
    DEEPJ_LIB=.../libdeepj_hip.phase.so python tools/note_phase_times.py"""
import ctypes as C, os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import quick_bench as q
from music_generator_amd import _lib
lib = _lib.load()
out = (C.c_ulonglong * 48)()
lib.dj_debug_phase_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
with contextlib.redirect_stdout(io.StringIO()):
    q.run("bf16", steps=3)
torch.cuda.synchronize()
assert lib.dj_debug_phase_read(out, 1) == 0
steps = 6
with contextlib.redirect_stdout(io.StringIO()):
    q.run("bf16", steps=steps)
torch.cuda.synchronize()
assert lib.dj_debug_phase_read(out, 0) == 0
nf = ["loop edge", "products (x W, h U)", "x request + gate math + h -> LDS", "stash stores (c, gate codes)",
      "x tile -> LDS", "barrier", "h tile -> HBM"]
nb = ["loop edge", "dH -> LDS, stash requests, first barrier", "gate math + dz -> LDS", "second barrier", "dz tile -> HBM",
      "fused dX product + stores", "dz U^T product (stationary)"]
n256 = ["loop edge", "top: stash requests, first barrier", "finish (six operations per cell, dz -> LDS)", "second barrier",
        "product + factors of the step before + row stores", "-", "-"]
for base, names, what in ((0, nf, "lstm_fwd_fused_kernel<bf16,128>"), (16, nb, "lstm_bwd_kernel<bf16,128>"),
                          (32, n256, "lstm_bwd256_kernel")):
    for off, who in ((0, "wave 0"), (8, "a later wave")):
        v = [out[base + off + k] for k in range(7)]
        tot = sum(v)
        print(what, who, "cycles per recurrence step (both layers averaged): %.0f" % (tot / (steps + 1.0) / 2 / 128))
        for k in range(7):
            print("  %-48s %6.1f %%  %8.0f" % (names[k], 100.0 * v[k] / max(tot, 1), v[k] / (steps + 1.0) / 2 / 128))
