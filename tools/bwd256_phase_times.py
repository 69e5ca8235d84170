"""Dev tool (GPU box): cycles per phase of one step of the split-gate-math H = 256 BPTT sweep (lstm_bwd256_kernel), from a
dev build with s_memtime stamps at its phase boundaries (tools/experiments/bwd256_phase_times_instrumentation.diff):
    DEEPJ_LIB=.../libdeepj_hip.phase.so [DEEPJ_BWD256_VARIANT=n] python tools/bwd256_phase_times.py"""
import ctypes as C, os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import quick_bench as q
from music_generator_amd import _lib
lib = _lib.load()
out = (C.c_ulonglong * 16)()
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
names = ["loop edge", "top: dH -> LDS, stash requests, first barrier", "finish (six operations per cell, dz -> LDS)",
         "second barrier", "product + factors of the step before", "dz tile -> HBM"]
for base, who in ((0, "wave 0"), (8, "a later wave")):
    v = [out[base + k] for k in range(6)]
    tot = sum(v)
    print(who, "total cycles", tot, "variant", os.environ.get("DEEPJ_BWD256_VARIANT", "0"))
    for k in range(6):
        print("  %-48s %6.1f %%  %8.0f cycles per launch-step" % (names[k], 100.0 * v[k] / max(tot, 1), v[k] / (steps + 1.0) / 2 / 128))
