"""Dev tool (GPU box): where the cycles of one step of the H = 256 BPTT sweep go.
    git apply tools/experiments/bwd_phase_times_instrumentation.diff && tools/build_variant.sh phase -DDJ_PHASE_TIMES=1 \
        && git checkout music-generator_amd/csrc/dj_lstm.hip
    DEEPJ_LIB=$PWD/music-generator_amd/lib/libdeepj_hip.phase.so python tools/bwd_phase_times.py
The variant adds s_memtime reads at the phase boundaries of lstm_bwd_kernel<bf16, 256> (workgroup 7: wave 0 and wave 5)
and sums the differences; this runs a few training steps at the BASELINE shape and prints cycles per recurrence step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import quick_bench as q
from music_generator_amd import _lib
lib = _lib.load()
out = (C.c_ulonglong * 16)()
lib.dj_debug_phase_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    q.run("bf16", steps=3)
torch.cuda.synchronize()
assert lib.dj_debug_phase_read(out, 1) == 0
steps = 6                                     # quick_bench: warm-up + timed steps
with contextlib.redirect_stdout(io.StringIO()):
    q.run("bf16", steps=steps)
torch.cuda.synchronize()
assert lib.dj_debug_phase_read(out, 0) == 0
names = ["loop edge + previous tail", "dH staging -> first barrier", "gate math (incl. stash wait)", "second barrier",
         "dz tile -> HBM", "U^T product (ring from L2)"]
for base, who in ((0, "wave 0"), (8, "wave 5")):
    v = [out[base + k] for k in range(6)]
    tot = sum(v)
    print(who, "total cycles", tot)
    for k in range(6):
        print("  %-32s %6.1f %%" % (names[k], 100.0 * v[k] / max(tot, 1)))
