"""Dev tool (GPU box, DJ_EXP_STAMP build): per-phase cycle stamps of one workgroup of the recurrent kernels."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from music_generator_amd import _lib
from tools.quick_bench import run
run("bf16", steps=1)
lib = _lib.load()
buf = np.zeros((2, 8, 1024), np.uint64)
lib.dj_debug_stamps.restype = C.c_int
assert lib.dj_debug_stamps(C.c_void_p(buf.ctypes.data)) == 0
for k, name, nph in ((0, "fwd (last fused/cluster launch with block 17)", 7), (1, "bwd", 8)):
    st = buf[k, :nph, :128].astype(np.int64)       # [phase][step]
    d = np.diff(st, axis=0)                         # phase durations within a step
    order = np.argsort(st[0])
    nxt = st[0][order][1:] - st[0][order][:-1]
    print(name, "step period (cycles @100MHz?)", np.median(nxt))
    print("  median phase deltas:", [float(np.median(d[i][1:-1])) for i in range(nph - 1)])
st = buf[0, :, :128].astype(np.int64)
if st[7].any():
    print("fwd: loads arrived after", float(np.median((st[7] - st[0])[1:-1])), "cycles of the first phase")
