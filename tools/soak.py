"""Dev tool (GPU box): 400 training steps at the BASELINE shape; loss must stay finite and fall, no cluster wait may expire."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np, torch
from tools import quick_bench as q
from music_generator_amd import _lib
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    eng = q.run("bf16", steps=400)
out = buf.getvalue().strip().splitlines()[-1]
i = out.index("losses ")
losses = eval(out[i + 7:])
print(out[:out.index("losses")])
print("first", losses[:3], "last", losses[-3:], "finite", all(np.isfinite(losses)), "monotone-ish", losses[-1] < losses[0])
print("cluster faults:", eng.cluster_faults())
