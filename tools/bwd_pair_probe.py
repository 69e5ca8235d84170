"""Dev tool: the H = 256 BPTT sweep alone at the BASELINE shape (256 tiles x 128 steps): per-tile kernel of libdeepj_hip.so
(0) vs the experiments of tools/bwd_decompositions (sh tools/bwd_decompositions/build.sh): workgroup pairs (1), two tiles
per pair interleaved (2).  HIP-event times, and how many pairs had both members on one compute unit (fault word 3).
    python tools/bwd_pair_probe.py [tiles] [steps]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music_generator_amd import _lib as L

lib = L.load()
exp = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bwd_decompositions", "libdeepj_bwd_exp.so"))
_P = C.c_void_p
for _n in ("dj_lstm_bwd_pair", "dj_lstm_bwd_dual"):
    getattr(exp, _n).restype = C.c_int32
    getattr(exp, _n).argtypes = [C.c_int32] * 4 + [_P] * 5 + [C.c_int64, _P, C.c_int32, _P, _P]
exp.dj_bwd_exp_scratch_bytes.restype = C.c_int64
dev = torch.device("cuda:0")
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H, R = 256, tiles * steps * 32
g = torch.Generator().manual_seed(1)
U = torch.randn(H, 4 * H, generator=g) / 16
upb = torch.empty(H * 4 * H * 2, dtype=torch.uint8, device=dev)
st = None
L.check(lib.dj_lstm_pack(1, H, L.ptr(U.to(dev)), None, L.ptr(upb), st), "pack")
Z = torch.randint(0, 256, (R * 4 * H,), dtype=torch.uint8, device=dev)
Cc = (torch.randn(R * H, device=dev) * 0.7).to(torch.bfloat16)
dH = (torch.randn(R, H, device=dev) * 0.1).to(torch.bfloat16)
cts = R * 256
dZ = torch.zeros(4 * cts, dtype=torch.bfloat16, device=dev)
db = torch.zeros(4 * H, dtype=torch.float32, device=dev)
cl = torch.zeros(exp.dj_bwd_exp_scratch_bytes(), dtype=torch.uint8, device=dev)

def run(pair):
    if pair == 2:
        L.check(exp.dj_lstm_bwd_dual(1, H, tiles, steps, L.ptr(Z), L.ptr(upb), L.ptr(Cc), L.ptr(dH), L.ptr(dZ), cts, L.ptr(db), 0, L.ptr(cl), st), "dual")
    elif pair:
        L.check(exp.dj_lstm_bwd_pair(1, H, tiles, steps, L.ptr(Z), L.ptr(upb), L.ptr(Cc), L.ptr(dH), L.ptr(dZ), cts, L.ptr(db), 0, L.ptr(cl), st), "pair")
    else:
        L.check(lib.dj_lstm_bwd(1, H, tiles, steps, L.ptr(Z), L.ptr(upb), L.ptr(Cc), L.ptr(dH), L.ptr(dZ), cts, L.ptr(db), 0, st), "bwd")

for pair in (0, 1, 2, 0, 1, 2):
    for _ in range(3):
        run(pair)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(pair)
    e1.record()
    torch.cuda.synchronize()
    words = cl[16384:16384 + 16].view(torch.int32).cpu().tolist()
    print(f"pair={pair}: {e0.elapsed_time(e1) / 10:.3f} ms per launch; fault words {words[:3]}, same-CU pairs (cumulative) {words[3]}", flush=True)
