"""Dev tool (GPU box): time the bf16 recurrent kernels against the number of sequence tiles, to see
whether a second workgroup per CU (tiles > 256) overlaps with the first."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music_generator_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
def t_ms(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for H in (128, 256):
    steps = 128
    U = torch.randn(H, 4 * H, device=dev) * 0.05
    upf = torch.empty(H * 4 * H * 2, dtype=torch.uint8, device=dev); upb = torch.empty_like(upf)
    _lib.check(lib.dj_lstm_pack(1, H, p(U), p(upf), p(upb), st()), "pack")
    for tiles in (tuple(int(x) for x in os.environ["REC_TILES"].split(",")) if os.environ.get("REC_TILES") else (128, 256, 257, 384, 512)):
        R = tiles * steps * 32
        Z = (torch.randn(R * 4 * H, device=dev) * 0.5).to(torch.bfloat16)           # x W + b, fragment-tiled
        G8 = torch.zeros(R * 4 * H, dtype=torch.uint8, device=dev)                 # 8-bit gate stash
        Hd = torch.zeros(R * H, dtype=torch.bfloat16, device=dev); Cd = torch.zeros_like(Hd)
        dH = (torch.randn(R * H, device=dev) * 0.1).to(torch.bfloat16)
        dZ = torch.zeros(R * 4 * H, dtype=torch.bfloat16, device=dev); db = torch.zeros(4 * H, device=dev)
        f = t_ms(lambda: _lib.check(lib.dj_lstm_fwd(1, H, tiles, steps, p(Z), p(G8), p(upf), p(Hd), p(Cd), 0, st()), "fwd"))
        b = t_ms(lambda: _lib.check(lib.dj_lstm_bwd(1, H, tiles, steps, p(G8), p(upb), p(Cd), p(dH), p(dZ), 0, p(db), 0, st()), "bwd"))
        print(f"H={H} tiles={tiles}: fwd {f:.3f} ms  bwd {b:.3f} ms", flush=True)
        del Z, G8, Hd, Cd, dH, dZ
